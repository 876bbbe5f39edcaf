// pcm_conv3x3_r64_kernel -- the rank-64 down-projection of a conv LoRA factor, t = conv3x3(x, A) with A: C -> 64 channels
// (peft lora.Conv2d.lora_A under train_pcm_lora_sd15.py:866-885; stride 1, padding 1, source read as is).
//
// As a member of the implicit-GEMM family this call has N = 64: every output pixel needs all 9*C*64 weights but produces only 64 values,
// so the im2col view re-reads each activation nine times for a quarter of a 256-wide tile's arithmetic -- the 4-wave GEMM tile ran it
// at 250-540 TFLOP/s, fabric bound (3.8 ms per bs-16 step over the resnet convs).  Here a workgroup owns a patch of 128 pixels
// (Wt = min(W, 64) columns x R = 128 / Wt rows of ONE image) and walks the channels in chunks of 64:
//   * the patch's zero-padded halo window, (R + 2) x (Wt + 2) pixels x 64 channels, is staged ONCE per chunk by LDS-DMA (image borders are
//     out-of-range buffer offsets, which the hardware zero-fills) -- the nine taps are nine scalar offsets into it: x is read once;
//   * the weights of the chunk are staged one tap row (3 taps x 64 ranks x 64 channels = 24 KB) at a time, triple-buffered, so one
//     workgroup barrier per step (12 MFMAs per wave) is enough: the image written during step s was last read in step s - 2;
//   * 8 waves = 4 pixel groups x 2 rank halves, one 32 x 32 accumulator each: D[rank][pixel] += A[rank][c] X[c][pixel], both operands read
//     k-contiguous (ds_read_b128) from row-major 128-B rows whose 16-B chunks are XOR-swizzled with (row >> 1) & 7 (applied to the SOURCE
//     chunk a DMA lane fetches, the DMA writes LDS linearly).
// LDS: 2 x 40 KB windows + 3 x 24 KB weight images = 152 KB, one workgroup per CU.
#include <stdlib.h>
#include <string.h>

#include "gemm_dev.h"

#define CR_RSRC_FLAGS 0x00020000
#define CR_OOB 0x80000000u
#define CR_WIN 40960          // 40 groups of 8 window entries x 128 B (up to 320 entries; a 4 x 66 window has 264)
#define CR_ATH 24576          // one tap row of weights: 3 taps x 64 ranks x 128 B

struct ConvR64 {
  const bf16_t* x; const bf16_t* a; bf16_t* out;
  int C, H, W, HW, Wt, lw, R, tiles_x, tiles_per_img, ldo;
};

__global__ __launch_bounds__(512) void pcm_conv3x3_r64_kernel(ConvR64 g) {
#if PCM_KERNEL_BODY
  PCM_DYN_SMEM(smem);
  char* const Win = smem;                     // [2][CR_WIN]
  char* const Abuf = smem + 2 * CR_WIN;       // [3][CR_ATH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave & 3, rh = wave >> 2;
  const int tile = blockIdx.x, bimg = tile / g.tiles_per_img, tin = tile - bimg * g.tiles_per_img;
  const int tyt = tin / g.tiles_x, txt = tin - tyt * g.tiles_x;
  const int y0 = tyt * g.R, x0 = txt * g.Wt;
  const int WP = g.Wt + 2, E = (g.R + 2) * WP, C = g.C, nch = C >> 6, S = 3 * nch;
  const unsigned c2 = (unsigned)C * 2u, k2 = 9u * c2;
  // ---- DMA geometry.  One instruction = 8 rows x 8 chunks (lane -> row lane >> 3, LDS chunk lane & 7); LDS chunk p of row rho holds source
  // chunk p ^ ((rho >> 1) & 7).  Window rows = entries (ry, cx) <-> pixel (y0 - 1 + ry, x0 - 1 + cx); weight rows = (tap column, rank).
  unsigned w_voff[5];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const int e = 8 * (wave + 8 * j) + (lane >> 3);
    const int ry = e / WP, cx = e - ry * WP;
    const int y = y0 - 1 + ry, x = x0 - 1 + cx;
    const bool ok = e < E && y >= 0 && y < g.H && x >= 0 && x < g.W;
    w_voff[j] = ok ? (unsigned)(bimg * g.HW + y * g.W + x) * c2 + (unsigned)(((lane & 7) ^ ((e >> 1) & 7)) << 4) : CR_OOB;
  }
  unsigned a_voff[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int rho = 8 * (wave + 8 * j) + (lane >> 3), tl = rho >> 6, r = rho & 63;
    a_voff[j] = (unsigned)r * k2 + (unsigned)tl * c2 + (unsigned)(((lane & 7) ^ ((rho >> 1) & 7)) << 4);
  }
  auto issue_w = [&](int ch) {          // window of channel chunk ch
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.x, 0, CR_OOB, CR_RSRC_FLAGS);
    char* dst = Win + (ch & 1) * CR_WIN + wave * 1024;
#pragma unroll
    for (int j = 0; j < 5; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, PCM_AS3(dst + j * 8192), 16, w_voff[j], (unsigned)(ch * 128), 0, 0);
  };
  auto issue_a = [&](int s) {           // weights of step s = (chunk s / 3, tap row s % 3)
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.a, 0, CR_OOB, CR_RSRC_FLAGS);
    const int ch = s / 3, ty = s - 3 * ch;
    char* dst = Abuf + (s % 3) * CR_ATH + wave * 1024;
    const unsigned soff = (unsigned)(3 * ty) * c2 + (unsigned)(ch * 128);
#pragma unroll
    for (int j = 0; j < 3; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, PCM_AS3(dst + j * 8192), 16, a_voff[j], soff, 0, 0);
  };
  // ---- fragment geometry: lane (n = lane & 31, kh = lane >> 5).  A operand row = rank 32 rh + n; B operand column = pixel 32 pg + n.
  const int n = lane & 31, kh = lane >> 5;
  const int ar = 32 * rh + n;
  const int pl = 32 * pg + n, yl = pl >> g.lw, xl = pl & (g.Wt - 1);
  const int e0 = yl * WP + xl;                  // window entry of tap (0, 0) for this lane's pixel
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;

  issue_w(0);
  issue_a(0);
  if (1 < S) issue_a(1);
  for (int s = 0; s < S; s++) {
    const int ch = s / 3, ty = s - 3 * ch;
    // everything issued AFTER the data of this step may stay in flight: the next step's weights (3 instructions) and, from the second
    // step of a chunk on, the next chunk's window (5), which is issued right after the weights of step 3 ch + 2
    const bool more_ch = ch + 1 < nch;
    const int pending = ty == 0 ? (s + 1 < S ? 3 : 0) : (more_ch ? 8 : (ty == 1 && s + 1 < S ? 3 : 0));
    if (pending == 8) { PCM_WAIT_VMCNT(8); } else if (pending == 3) { PCM_WAIT_VMCNT(3); } else { PCM_WAIT_VMCNT(0); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();       // every wave's pieces of step s have landed; every wave is done with step s - 1
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");      // the fragment reads below stay below (the DMA's LDS writes are invisible to the compiler)
    const char* A = Abuf + (s % 3) * CR_ATH;
    const char* X = Win + (ch & 1) * CR_WIN;
#pragma unroll
    for (int tl = 0; tl < 3; tl++) {
      const int e = e0 + ty * WP + tl;
      const char* xp = X + e * 128;
      const int xs = (e >> 1) & 7;
      const int arow = tl * 64 + ar;
      const char* ap = A + arow * 128;
      const int as = (arow >> 1) & 7;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bf16x8 af = *(const bf16x8*)(ap + (((2 * k + kh) ^ as) << 4));
        const bf16x8 xf = *(const bf16x8*)(xp + (((2 * k + kh) ^ xs) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xf, acc, 0, 0, 0);   // D[i = rank][j = pixel]
      }
    }
    if (s + 2 < S) issue_a(s + 2);                 // into the image last read in step s - 1 (all waves passed this step's barrier)
    if (ty == 0 && more_ch) issue_w(ch + 1);       // into the window last read in chunk ch - 1
  }
  // ---- epilogue: lane -> pixel, registers -> ranks (e & 3) + 8 (e >> 2) + 4 kh: four 8-byte pieces of the pixel's 128-B row of t
  const size_t m = (size_t)bimg * g.HW + (size_t)(y0 + yl) * g.W + x0 + xl;
  bf16_t* orow = g.out + m * g.ldo + 32 * rh + 4 * kh;
#pragma unroll
  for (int q = 0; q < 4; q++)
    *(uint2*)(orow + 8 * q) = make_uint2(pack_bf2(acc[4 * q], acc[4 * q + 1]), pack_bf2(acc[4 * q + 2], acc[4 * q + 3]));
#endif
}

// -1: PCM_CONV_R64 env (default 1); 0 = always the generic GEMM path (A/B, tests); 2 = also below the size threshold
PCM_LAZY_KNOB(cr64_mode, g_cr64_mode, "PCM_CONV_R64", 1)
PCM_TOOLS_ONLY(extern "C" void pcm_debug_conv_r64(int mode) { g_cr64_mode = mode; }
               static long g_cr64_count = 0;
               extern "C" long pcm_debug_conv_r64_count(void) { return g_cr64_count; })

// returns 0 when this kernel took the call, 1 when the call is not one of its shapes (caller falls back), < 0 on error
// plan_only: decide and report (0 / 1), launch nothing
int pcm_conv_r64_launch(const GemmDev& d, void* stream, bool plan_only) {
  const SegDev& s = d.seg[0];
  if (!cr64_mode() || d.nseg != 1 || s.mode != PCM_SEG_CONV3X3 || d.N != 64 || d.out_f32 || d.bias || d.rowvec || d.res ||
      d.act != PCM_ACT_NONE || d.alpha != 1.0f || s.stride != 1 || s.src_mode != PCM_SRC_DIRECT || s.Hs != d.Ho || s.Ws != d.Wo || (s.C % 64) || (d.ldo % 4))
    return 1;
  const int H = d.Ho, W = d.Wo;
  if (W < 8 || (W & (W - 1)) || (size_t)d.M * s.C * 2 >= 0x7ff00000u) return 1;
  // one workgroup per 128 pixels and one workgroup per CU: below ~256 patches the generic tile (split over K) fills the chip better
  // (measured on MI355X, tools/conv_r64_ab.py: 64x64 x1.3-1.8, 32x32 x1.9-2.1 at M >= 32768; 16x16 at M = 8192 x0.5)
  if (d.M < PCM_GRID_CAP(256) * 128 && cr64_mode() != 2) return 1;
  ConvR64 g;
  g.Wt = W < 64 ? W : 64; g.R = 128 / g.Wt;
  if (H % g.R) return 1;
  if (plan_only) return 0;
  g.lw = 31 - __builtin_clz((unsigned)g.Wt);
  g.x = s.a; g.a = s.w; g.out = (bf16_t*)d.out; g.C = s.C; g.H = H; g.W = W; g.HW = H * W; g.ldo = d.ldo;
  g.tiles_x = W / g.Wt; g.tiles_per_img = g.tiles_x * (H / g.R);
  const int nimg = d.M / g.HW;
  const size_t smem = 2 * CR_WIN + 3 * CR_ATH;
  static bool lds_ok = false;
  if (!lds_ok) {
    hipError_t er = hipFuncSetAttribute((const void*)pcm_conv3x3_r64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    PCM_CHECK(er == hipSuccess, PCM_EHIP, "pcm_gemm_bf16: hipFuncSetAttribute(LDS %zu): %s", smem, hipGetErrorString(er));
    lds_ok = true;
  }
  PCM_LAUNCH(pcm_conv3x3_r64_kernel, dim3(nimg * g.tiles_per_img), dim3(512), smem, stream, g);
  PCM_TOOLS_ONLY(g_cr64_count++;)
  return 0;
}
