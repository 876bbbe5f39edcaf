// pcm_conv3x3_wgrad_bf16 -- DENSE weight gradient of a 3x3 / stride 1 / pad 1 convolution, channels-last:
//     dW[co][kh][kw][ci] += alpha * sum_{b,y,x} dy[b][y][x][co] * x[b][y + kh - 1][x + kw - 1][ci]          (zero outside the image)
// This is the autograd of the trainable nn.Conv2d layers of the reference's DiscriminatorHead (discriminator_sd15.py:349-362) in the
// discriminator step (train_pcm_lora_sd15_adv.py:1383-1391): 36 heads x 2 convs, C = 320 / 640 / 1280, 10.9 TFLOP per step at bs 8.
// Until round 4 they ran as Cout/64 launches of the rank-64 LoRA kernel (wgrad_tr.hip): x re-staged per 64 output channels, every block
// ending in 36864 fp32 atomics, ~160 blocks per launch -- 236 TFLOP/s over the step.
//
// One workgroup (8 waves, two per SIMD) owns 64 output channels x 128 input channels x ALL 9 taps: wave (wr, wq) = 32 co x 32 ci x 9
// taps = 9 accumulator tiles of 32x32 (144 accumulator registers; 18 tiles per wave at one wave per SIMD do not fit the 256 AGPRs and
// hipcc then moves two tiles through v_accvgpr copies every stage).  The contraction runs over pixels, 64 per stage, taken as an
// 8x8 PATCH of one image (not 64 consecutive pixels): the dy operand is staged once as the zero-framed 10x10 window of the patch (100
// entries of 64 channels; a 1x64 strip would need 198), and the 9 shifted dy operands of a k-step are transpose reads
// (ds_read_b64_tr_b16) of that window at 9 compile-time offsets; image borders are the window's zero frame, written by the LDS-DMA itself
// through out-of-range buffer offsets.  A stage moves 28.8 KB of operands for 9.4 MFLOP (the rank-64 kernel: up to 36 KB for 4.7).
// Both LDS images carry an XOR swizzle on the 16-byte chunk index (applied to the SOURCE chunk a DMA lane fetches) so that the 4 rows x
// 32 B quads of a transpose read fall on distinct banks: x tile rows of 256 B: chunk ^ ((row & 3) << 2); window entries of 128 B:
// chunk ^ (((entry >> 1) & 1) << 2) -- with the 8x8 patch every entry index is lane constant + compile-time constant, so the swizzled
// address of a tap is one of four per-lane bases (by the constant's value mod 4) plus an immediate.
// Work split: tiles (Cin/128 x Cout/64) x an M split.  A block's 73728 outputs are plain read-add-write when the tile has ONE owner
// (no M split: C = 1280) and fp32 atomics otherwise; the M split is only as fine as needed for ~one block per CU.
#include <stdlib.h>
#include <string.h>

#include "pcm_common.h"

#define WD_RSRC_FLAGS 0x00020000
#define WD_OOB 0x80000000u

struct WdDev {
  const bf16_t* x; const bf16_t* dy; float* out;
  int H, W, Cin, Cout;
  float alpha;
  int tiles_c, tiles_r, nblocks, grid8;   // grid8 = launched blocks / 8 (XCD-aware logical index)
  int spb, nstages, atomic;               // stages (8x8 patches) per block, total, epilogue form
};

__global__ __launch_bounds__(512) void pcm_wgrad_dense_kernel(WdDev a) {
#if PCM_KERNEL_BODY
  constexpr int XB = 64 * 256, UB = 128 * 128, STAGE = XB + UB, NBUF = 3;
  PCM_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave & 3, wr = wave >> 2;
  // consecutive blocks go to consecutive XCDs: give every XCD a contiguous range of logical tiles (they share x / dy tiles in its L2)
  const int L = (int)(blockIdx.x & 7) * a.grid8 + (int)(blockIdx.x >> 3);
  if (L >= a.nblocks) return;
  const int tc = L % a.tiles_c, t2 = L / a.tiles_c, tr = t2 % a.tiles_r, ms = t2 / a.tiles_r;
  const int c0 = tc * 128, r0 = tr * 64;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int PW = W >> 3, PH = H >> 3;
  const int st_begin = ms * a.spb;
  int st_end = st_begin + a.spb; if (st_end > a.nstages) st_end = a.nstages;
  const int nst = st_end - st_begin;

  // ---- DMA geometry (lane constants).  x: one instruction = 4 pixels x 16 chunks; window: 8 entries x 8 chunks.
  unsigned x_voff[2];
  int u_const[2], u_flags[2];
#pragma unroll
  for (int jj = 0; jj < 2; jj++) {
    const int p = 4 * (wave + 8 * jj) + (lane >> 4);                         // pixel of the patch: (p >> 3, p & 7)
    const int src = (lane & 15) ^ ((p & 3) << 2);
    x_voff[jj] = (c0 + 8 * src < Cin) ? (unsigned)(((p >> 3) * W + (p & 7)) * Cin + c0 + 8 * src) * 2u : WD_OOB;
    const int e = 8 * (wave + 8 * jj) + (lane >> 3);                         // window entry (ry, cx) = (e / 10, e % 10); e >= 100: unused
    const int ry = e / 10, cx = e - ry * 10;
    const int usrc = (lane & 7) ^ (((e >> 1) & 1) << 2);
    u_const[jj] = (((ry - 1) * W + (cx - 1)) * Cout + r0 + 8 * usrc) * 2;   // relative to the patch's first pixel (may be negative)
    u_flags[jj] = (e >= 100 ? 16 : 0) | (ry == 0 ? 1 : 0) | (ry == 9 ? 2 : 0) | (cx == 0 ? 4 : 0) | (cx == 9 ? 8 : 0);
  }
  // the stage to issue next: image / patch row / patch column (advanced incrementally)
  int i_img, i_py, i_px;
  {
    const int ppi = PW * PH;
    i_img = st_begin / ppi;
    const int rem = st_begin - i_img * ppi;
    i_py = rem / PW; i_px = rem - i_py * PW;
  }
  auto issue = [&](int buf) {
    const int pix0 = (i_img * H + 8 * i_py) * W + 8 * i_px;
    const int smask = 16 | (i_py == 0 ? 1 : 0) | (i_py == PH - 1 ? 2 : 0) | (i_px == 0 ? 4 : 0) | (i_px == PW - 1 ? 8 : 0);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, WD_OOB, WD_RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, WD_OOB, WD_RSRC_FLAGS);
    char* base = smem + buf * STAGE;
    const unsigned xs = (unsigned)pix0 * (unsigned)Cin * 2u;
    const int us = pix0 * Cout * 2;
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, PCM_AS3(base + (wave + 8 * jj) * 1024), 16, x_voff[jj], xs, 0, 0);
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      const unsigned voff = (u_flags[jj] & smask) ? WD_OOB : (unsigned)(u_const[jj] + us);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, PCM_AS3(base + XB + (wave + 8 * jj) * 1024), 16, voff, 0, 0, 0);
    }
    if (++i_px == PW) { i_px = 0; if (++i_py == PH) { i_py = 0; i_img++; } }
  };

  // ---- transpose-read geometry (wgrad_tr.hip): 16-lane group gq -> column half cb, k half kg; source lane 4j + q of the group addresses
  // the quad (row 8 kg + j [+4 for the second read], columns 4q..4q+3 of the group's 16 columns)
  const int sl = lane & 15, j = sl >> 2, q = sl & 3, gq = lane >> 4, cb = gq & 1, kg = gq >> 1;
  int u_base[4];
  const int x_base = (8 * kg + j) * 256 + (((4 * wq + 2 * cb + (q >> 1)) ^ (j << 2)) * 16) + 8 * (q & 1);
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const int el = 10 * kg + j;                                   // lane part of the window entry index
    u_base[d] = XB + el * 128 + (((4 * wr + 2 * cb + (q >> 1)) ^ ((((el + d) >> 1) & 1) << 2)) * 16) + 8 * (q & 1);
  }
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

  if (nst > 0) issue(0);
  if (nst > 1) issue(1);
  int buf = 0;
  for (int st = 0; st < nst; st++) {
    if (st + 2 < nst) { issue(buf >= 1 ? buf - 1 : 2); PCM_WAIT_VMCNT(4); }
    else if (st + 1 < nst) { PCM_WAIT_VMCNT(2); }
    else { PCM_WAIT_VMCNT(0); }
    __builtin_amdgcn_s_barrier();            // every wave's pieces of stage st have landed
    const char* S = smem + buf * STAGE;
    const char* px = S + x_base;
    const char* pu[4] = {S + u_base[0], S + u_base[1], S + u_base[2], S + u_base[3]};
    // 12 pipeline steps per stage = 4 k-steps x 3 tap rows: step n waits for its own fragments, puts those of step n + 1 in flight
    // (untracked asm reads into the other register set) and issues its 3 MFMAs; the SIMD's other wave fills the gaps
    bf16x4 xlo[2], xhi[2], ulo[2][3], uhi[2][3];
    // window entry of pixel (2 KS + kg, 4 h + j) shifted by tap (TY, TX): lane part 10 kg + j, constant part below
#define WD_EIMM(KS, TY, TX, Hh) ((2 * (KS) + 2 - (TY)) * 10 + 4 * (Hh) + 2 - (TX))
#define WD_UREAD(dst, KS, TY, TX, Hh) PCM_TR16_ISSUE(dst, pu[WD_EIMM(KS, TY, TX, Hh) & 3], WD_EIMM(KS, TY, TX, Hh) * 128)
#define WD_ISSUE(N)                                                                                                  \
  {                                                                                                                  \
    constexpr int KS = (N) / 3, G = (N) % 3, BU = (N) & 1, BX = KS & 1;                                              \
    if (G == 0) { PCM_TR16_ISSUE(xlo[BX], px, KS * 4096); PCM_TR16_ISSUE(xhi[BX], px, KS * 4096 + 1024); }          \
    WD_UREAD(ulo[BU][0], KS, G, 0, 0); WD_UREAD(uhi[BU][0], KS, G, 0, 1);                                            \
    WD_UREAD(ulo[BU][1], KS, G, 1, 0); WD_UREAD(uhi[BU][1], KS, G, 1, 1);                                            \
    WD_UREAD(ulo[BU][2], KS, G, 2, 0); WD_UREAD(uhi[BU][2], KS, G, 2, 1);                                            \
  }
#define WD_STEP(N)                                                                                                   \
  {                                                                                                                  \
    constexpr int KS = (N) / 3, G = (N) % 3, BU = (N) & 1, BX = KS & 1;                                              \
    PCM_TR16_WAIT8(xlo[BX], xhi[BX], ulo[BU][0], uhi[BU][0], ulo[BU][1], uhi[BU][1], ulo[BU][2], uhi[BU][2]);        \
    const bf16x8 xf = pcm_join4(xlo[BX], xhi[BX]);                                                                   \
    const bf16x8 u0 = pcm_join4(ulo[BU][0], uhi[BU][0]), u1 = pcm_join4(ulo[BU][1], uhi[BU][1]), u2 = pcm_join4(ulo[BU][2], uhi[BU][2]); \
    if ((N) + 1 < 12) WD_ISSUE(((N) + 1) % 12)                                                                       \
    acc[3 * G + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u0, xf, acc[3 * G + 0], 0, 0, 0);   /* D[i = co][j = ci] */ \
    acc[3 * G + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u1, xf, acc[3 * G + 1], 0, 0, 0);                       \
    acc[3 * G + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u2, xf, acc[3 * G + 2], 0, 0, 0);                       \
  }
    WD_ISSUE(0)
    WD_STEP(0) WD_STEP(1) WD_STEP(2) WD_STEP(3) WD_STEP(4) WD_STEP(5) WD_STEP(6) WD_STEP(7) WD_STEP(8) WD_STEP(9) WD_STEP(10) WD_STEP(11)
#undef WD_ISSUE
#undef WD_STEP
#undef WD_UREAD
#undef WD_EIMM
    PCM_WAIT_LGKMCNT0();
    __builtin_amdgcn_s_barrier();            // reads of this buffer are done before stage st + 3 is issued into it
    buf = buf == NBUF - 1 ? 0 : buf + 1;
  }
  // ---- epilogue: lanes along ci (contiguous in dW[co][tap][ci])
  const int l31 = lane & 31, hi = lane >> 5;
  const size_t ldw = (size_t)9 * Cin;
  const int ci = c0 + 32 * wq + l31;
  if (ci < Cin) {
    float* col = a.out + (size_t)(r0 + 32 * wr + 4 * hi) * ldw + ci;
    if (a.atomic) {
#pragma unroll
      for (int t = 0; t < 9; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) atomicAdd(col + (size_t)((e & 3) + 8 * (e >> 2)) * ldw + (size_t)t * Cin, acc[t][e] * a.alpha);
    } else {                                   // the tile has one owner: plain read-add-write, 16 loads in flight at a time
#pragma unroll
      for (int t = 0; t < 9; t++) {
        float old[16];
#pragma unroll
        for (int e = 0; e < 16; e++) old[e] = col[(size_t)((e & 3) + 8 * (e >> 2)) * ldw + (size_t)t * Cin];
#pragma unroll
        for (int e = 0; e < 16; e++) col[(size_t)((e & 3) + 8 * (e >> 2)) * ldw + (size_t)t * Cin] = old[e] + acc[t][e] * a.alpha;
      }
    }
  }
#endif
}

// M split: only as fine as needed for about one block per CU (every extra split is 9 * Cin * Cout further fp32 atomics)
PCM_LAZY_KNOB(wd_forced_msplit, g_wd_msplit, "PCM_WGRAD_DENSE_MSPLIT", 0)
static int wd_msplit(int tiles, int stages) {
  const int forced = wd_forced_msplit();
  int ms = forced > 0 ? forced : (tiles >= 160 ? 1 : PCM_GRID_CAP(256) / tiles);
  if (ms > stages / 8) ms = stages / 8;
  if (ms < 1) ms = 1;
  return ms;
}

extern "C" int pcm_conv3x3_wgrad_bf16(const void* x, const void* dy, float* dW, int B, int H, int W, int Cin, int Cout, float alpha, void* stream) {
  PCM_CHECK(x && dy && dW && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, PCM_EINVAL, "pcm_conv3x3_wgrad_bf16: null/empty");
  PCM_CHECK(PCM_ALIGNED16(x) && PCM_ALIGNED16(dy) && ((size_t)dW & 3) == 0, PCM_EALIGN, "pcm_conv3x3_wgrad_bf16: operand alignment");
  PCM_CHECK((H % 8) == 0 && (W % 8) == 0 && (Cin % 8) == 0 && (Cout % 64) == 0, PCM_EINVAL,
            "pcm_conv3x3_wgrad_bf16: needs H%%8==0, W%%8==0, Cin%%8==0, Cout%%64==0 (got %dx%d, %d -> %d)", H, W, Cin, Cout);
  const size_t M = (size_t)B * H * W;
  PCM_CHECK(M * (size_t)Cin * 2 < 0x7ff00000u && M * (size_t)Cout * 2 < 0x7ff00000u, PCM_EINVAL,
            "pcm_conv3x3_wgrad_bf16: operands beyond the 2 GB buffer range (split the batch)");
  WdDev a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.out = dW; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.alpha = alpha;
  a.tiles_c = (Cin + 127) / 128; a.tiles_r = Cout / 64;
  a.nstages = (int)(M / 64);
  const int tiles = a.tiles_c * a.tiles_r;
  int msplit = wd_msplit(tiles, a.nstages);
  a.spb = (a.nstages + msplit - 1) / msplit;
  msplit = (a.nstages + a.spb - 1) / a.spb;
  a.atomic = msplit > 1;
  a.nblocks = tiles * msplit;
  a.grid8 = (a.nblocks + 7) / 8;
  const size_t smem = 3 * (64 * 256 + 128 * 128);
  static bool lds_ok = false;
  if (!lds_ok) {
    hipError_t er = hipFuncSetAttribute((const void*)pcm_wgrad_dense_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    PCM_CHECK(er == hipSuccess, PCM_EHIP, "pcm_conv3x3_wgrad_bf16: hipFuncSetAttribute(LDS %zu): %s", smem, hipGetErrorString(er));
    lds_ok = true;
  }
  PCM_LAUNCH(pcm_wgrad_dense_kernel, dim3(a.grid8 * 8), dim3(512), smem, stream, a);
  return pcm_post_launch("pcm_conv3x3_wgrad_bf16");
}
