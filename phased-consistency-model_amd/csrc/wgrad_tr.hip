// LoRA weight gradients with LDS-DMA staging and LDS transpose reads (gfx950 ds_read_b64_tr_b16):
//     G[g][r] += alpha * sum_m Big[m][g] * Small[m][r]            (same contract as wgrad.hip)
// The contraction index m is the ROW index of both operands.  wgrad.hip transposes 8x8 blocks in registers on the way into LDS; here the
// tiles go into LDS row-major exactly as they sit in HBM (buffer_load ... lds, 16 B per lane, no VGPR round trip, two stages in
// flight) and BOTH MFMA operands are read k-along-the-rows with two transpose reads per fragment (pcm_common.h PCM_DS_READ_TR16).
//
//  pcm_wgrad_tr_kernel       plain view: block = 128 columns of Big x 64 ranks, 64 rows of m per stage, 4 waves x (32 g x 64 r).
//                            Small is re-read once per 128 Big columns (wgrad.hip: once per 64).
//  pcm_wgrad_tr_conv_kernel  3x3 / stride 1 / pad 1 view (dA of the conv LoRA factors): out[tap][c][r] = sum_p x[p][c] u[p - off(tap)][r].
//                            wgrad.hip walks the 9 taps as 9 im2col column blocks: x is staged 9 times and u 9*C/64 times.  Here a block
//                            owns 64 input channels x 64 ranks x ALL 9 taps (9 accumulator tiles per wave): per 64-pixel stage it stages
//                            the x tile ONCE plus a zero-padded (rows+2) x (width+2) window of u, and the 9 shifted u operands are
//                            transpose reads of that window at 9 scalar offsets (image borders = the window's zero frame, written by
//                            the DMA itself through out-of-range buffer offsets).
// LDS images carry an XOR swizzle on the 16-B chunk index so that the 4 rows x 32 B quads of a transpose read fall on distinct banks;
// the DMA writes LDS linearly, so the swizzle is applied to the SOURCE chunk a lane fetches (guide rule 21).
#include <stdlib.h>
#include <string.h>

#include "wgrad_dev.h"

#define WT_RSRC_FLAGS 0x00020000
#define WT_OOB 0x80000000u

// ------------------------------------------------------------------------------------------------ plain view
template <bool SWAP, typename WG>
__device__ __forceinline__ void pcm_wgrad_tr_body(const WG& a, const int bx, const int by, char* smem) {
  constexpr int BIGB = 64 * 256, SMB = 64 * 128, STAGE = BIGB + SMB;   // Big tile [64][128] bf16, Small tile [64][64] bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g0 = bx * 128;
  const int m_begin = by * a.m_per_block;
  int m_end = m_begin + a.m_per_block; if (m_end > a.M) m_end = a.M;
  const int nst = (m_end - m_begin + 63) >> 6;
  // ---- DMA geometry.  Big: one instruction = 4 rows x 16 chunks (lane -> row lane>>4, LDS chunk lane&15); Small: 8 rows x 8 chunks.
  // LDS chunk position p of row r holds source chunk p ^ swz(r): swz = (r&3)<<2 for the 256-B rows, ((r>>1)&1)<<2 for the 128-B rows.
  const int b_rl = lane >> 4, b_c = (lane & 15) ^ ((b_rl & 3) << 2);
  const int s_rl = lane >> 3, s_c = (lane & 7) ^ (((s_rl >> 1) & 1) << 2);
  const bool b_colok = g0 + 8 * b_c < a.G;
  const unsigned b_coloff = (unsigned)(g0 + 8 * b_c) * 2u, s_coloff = (unsigned)s_c * 16u;
  const unsigned ldb2 = (unsigned)a.ldb * 2u, lds2 = (unsigned)a.lds_ * 2u;
  auto issue = [&](int st, int buf) {
    const int m0 = m_begin + 64 * st;
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.big, 0, WT_OOB, WT_RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.small_, 0, WT_OOB, WT_RSRC_FLAGS);
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      const int t = wave + 4 * jj, row = m0 + 4 * t + b_rl;
      const unsigned voff = (row < m_end && b_colok) ? (unsigned)row * ldb2 + b_coloff : WT_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, PCM_AS3(base + t * 1024), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      const int t = wave + 4 * jj, row = m0 + 8 * t + s_rl;
      const unsigned voff = row < m_end ? (unsigned)row * lds2 + s_coloff : WT_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, PCM_AS3(base + BIGB + t * 1024), 16, voff, 0, 0, 0);
    }
  };
  // ---- transpose-read geometry: 16-lane group gq = lane>>4 -> column half cb = gq&1, k half kg = gq>>1; source lane 4j+q of the group
  // addresses the quad (row 8kg + j [+4 for the second read], columns 4q..4q+3 of the group's 16 columns)
  const int sl = lane & 15, j = sl >> 2, q = sl & 3, gq = lane >> 4, cb = gq & 1, kg = gq >> 1;
  const int a_chunk = (4 * wave + 2 * cb + (q >> 1)) ^ (j << 2);
  const int a_base = (8 * kg + j) * 256 + a_chunk * 16 + 8 * (q & 1);
  const int jb = (j >> 1) & 1;
  int b_base[2];
#pragma unroll
  for (int rt = 0; rt < 2; rt++) b_base[rt] = (8 * kg + j) * 128 + (((4 * rt + 2 * cb + (q >> 1)) ^ (jb << 2)) * 16) + 8 * (q & 1);
  f32x16 acc[2];
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[rt][r] = 0.f;

  issue(0, 0);
  for (int st = 0; st < nst; st++) {
    if (st + 1 < nst) { issue(st + 1, (st + 1) & 1); PCM_WAIT_VMCNT(6); } else { PCM_WAIT_VMCNT(0); }
    __builtin_amdgcn_s_barrier();            // every wave's pieces of stage st have landed
    const char* B = smem + (st & 1) * STAGE;
    const char* S = B + BIGB;
    // all 24 transpose reads of the stage as untracked asm reads (the builtin form makes hipcc drain the DMA queue: pcm_common.h), one wait
    bf16x4 alo[4], ahi[4], blo[2][4], bhi[2][4];
    const char* pa = B + a_base;
    const char* pb0 = S + b_base[0];
    const char* pb1 = S + b_base[1];
#define WT_KS(KS)                                                                                                   \
  PCM_TR16_ISSUE(alo[KS], pa, (16 * KS) * 256); PCM_TR16_ISSUE(ahi[KS], pa, (16 * KS + 4) * 256);                    \
  PCM_TR16_ISSUE(blo[0][KS], pb0, (16 * KS) * 128); PCM_TR16_ISSUE(bhi[0][KS], pb0, (16 * KS + 4) * 128);            \
  PCM_TR16_ISSUE(blo[1][KS], pb1, (16 * KS) * 128); PCM_TR16_ISSUE(bhi[1][KS], pb1, (16 * KS + 4) * 128);
    WT_KS(0) WT_KS(1) WT_KS(2) WT_KS(3)
#undef WT_KS
    PCM_TR16_WAIT8(alo[0], ahi[0], alo[1], ahi[1], alo[2], ahi[2], alo[3], ahi[3]);
    PCM_TR16_KEEP8(blo[0][0], bhi[0][0], blo[0][1], bhi[0][1], blo[0][2], bhi[0][2], blo[0][3], bhi[0][3]);
    PCM_TR16_KEEP8(blo[1][0], bhi[1][0], blo[1][1], bhi[1][1], blo[1][2], bhi[1][2], blo[1][3], bhi[1][3]);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const bf16x8 af = pcm_join4(alo[ks], ahi[ks]);
#pragma unroll
      for (int rt = 0; rt < 2; rt++) {
        const bf16x8 bf = pcm_join4(blo[rt][ks], bhi[rt][ks]);
        acc[rt] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, af, acc[rt], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[rt], 0, 0, 0);
      }
    }
    PCM_WAIT_LGKMCNT0();
    __builtin_amdgcn_s_barrier();            // reads of this buffer are done before stage st+2 is issued into it
  }
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int ii = (e & 3) + 8 * (e >> 2) + 4 * hi;
      const int g = g0 + 32 * wave + (SWAP ? l31 : ii), r = 32 * rt + (SWAP ? ii : l31);
      if (g < a.G) wg_emit(a, (size_t)g * a.g_stride + (size_t)r * a.r_stride, by, g, r, acc[rt][e] * a.alpha);
    }
}
template <bool SWAP>
__global__ __launch_bounds__(256) void pcm_wgrad_tr_kernel(WgDev a) {
#if PCM_KERNEL_BODY
  PCM_DYN_SMEM(smem);
  pcm_wgrad_tr_body<SWAP>(a, blockIdx.x, blockIdx.y, smem);
#endif
}
// Several independent weight gradients in ONE launch (the dA / dB pair of a LoRA module, the six of a fused q/k/v projection): the
// single launches are 3-20 us of work behind ~4 us of launch + ramp each (556 per bs-16 step).  1-D grid; a block finds its job by the
// prefix table and reads that job's argument block from the kernarg segment by index (scalar loads; no per-job copies in registers).
struct WgMulti { WgDev d[PCM_WGRAD_MULTI_MAX]; int blk_start[PCM_WGRAD_MULTI_MAX + 1]; int tiles_g[PCM_WGRAD_MULTI_MAX]; int n; };
__global__ __launch_bounds__(256) void pcm_wgrad_tr_multi_kernel(WgMulti mm) {
#if PCM_KERNEL_BODY
  PCM_DYN_SMEM(smem);
  const int bid = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 1; k < PCM_WGRAD_MULTI_MAX; k++)
    if (k < mm.n && bid >= mm.blk_start[k]) i = k;
  const int lid = bid - mm.blk_start[i], tg = mm.tiles_g[i];
  const int by = lid / tg, bx = lid - by * tg;
  const auto& a = PCM_KERNARG_REF(WgDev, mm.d, i);   // d[] is the first member
  if (a.swap) pcm_wgrad_tr_body<true>(a, bx, by, smem);
  else pcm_wgrad_tr_body<false>(a, bx, by, smem);
#endif
}

// ------------------------------------------------------------------------------------------------ 3x3 / stride 1 view
// stage = 64 consecutive pixels of one image: Wt = min(W, 64) columns x R = 64/Wt rows at (y0, x0); u window = (R+2) x (Wt+2) entries of
// 128 B, entry (ry, cx) <-> pixel (y0 - 1 + ry, x0 - 1 + cx), zero outside the image.  LDS: x tile 8 KB + window 28 KB, two stages.
__global__ __launch_bounds__(256) void pcm_wgrad_tr_conv_kernel(WgDev a) {
#if PCM_KERNEL_BODY
  constexpr int XB = 64 * 128, UB = 28 * 1024, STAGE = XB + UB;
  PCM_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wr = wave >> 1;
  const int c0 = blockIdx.x * 64;
  const int W = a.Wo, H = a.Ho, HW = H * W, C = a.C;
  const int Wt = W < 64 ? W : 64, R = 64 / Wt, WP = Wt + 2, E = (R + 2) * WP;
  const int lw = 31 - __builtin_clz((unsigned)Wt);
  const int st_begin = blockIdx.y * (a.m_per_block >> 6);
  int st_end = st_begin + (a.m_per_block >> 6);
  if (st_end > (a.M >> 6)) st_end = a.M >> 6;
  const int nst = st_end - st_begin, spi = HW >> 6;       // stages per image
  // ---- DMA geometry
  const int x_rl = lane >> 3;
  int u_ry[7], u_cx[7];
#pragma unroll
  for (int jj = 0; jj < 7; jj++) {
    const int e = 8 * (wave + 4 * jj) + (lane >> 3);
    u_ry[jj] = e < E ? e / WP : -4096;        // entries beyond the window: always out of range (zero fill keeps the wait counts uniform)
    u_cx[jj] = e < E ? e - (e / WP) * WP : 0;
  }
  const unsigned u_coloff = (unsigned)(lane & 7) * 16u, lds2 = (unsigned)a.lds_ * 2u, c2 = (unsigned)C * 2u;
  auto issue = [&](int sti, int buf) {
    const int sg = st_begin + sti, bimg = sg / spi, pix0 = (sg - bimg * spi) << 6;
    const int y0 = pix0 / W, x0 = pix0 - y0 * W;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.big, 0, WT_OOB, WT_RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)a.small_, 0, WT_OOB, WT_RSRC_FLAGS);
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {          // x tile: 8 pixels x 8 chunks per instruction, chunk swizzle ((row>>1)&1)<<2
      const int t = wave + 4 * jj, rl = 8 * t + x_rl;
      const int c = (lane & 7) ^ (((rl >> 1) & 1) << 2);
      const unsigned voff = (unsigned)(bimg * HW + pix0 + rl) * c2 + (unsigned)(c0 + 8 * c) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, PCM_AS3(base + t * 1024), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int jj = 0; jj < 7; jj++) {          // u window: 8 entries x 8 chunks per instruction, linear
      const int t = wave + 4 * jj;
      const int y = y0 - 1 + u_ry[jj], x = x0 - 1 + u_cx[jj];
      const bool ok = y >= 0 && y < H && x >= 0 && x < W;
      const unsigned voff = ok ? (unsigned)(bimg * HW + y * W + x) * lds2 + u_coloff : WT_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, PCM_AS3(base + XB + t * 1024), 16, voff, 0, 0, 0);
    }
  };
  // ---- transpose-read geometry (see the plain kernel); pixel quads never straddle an image row (Wt % 4 == 0)
  const int sl = lane & 15, j = sl >> 2, q = sl & 3, gq = lane >> 4, cb = gq & 1, kg = gq >> 1;
  const int x_base = (8 * kg + j) * 128 + (((4 * wc + 2 * cb + (q >> 1)) ^ (((j >> 1) & 1) << 2)) * 16) + 8 * (q & 1);
  int u_base[4][2];                            // [k-step][read]: byte offset of entry (yl, xl + j) of the un-shifted tap (ty, tx) = (2, 2)
#pragma unroll
  for (int ks = 0; ks < 4; ks++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int pl = 16 * ks + 8 * kg + 4 * h, yl = pl >> lw, xl = pl & (Wt - 1);
      u_base[ks][h] = (yl * WP + xl + j) * 128 + (32 * wr + 16 * cb + 4 * q) * 2;
    }
  int toff[9];                                 // wave-uniform: entry of pixel p - off(tap) relative to tap (2, 2), in bytes
#pragma unroll
  for (int t = 0; t < 9; t++) toff[t] = ((2 - t / 3) * WP + (2 - t % 3)) * 128;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

  if (nst > 0) issue(0, 0);
  for (int st = 0; st < nst; st++) {
    if (st + 1 < nst) { issue(st + 1, (st + 1) & 1); PCM_WAIT_VMCNT(9); } else { PCM_WAIT_VMCNT(0); }
    __builtin_amdgcn_s_barrier();
    const char* X = smem + (st & 1) * STAGE;
    const char* U = X + XB;
    // 12 pipeline steps per stage = 4 k-steps x 3 tap rows: step n waits for its own fragments (x of the k-step + 3 shifted u), puts the
    // fragments of step n+1 in flight (untracked asm reads, double-buffered registers) and multiplies -- LDS latency under 3 MFMAs
    bf16x4 xlo[2], xhi[2], ulo[2][3], uhi[2][3];
    const char* px = X + x_base;
#define WC_ISSUE(N)                                                                                                  \
  {                                                                                                                  \
    constexpr int KS = (N) / 3, G = (N) % 3, BU = (N) & 1;                                                           \
    if (G == 0) { PCM_TR16_ISSUE(xlo[KS & 1], px, (16 * KS) * 128); PCM_TR16_ISSUE(xhi[KS & 1], px, (16 * KS + 4) * 128); } \
    const char* q0 = U + u_base[KS][0];                                                                              \
    const char* q1 = U + u_base[KS][1];                                                                              \
    PCM_TR16_ISSUE(ulo[BU][0], q0 + toff[3 * G + 0], 0); PCM_TR16_ISSUE(uhi[BU][0], q1 + toff[3 * G + 0], 0);         \
    PCM_TR16_ISSUE(ulo[BU][1], q0 + toff[3 * G + 1], 0); PCM_TR16_ISSUE(uhi[BU][1], q1 + toff[3 * G + 1], 0);         \
    PCM_TR16_ISSUE(ulo[BU][2], q0 + toff[3 * G + 2], 0); PCM_TR16_ISSUE(uhi[BU][2], q1 + toff[3 * G + 2], 0);         \
  }
#define WC_STEP(N)                                                                                                   \
  {                                                                                                                  \
    constexpr int KS = (N) / 3, G = (N) % 3, BU = (N) & 1;                                                           \
    PCM_TR16_WAIT8(xlo[KS & 1], xhi[KS & 1], ulo[BU][0], uhi[BU][0], ulo[BU][1], uhi[BU][1], ulo[BU][2], uhi[BU][2]); \
    const bf16x8 xf = pcm_join4(xlo[KS & 1], xhi[KS & 1]);                                                           \
    const bf16x8 u0 = pcm_join4(ulo[BU][0], uhi[BU][0]), u1 = pcm_join4(ulo[BU][1], uhi[BU][1]), u2 = pcm_join4(ulo[BU][2], uhi[BU][2]); \
    if ((N) + 1 < 12) WC_ISSUE(((N) + 1) % 12)                                                                       \
    acc[3 * G + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u0, xf, acc[3 * G + 0], 0, 0, 0);   /* D[i = r][j = c] */ \
    acc[3 * G + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u1, xf, acc[3 * G + 1], 0, 0, 0);                       \
    acc[3 * G + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u2, xf, acc[3 * G + 2], 0, 0, 0);                       \
  }
    WC_ISSUE(0)
    WC_STEP(0) WC_STEP(1) WC_STEP(2) WC_STEP(3) WC_STEP(4) WC_STEP(5) WC_STEP(6) WC_STEP(7) WC_STEP(8) WC_STEP(9) WC_STEP(10) WC_STEP(11)
#undef WC_ISSUE
#undef WC_STEP
    PCM_WAIT_LGKMCNT0();
    __builtin_amdgcn_s_barrier();
  }
  const int l31 = lane & 31, hi = lane >> 5;
  const int c = c0 + 32 * wc + l31;
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int r = 32 * wr + (e & 3) + 8 * (e >> 2) + 4 * hi;
      wg_emit(a, (size_t)(t * C + c) * a.g_stride + (size_t)r * a.r_stride, blockIdx.y, t * C + c, r, acc[t][e] * a.alpha);
    }
#endif
}

// -1: PCM_WGRAD_TR env (default 1); 0 = always wgrad.hip (A/B, tests)
PCM_LAZY_KNOB(wgtr_mode, g_wgtr_mode, "PCM_WGRAD_TR", 1)
PCM_TOOLS_ONLY(extern "C" void pcm_debug_wgrad_tr(int mode) { g_wgtr_mode = mode; }
               static long g_wgtr_count[2] = {0, 0};   // tests: launches taken by the plain / conv kernel
               extern "C" long pcm_debug_wgrad_tr_count(int conv) { return g_wgtr_count[conv ? 1 : 0]; })
PCM_KNOB int g_wgtr_blocks = 512, g_wgtr_auto = 1;      // tuning hook: n > 0 forces ~n blocks, 0 restores the shipped rule
PCM_TOOLS_ONLY(extern "C" void pcm_debug_wgrad_tr_blocks(int n) { g_wgtr_auto = n <= 0; g_wgtr_blocks = n > 0 ? n : 512; })

// plain view: is it one of this file's, and how is it split?  (fills a.m_per_block; returns false -> caller falls back)
static bool wgtr_plan_plain(WgDev& a, int* tiles_g_out, int* msplit_out) {
  if (!wgtr_mode() || a.out_conv || a.mode == PCM_SEG_CONV3X3 || a.part) return false;     // reproducible-form jobs run one by one
  if ((size_t)a.M * a.lds_ * 2 >= 0x7ff00000u || (size_t)a.M * a.ldb * 2 >= 0x7ff00000u) return false;
  auto cdiv = [](long x, long y) { return (int)((x + y - 1) / y); };
  const int tiles_g = cdiv(a.G, 128), stages = cdiv(a.M, 64);
  int msplit = a.M / 1024;
  if (msplit * tiles_g < 128) msplit = cdiv(128, tiles_g);
  if (!g_wgtr_auto) msplit = cdiv(PCM_GRID_CAP(g_wgtr_blocks), tiles_g);
  if (msplit * tiles_g > PCM_GRID_CAP(2048)) msplit = cdiv(PCM_GRID_CAP(2048), tiles_g);
  if (msplit > cdiv(stages, 2)) msplit = cdiv(stages, 2);
  if (msplit < 1) msplit = 1;
  a.m_per_block = cdiv(stages, msplit) * 64;
  *tiles_g_out = tiles_g; *msplit_out = cdiv(a.M, a.m_per_block);
  return true;
}
PCM_TOOLS_ONLY(static long g_wgtr_multi = 0;   // tests: multi-job launches
               extern "C" long pcm_debug_wgrad_tr_multi_count(void) { return g_wgtr_multi; })
// jobs[0..n): plain-view jobs that wgtr_plan_plain accepted are packed into launches of up to PCM_WGRAD_MULTI_MAX; taken[i] = 1 for those
int pcm_wgrad_tr_launch_multi(const WgDev* jobs, int n, unsigned char* taken, void* stream) {
  WgMulti mm; memset(&mm, 0, sizeof(mm));
  const size_t smem = 2 * (64 * 256 + 64 * 128);
  auto flush = [&]() {
    if (mm.n == 0) return;
    if (mm.n == 1) {
      const WgDev& a = mm.d[0];
      const int msplit = mm.blk_start[1] / mm.tiles_g[0];
      if (a.swap) PCM_LAUNCH((pcm_wgrad_tr_kernel<true>), dim3(mm.tiles_g[0], msplit), dim3(256), smem, stream, a);
      else PCM_LAUNCH((pcm_wgrad_tr_kernel<false>), dim3(mm.tiles_g[0], msplit), dim3(256), smem, stream, a);
    } else {
      PCM_LAUNCH(pcm_wgrad_tr_multi_kernel, dim3(mm.blk_start[mm.n]), dim3(256), smem, stream, mm);
      PCM_TOOLS_ONLY(g_wgtr_multi++;)
    }
    PCM_TOOLS_ONLY(g_wgtr_count[0] += mm.n;)
    mm.n = 0;
  };
  for (int i = 0; i < n; i++) {
    WgDev a = jobs[i];
    int tg = 0, ms = 0;
    taken[i] = 0;
    if (!wgtr_plan_plain(a, &tg, &ms)) continue;
    taken[i] = 1;
    mm.d[mm.n] = a; mm.tiles_g[mm.n] = tg; mm.blk_start[mm.n + 1] = mm.blk_start[mm.n] + tg * ms; mm.n++;
    if (mm.n == PCM_WGRAD_MULTI_MAX) flush();
  }
  flush();
  return 0;
}

// msplit_out: the M split taken (= slabs of the reproducible form); plan_only: decide and report, launch nothing
int pcm_wgrad_tr_launch(const WgDev& a0, void* stream, int* msplit_out, bool plan_only) {
  if (!wgtr_mode() || a0.out_conv) return 1;
  WgDev a = a0;
  const bool conv = a.mode == PCM_SEG_CONV3X3;
  if ((size_t)a.M * a.lds_ * 2 >= 0x7ff00000u) return 1;
  auto cdiv = [](long x, long y) { return (int)((x + y - 1) / y); };
  if (!conv) {
    if ((size_t)a.M * a.ldb * 2 >= 0x7ff00000u) return 1;
    const int tiles_g = cdiv(a.G, 128), stages = cdiv(a.M, 64);
    // split over M.  A block ends with 8192 fp32 atomics (32 KB) against 24 KB of operand reads per stage, and the L2 atomic units, not
    // HBM, bound the launch once the atomics pass ~10 % of the reads: measured (MI355X, tools/jobs/r02_wgrad_blocks.sh) M = 65536 x G = 320 best at
    // 128..256 blocks (17 us; 512: 23.5), G = 2560 at ~1024 (71 us), so: msplit = M / 1024 rows, but at least ~128 blocks in all.
    int msplit = a.M / 1024;
    if (msplit * tiles_g < 128) msplit = cdiv(128, tiles_g);
    if (!g_wgtr_auto) msplit = cdiv(PCM_GRID_CAP(g_wgtr_blocks), tiles_g);
    if (msplit * tiles_g > PCM_GRID_CAP(2048)) msplit = cdiv(PCM_GRID_CAP(2048), tiles_g);
    if (msplit > cdiv(stages, 2)) msplit = cdiv(stages, 2);
    if (msplit < 1) msplit = 1;
    a.m_per_block = cdiv(stages, msplit) * 64;
    msplit = cdiv(a.M, a.m_per_block);
    if (msplit_out) *msplit_out = msplit;
    if (plan_only) return 0;
    const size_t smem = 2 * (64 * 256 + 64 * 128);
    if (a.swap) PCM_LAUNCH((pcm_wgrad_tr_kernel<true>), dim3(tiles_g, msplit), dim3(256), smem, stream, a);
    else PCM_LAUNCH((pcm_wgrad_tr_kernel<false>), dim3(tiles_g, msplit), dim3(256), smem, stream, a);
    PCM_TOOLS_ONLY(g_wgtr_count[0]++;)
    return 0;
  }
  const int W = a.Wo, HW = a.Ho * a.Wo;
  const bool wok = W >= 64 ? (W % 64) == 0 : (W >= 4 && (W & (W - 1)) == 0);
  if (a.M < 4096) return 1;   // few stages per block: staging the 28 KB window per 64 pixels costs more than wgrad.hip's 9 passes (8x8x1280: 24 vs 17 us)
  if (a.stride != 1 || a.src_mode != PCM_SRC_DIRECT || a.Hs != a.Ho || a.Ws != a.Wo || !wok || (HW % 64) || (a.C % 64) || !a.swap) return 1;
  if ((size_t)a.M * a.C * 2 >= 0x7ff00000u) return 1;
  const int tiles_c = a.C / 64, stages = a.M / 64;
  // a block ends with 36864 fp32 atomics (9 taps x 64 x 64): measured best at 128..256 blocks for every UNet level (64x64x320: 80 us at
  // 128..256 blocks, 100 at 512, 216 at 2048; 16x16x1280: 32 us at 128, 58 at 512) -- ~160 blocks, never less than 2 stages per block
  int msplit = cdiv(g_wgtr_auto ? 160 : PCM_GRID_CAP(g_wgtr_blocks), tiles_c);
  if (msplit > cdiv(stages, 2)) msplit = cdiv(stages, 2);
  if (msplit < 1) msplit = 1;
  a.m_per_block = cdiv(stages, msplit) * 64;
  msplit = cdiv(a.M, a.m_per_block);
  if (msplit_out) *msplit_out = msplit;
  if (plan_only) return 0;
  const size_t smem = 2 * (64 * 128 + 28 * 1024);
  static bool lds_ok = false;
  if (!lds_ok) {
    hipError_t er = hipFuncSetAttribute((const void*)pcm_wgrad_tr_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    PCM_CHECK(er == hipSuccess, PCM_EHIP, "pcm_lora_wgrad_bf16: hipFuncSetAttribute(LDS %zu): %s", smem, hipGetErrorString(er));
    lds_ok = true;
  }
  PCM_LAUNCH(pcm_wgrad_tr_conv_kernel, dim3(tiles_c, msplit), dim3(256), smem, stream, a);
  PCM_TOOLS_ONLY(g_wgtr_count[1]++;)
  return 0;
}
