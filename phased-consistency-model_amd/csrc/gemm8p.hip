// pcm_gemm8p_kernel — the large-tile member of the pcm_gemm_bf16 family (same contract as gemm.hip:
// Linear / conv1x1 / implicit-im2col conv3x3 with an optional second K segment for the LoRA branch).
//
// Why a second kernel: the 4-wave 128x128 tile of gemm.hip refills (128+128) x 128 B of LDS per 2.1 MFLOP, which
// is as many L1->LDS cycles (64 B/clk/CU) as MFMA cycles -- it tops out near 0.85 PFLOP/s.  This kernel owns a
// 256 x (64*FN) tile per CU (FN = 5: 256x320, all of N for the 320-channel layers; FN = 4: 256x256), 8 waves as
// 2 (M) x 4 (N), per-wave 128 x (16*FN) outputs held in 8 x FN 16x16 accumulators, BK = 64.
//
// Pipeline (one block per CU, so latency is hidden inside the block):
//   * each K-tile is staged as four LDS-DMA pieces: A0 / A1 (the two 64-row halves of every wave's pixel rows) and
//     B0 / B1 (the first 16*F0 and last 32 of every wave's channel columns); two K-tile buffers.
//   * a K-tile is four phases, one accumulator quadrant each:  C00 = A0xB0, C01 = A0xB1, C11 = A1xB1, C10 = A1xB0.
//     Every phase = { ds_read the operand fragments it needs; issue ONE piece of a later K-tile into the region
//     whose last reader finished the phase before; lgkmcnt(0); barrier; MFMAs; barrier }.
//   * pieces land two K-tiles ahead (B0: one), retired by a single counted wait per K-tile (vmcnt(6), never 0 in
//     steady state) placed before the first barrier of phase 4, i.e. one full phase before the first read.
//   * the two M wave-groups run one barrier apart (group 1 executes one extra barrier up front): on every SIMD one
//     wave is in its MFMA half-phase while the other reads LDS / issues DMA.
// Sources are addressed through raw buffer resources: per-row 32-bit byte offsets in VGPRs, the K advance in the
// scalar offset; out-of-range rows / padding taps use an offset beyond num_records, which the hardware zero-fills.
#include "gemm_dev.h"
#include "gemm_epilogue.h"

// cycle stamps (tools/gemm8p_timeline.py, -DPCM_ABLATE builds only): lane 0 of every wave of ONE mid-grid tile records s_memtime at three
// points of each phase (start | fragment reads retired + barrier passed | MFMAs issued + closing barrier passed) for its first 16 K-tiles
#ifdef PCM_ABLATE
__device__ unsigned long long g_g8_stamps[8][16][12];
extern "C" int pcm_debug_gemm8p_stamps(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_g8_stamps), sizeof(g_g8_stamps)); }
#define G8_STAMP(k) do { if (stamp_on && t < 16 && lane == 0) g_g8_stamps[wave][t][k] = __builtin_readcyclecounter(); } while (0)
#else
#define G8_STAMP(k) do { } while (0)
#endif
#define PCM_RSRC_FLAGS 0x00020000
#define PCM_OOB 0x80000000u

template <int F0, bool MD, bool CO>
__global__ __launch_bounds__(512) void pcm_gemm8p_kernel(GemmDev g) {
#if PCM_KERNEL_BODY   // the host pass only needs the launch stub (buffer-resource builtins are device-only)
  constexpr int F1 = 2, FN = F0 + F1, WNC = 16 * FN, BN = 4 * WNC;
  constexpr int RB0 = 64 * F0;                                         // LDS rows of B part 0 (4 waves x 16*F0)
  constexpr int OFF_A1 = 128 * 128, OFF_B0 = 256 * 128, OFF_B1 = OFF_B0 + RB0 * 128;
  constexpr int STAGE = (256 + BN) * 128;
  PCM_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;        // waves w and w+4 share a SIMD: one from each M group
  int nwg = g.tiles_m * g.tiles_n, bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = bid % g.tiles_n, tile_m = bid / g.tiles_n;
  const int m0 = tile_m * 256, n0 = tile_n * BN;

  // ---- loader geometry: every LDS-DMA instruction of a wave fills 8 rows x 128 B (lane -> row lane>>3, 16-B slot lane&7);
  // this wave owns groups wave + 8j of every piece, so its swizzled source chunk is the same for all its rows.
  const int lrow = lane >> 3;
  const unsigned csw16 = (unsigned)(((lane & 7) ^ ((4 * (wave & 1) + (lrow >> 1)) & 7)) << 4);
  const int HoWo = g.Ho * g.Wo;
  auto a_tile_row = [&](int h, int j) { int r = 8 * (wave + 8 * j) + lrow; return 128 * (r >> 6) + 64 * h + (r & 63); };
  auto b_tile_col = [&](int p, int j) {
    const int fp = p ? F1 : F0;
    int r = 8 * (wave + 8 * j) + lrow;
    return WNC * (r / (16 * fp)) + (p ? 16 * F0 : 0) + r % (16 * fp);
  };

  // ---- K iterators (wave-uniform).  A pieces walk (segment, tap, 64-channel chunk); B pieces walk (segment, K-tile).
  int total_kt = g.seg[0].ktiles + (g.nseg > 1 ? g.seg[1].ktiles : 0);
  int kt0 = 0;
  if (g.splitk > 1) {
    kt0 = blockIdx.y * g.kt_per_split;
    int kt1 = kt0 + g.kt_per_split; if (kt1 > total_kt) kt1 = total_kt;
    total_kt = kt1 - kt0;
  }
  const int T = total_kt;
  // the segment descriptors in use are held as local (SGPR) copies: no runtime-indexed kernarg access
  // K order of a 3x3 segment.  tap-outer (tap, chunk): the row offsets change once per tap, but a pixel's 9 shifted reads are a whole pass
  // over the channels apart -- with 32 tiles per XCD the hot set (32 x 256 pixels x C x 2 B) outgrows the 4 MB L2 and every tap
  // re-fetches the activation tile through the fabric (measured 4.65x the algorithmic bytes on the 320-channel 64x64 convs).
  // chunk-outer (chunk, tap), taken for the stride-1 direct view: the 9 taps of one 64-channel chunk are CONSECUTIVE K-tiles, so 8 of the
  // 9 reads hit L2.  Its cost is a new tap every K-tile: rows keep the CENTRE-tap offset and a 9-bit validity mask (once per segment),
  // the tap is one wave-uniform signed delta -- 3 VALU per row at issue time instead of the per-tap re-key.
  // Kernel variants (the launcher picks; each compiles one address path -- merged into one kernel the scalar state spilled 46 SGPRs).
  // SHIPPED: <F0, false, false> -- tap-outer order, per-tap re-key.  The two alternatives were built and MEASURED SLOWER on MI355X on all
  // 12 conv shapes of the bs-16 step (tools/gemm_conv_order_ab.py, profiles/r02_e_gemm8p_conv_variants_ab.txt); they stay as A/B options:
  //   MD  (PCM_GEMM_CONV_MD=1) "mask + delta" addressing for calls whose 3x3 segments are all stride 1 / direct: rows keep the centre-tap
  //       offset and the 9-bit mask, the tap is a scalar delta applied at issue time (3 VALU per row per piece) instead of the per-tap
  //       re-key: 0.88-1.02x (the issue-time VALU sits on the phase path; the re-key runs once per C/64 K-tiles);
  //   CO  (PCM_GEMM_CONV_CO=1, implies MD) chunk-outer K order: 4x less fabric traffic, 0.82-0.95x the speed (1.12x only at 8x8).
  auto seg_co = [&](const SegDev& s_) { return MD && s_.mode == PCM_SEG_CONV3X3; };
  int a_seg = 0, a_tap = 0, a_chunk = 0, a_nchunk, a_ntap, a_delta = 0;
  bool a_co;
  int b_seg[2], b_kt[2];
  SegDev ca = g.seg[0];
  {
    int k = kt0;
    if (k >= g.seg[0].ktiles) { k -= g.seg[0].ktiles; a_seg = 1; ca = g.seg[1]; }
    b_seg[0] = b_seg[1] = a_seg; b_kt[0] = b_kt[1] = k;
    a_nchunk = (ca.mode == PCM_SEG_CONV3X3 ? ca.C : ca.K) >> 6;
    a_ntap = ca.mode == PCM_SEG_CONV3X3 ? 9 : 1;
    a_co = seg_co(ca);
    if (CO && a_co) { a_chunk = k / 9; a_tap = k - 9 * a_chunk; } else { a_tap = k / a_nchunk; a_chunk = k - a_tap * a_nchunk; }
  }
  const bf16_t* b_w[2] = {ca.w, ca.w};
  int b_K[2] = {ca.K, ca.K}, b_nkt[2] = {ca.ktiles, ca.ktiles}, b_nch[2] = {a_nchunk, a_nchunk};
  bool b_co[2] = {CO && a_co, CO && a_co};
  int a_key[2][2];          // per A row: plain -> m, conv -> b<<20 | y<<10 | x ; -1 = row beyond M.  chunk-outer: the 9-bit tap validity mask
  unsigned a_voff[2][2];    // byte offset of the row's 16-B chunk for the current (segment, tap).  chunk-outer: for the CENTRE tap
  unsigned w_voff0[F0], w_voff1[F1];
  auto a_rekey = [&]() {
    const SegDev& cs = ca;
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        int m = m0 + a_tile_row(h, j);
        int key = m;
        if (cs.mode == PCM_SEG_CONV3X3) {
          int bb = m / HoWo, rem = m - bb * HoWo, y = rem / g.Wo, x = rem - y * g.Wo;
          key = (bb << 20) | (y << 10) | x;
          if (MD) {
            const int ym = (y > 0 ? 1 : 0) | 2 | (y < cs.Hs - 1 ? 4 : 0), xm = (x > 0 ? 1 : 0) | 2 | (x < cs.Ws - 1 ? 4 : 0);
            key = ((ym & 1) ? xm : 0) | (xm << 3) | ((ym & 4) ? xm << 6 : 0);
            a_voff[h][j] = (unsigned)((bb * cs.Hs + y) * cs.Ws + x) * (unsigned)(cs.C * 2) + csw16;
            if (m >= g.M) key = 0;
          }
        }
        a_key[h][j] = (m < g.M || a_co) ? key : -1;
      }
  };
  auto a_prepare_tap = [&]() {
    const SegDev& cs = ca;
    const int ty = a_tap / 3, tx = a_tap - ty * 3;
    if (a_co) { a_delta = ((ty - 1) * cs.Ws + (tx - 1)) * (cs.C * 2); return; }
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int key = a_key[h][j];
        bool ok = key >= 0;
        unsigned off;
        if (MD || cs.mode == PCM_SEG_PLAIN) {          // (MD variant: a segment that is not mask + delta addressed is a plain one)
          off = (unsigned)key * (unsigned)(cs.lda * 2);
        } else {
          int bb = key >> 20, y = (key >> 10) & 1023, x = key & 1023;
          int vy = y * cs.stride + ty - 1, vx = x * cs.stride + tx - 1;
          int sh = cs.src_mode != PCM_SRC_DIRECT;
          ok = ok && vy >= 0 && vy < (cs.Hs << sh) && vx >= 0 && vx < (cs.Ws << sh);
          if (cs.src_mode == PCM_SRC_ZEROINS2) ok = ok && !((vy | vx) & 1);
          off = (unsigned)(((bb * cs.Hs + (vy >> sh)) * cs.Ws + (vx >> sh))) * (unsigned)(cs.C * 2);
        }
        a_voff[h][j] = ok ? off + csw16 : PCM_OOB;
      }
  };
  auto w_prepare = [&](int p) {
    if (p == 0) {
#pragma unroll
      for (int j = 0; j < F0; j++) { int n = n0 + b_tile_col(0, j); w_voff0[j] = n < g.N ? (unsigned)n * (unsigned)(b_K[0] * 2) + csw16 : PCM_OOB; }
    } else {
#pragma unroll
      for (int j = 0; j < F1; j++) { int n = n0 + b_tile_col(1, j); w_voff1[j] = n < g.N ? (unsigned)n * (unsigned)(b_K[1] * 2) + csw16 : PCM_OOB; }
    }
  };
  auto a_advance = [&]() {
    if (CO && a_co) {
      a_tap++;
      if (a_tap == 9) {
        a_tap = 0; a_chunk++;
        if (a_chunk == a_nchunk) {
          a_chunk = 0; a_seg++;
          if (a_seg < g.nseg) {
            ca = g.seg[1];
            a_nchunk = (ca.mode == PCM_SEG_CONV3X3 ? ca.C : ca.K) >> 6;
            a_ntap = ca.mode == PCM_SEG_CONV3X3 ? 9 : 1;
            a_co = seg_co(ca);
            a_rekey();
          }
        }
      }
      if (a_seg < g.nseg) a_prepare_tap();
      return;
    }
    a_chunk++;
    if (a_chunk == a_nchunk) {
      a_chunk = 0; a_tap++;
      if (a_tap == a_ntap) {
        a_tap = 0; a_seg++;
        if (a_seg < g.nseg) {
          ca = g.seg[1];
          a_nchunk = (ca.mode == PCM_SEG_CONV3X3 ? ca.C : ca.K) >> 6;
          a_ntap = ca.mode == PCM_SEG_CONV3X3 ? 9 : 1;
          a_co = seg_co(ca);
          a_rekey();
        }
      }
      if (a_seg < g.nseg) a_prepare_tap();
    }
  };
  auto b_advance = [&](int p) {
    b_kt[p]++;
    if (b_kt[p] == b_nkt[p]) {
      b_kt[p] = 0; b_seg[p]++;
      if (b_seg[p] < g.nseg) {
        b_w[p] = g.seg[1].w; b_K[p] = g.seg[1].K; b_nkt[p] = g.seg[1].ktiles; b_co[p] = CO && seg_co(g.seg[1]); b_nch[p] = g.seg[1].C >> 6;
        w_prepare(p);
      }
    }
  };
  bool dma_on = true;
  auto issue_a = [&](int h, int stage) {
    if (PCM_ABL(8) && !dma_on) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ca.a, 0, PCM_OOB, PCM_RSRC_FLAGS);
    char* dst = smem + stage * STAGE + (h ? OFF_A1 : 0) + wave * 1024;
    const unsigned soff = (unsigned)(a_chunk * 128);   // the tap is in the row offsets; plain segments have one tap
#pragma unroll
    for (int j = 0; j < 2; j++) {
      unsigned v = a_voff[h][j];
      if (a_co) v = ((a_key[h][j] >> a_tap) & 1) ? v + (unsigned)a_delta : PCM_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, PCM_AS3(dst + j * 8192), 16, v, soff, 0, 0);
    }
  };
  auto issue_b = [&](int p, int stage) {
    if (PCM_ABL(8) && !dma_on) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)b_w[p], 0, PCM_OOB, PCM_RSRC_FLAGS);
    char* dst = smem + stage * STAGE + (p ? OFF_B1 : OFF_B0) + wave * 1024;
    int kk = b_kt[p];                                  // K-tile of the weight operand in ITS storage order (tap, chunk)
    if (b_co[p]) { const int c = kk / 9; kk = (kk - 9 * c) * b_nch[p] + c; }
    const unsigned soff = (unsigned)(kk * 128);
    if (p == 0) {
#pragma unroll
      for (int j = 0; j < F0; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, PCM_AS3(dst + j * 8192), 16, w_voff0[j], soff, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < F1; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, PCM_AS3(dst + j * 8192), 16, w_voff1[j], soff, 0, 0);
    }
  };

  f32x4 acc[8][FN];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int f = 0; f < FN; f++) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: all of K-tile 0, and A0 / B1 / A1 of K-tile 1 (its B0 is issued by phase 1 of tile 0)
  a_rekey(); a_prepare_tap(); w_prepare(0); w_prepare(1);
  issue_b(0, 0); if (1 < T) b_advance(0);
  issue_a(0, 0); issue_b(1, 0); issue_a(1, 0);
  if (1 < T) {
    b_advance(1); a_advance();
    issue_a(0, 1); issue_b(1, 1); issue_a(1, 1);
    if (2 < T) { b_advance(1); a_advance(); }
    PCM_WAIT_VMCNT(6);
  } else {
    PCM_WAIT_VMCNT(0);
  }
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one half-phase behind group 0

  // fragment reads: lane (frow = lane&15, fk = lane>>4) reads row rowbase+frow, logical chunk 4*kh+fk.  Every rowbase is a
  // multiple of 16, so the swizzle key ((row>>1)&7) depends on frow only: two per-lane offsets, the rest is uniform.
  const int frow = lane & 15, fk = lane >> 4;
  const int foff0 = frow * 128 + (((fk) ^ ((frow >> 1) & 7)) << 4), foff1 = frow * 128 + (((4 + fk) ^ ((frow >> 1) & 7)) << 4);
  bf16x8 af[4][2], b0f[F0][2], b1f[F1][2];
  auto read_a = [&](const char* base) {   // base = stage + region of the A half
    const char* p = base + (64 * wm) * 128;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      af[i][0] = *(const bf16x8*)(p + i * 2048 + foff0);
      af[i][1] = *(const bf16x8*)(p + i * 2048 + foff1);
    }
  };
  auto read_b0 = [&](const char* base) {
    const char* p = base + (16 * F0 * wn) * 128;
#pragma unroll
    for (int f = 0; f < F0; f++) {
      b0f[f][0] = *(const bf16x8*)(p + f * 2048 + foff0);
      b0f[f][1] = *(const bf16x8*)(p + f * 2048 + foff1);
    }
  };
  auto read_b1 = [&](const char* base) {
    const char* p = base + (16 * F1 * wn) * 128;
#pragma unroll
    for (int f = 0; f < F1; f++) {
      b1f[f][0] = *(const bf16x8*)(p + f * 2048 + foff0);
      b1f[f][1] = *(const bf16x8*)(p + f * 2048 + foff1);
    }
  };
#define PCM_PHASE_SYNC_IN()                  \
  PCM_WAIT_LGKMCNT0();                       \
  __builtin_amdgcn_sched_barrier(0);         \
  __builtin_amdgcn_s_barrier();              \
  __builtin_amdgcn_sched_barrier(0);         \
  __builtin_amdgcn_s_setprio(1)
#define PCM_PHASE_SYNC_OUT()                 \
  __builtin_amdgcn_s_setprio(0);             \
  __builtin_amdgcn_sched_barrier(0);         \
  __builtin_amdgcn_s_barrier();              \
  __builtin_amdgcn_sched_barrier(0)

  dma_on = false;
#ifdef PCM_ABLATE
  const bool stamp_on = blockIdx.x == gridDim.x / 2;
#endif
  for (int t = 0; t < T; t++) {
    const char* cur = smem + (t & 1) * STAGE;
    const bool more1 = t + 1 < T, more2 = t + 2 < T;
    // ---- phase 1: C00 = A0 x B0; stage B0 of tile t+1 (its region in the other buffer was last read in phase 4 of t-1)
    G8_STAMP(0);
    read_a(cur); read_b0(cur + OFF_B0);
    if (more1) { issue_b(0, (t + 1) & 1); if (t + 2 < T) b_advance(0); }
    PCM_PHASE_SYNC_IN();
    G8_STAMP(1);
    if (!PCM_ABL(4))
#pragma unroll
    for (int kh = 0; kh < 2; kh++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int f = 0; f < F0; f++) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[f][kh], af[i][kh], acc[i][f], 0, 0, 0);
    PCM_PHASE_SYNC_OUT();
    G8_STAMP(2);
    // ---- phase 2: C01 = A0 x B1; stage A0 of tile t+2 over the A0 just consumed
    read_b1(cur + OFF_B1);
    if (more2) issue_a(0, t & 1);
    PCM_PHASE_SYNC_IN();
    G8_STAMP(3);
    if (!PCM_ABL(4))
#pragma unroll
    for (int kh = 0; kh < 2; kh++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int f = 0; f < F1; f++) acc[i][F0 + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[f][kh], af[i][kh], acc[i][F0 + f], 0, 0, 0);
    PCM_PHASE_SYNC_OUT();
    G8_STAMP(4);
    // ---- phase 3: C11 = A1 x B1; stage B1 of tile t+2
    read_a(cur + OFF_A1);
    G8_STAMP(10);
    if (more2) { issue_b(1, t & 1); if (t + 3 < T) b_advance(1); }
    G8_STAMP(11);
    PCM_PHASE_SYNC_IN();
    G8_STAMP(5);
    if (!PCM_ABL(4))
#pragma unroll
    for (int kh = 0; kh < 2; kh++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int f = 0; f < F1; f++) acc[4 + i][F0 + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[f][kh], af[i][kh], acc[4 + i][F0 + f], 0, 0, 0);
    PCM_PHASE_SYNC_OUT();
    G8_STAMP(6);
    // ---- phase 4: C10 = A1 x B0 (B0 fragments re-read); stage A1 of tile t+2; retire everything tile t+1 needs
    read_b0(cur + OFF_B0);
    if (more2) {
      issue_a(1, t & 1);
      if (t + 3 < T) a_advance();
      PCM_WAIT_VMCNT(6);
    } else {
      PCM_WAIT_VMCNT(0);
    }
    G8_STAMP(7);
    PCM_PHASE_SYNC_IN();
    G8_STAMP(8);
    if (!PCM_ABL(4))
#pragma unroll
    for (int kh = 0; kh < 2; kh++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int f = 0; f < F0; f++) acc[4 + i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[f][kh], af[i][kh], acc[4 + i][f], 0, 0, 0);
    PCM_PHASE_SYNC_OUT();
    G8_STAMP(9);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // realign the two groups

  if (PCM_ABL(2)) { if (g.alpha != 123456.f) return; }
  // ---- epilogue.  lane owns pixel row (lane&15) of a fragment and 4 consecutive channels 4*(lane>>4)+r
  if (g.splitk > 1) {
    float* slab = g.ws + (size_t)blockIdx.y * g.M * g.N;
#pragma unroll
    for (int i8 = 0; i8 < 8; i8++) {
      const int m = m0 + 128 * wm + 16 * i8 + frow;
      if (m >= g.M) continue;
#pragma unroll
      for (int f = 0; f < FN; f++) {
        const int n = n0 + WNC * wn + 16 * f + 4 * fk;
        if (n < g.N) *(float4*)(slab + (size_t)m * g.N + n) = make_float4(acc[i8][f][0], acc[i8][f][1], acc[i8][f][2], acc[i8][f][3]);
      }
    }
    return;
  }
  // four passes of 64 rows x BN fp32 through LDS (the K-loop buffers are dead), shared with gemm4w.hip: gemm_epilogue.h
  PcmEpi<FN, 2>::run(g, smem, acc, tid, wm, wn, m0, n0, true);
#endif
}

size_t pcm_gemm8p_lds_bytes(int fn) { return 2 * (size_t)(256 + 64 * fn) * 128; }

PCM_TOOLS_ONLY(static int g_last_8p_variant = -1;      // of the last launch: bit 0 = mask + delta addressing, bit 1 = chunk-outer K order (tests read it)
               extern "C" int pcm_debug_last_gemm8p_variant() { return g_last_8p_variant; })

template <int F0, bool MD, bool CO>
static int launch8p(const GemmDev& g, void* stream) {
  PCM_TOOLS_ONLY(g_last_8p_variant = (MD ? 1 : 0) | (CO ? 2 : 0);)
  const size_t smem = pcm_gemm8p_lds_bytes(F0 + 2);
  dim3 grid(g.tiles_m * g.tiles_n, g.splitk);
  static bool lds_ok = false;
  if (!lds_ok) {
    hipError_t er = hipFuncSetAttribute((const void*)pcm_gemm8p_kernel<F0, MD, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    PCM_CHECK(er == hipSuccess, PCM_EHIP, "pcm_gemm_bf16: hipFuncSetAttribute(LDS %zu): %s", smem, hipGetErrorString(er));
    lds_ok = true;
  }
  PCM_LAUNCH((pcm_gemm8p_kernel<F0, MD, CO>), grid, dim3(512), smem, stream, g);
  return 0;
}
int pcm_gemm8p_launch(const GemmDev& g, int fn, void* stream) {
#if !PCM_HAS_TOOLS
  // product build: ONE conv addressing variant -- tap-outer K order with the per-tap re-key.  The mask + delta (MD) and chunk-outer (CO)
  // instantiations measured slower on every step shape in rounds 2, 3 and 4; they live in the tools build for A/B only.
  return fn == 5 ? launch8p<3, false, false>(g, stream) : launch8p<2, false, false>(g, stream);
#else
  // mask + delta variant: the call has a 3x3 segment and every 3x3 segment is the stride-1 direct view (g.conv_md: A/B hook, default on)
  bool md = false;
  if (g.conv_md) {
    for (int i = 0; i < g.nseg; i++)
      if (g.seg[i].mode == PCM_SEG_CONV3X3) md = true;
    for (int i = 0; i < g.nseg; i++)
      if (g.seg[i].mode == PCM_SEG_CONV3X3 && (g.seg[i].stride != 1 || g.seg[i].src_mode != PCM_SRC_DIRECT)) md = false;
  }
  // by shape (opt-in, conv order hook / env = 2): chunk-outer only on 8x8 feature maps, where a tile spans four images and the tap-outer
  // re-key runs every K-tile anyway (x1.12 on that launch in isolation, profiles/r02_e_*; no gain on the whole step, profiles/r03_k_*)
  if (!g.conv_md && g.conv_auto) {
    bool small = false, all_direct = true;
    for (int i = 0; i < g.nseg; i++)
      if (g.seg[i].mode == PCM_SEG_CONV3X3) {
        if (g.seg[i].stride != 1 || g.seg[i].src_mode != PCM_SRC_DIRECT) all_direct = false;
        else if (g.seg[i].Hs * g.seg[i].Ws <= 64) small = true;
      }
    if (small && all_direct) return fn == 5 ? launch8p<3, true, true>(g, stream) : launch8p<2, true, true>(g, stream);
  }
  const bool co = md && g.conv_co;
  if (fn == 5) return co ? launch8p<3, true, true>(g, stream) : (md ? launch8p<3, true, false>(g, stream) : launch8p<3, false, false>(g, stream));
  return co ? launch8p<2, true, true>(g, stream) : (md ? launch8p<2, true, false>(g, stream) : launch8p<2, false, false>(g, stream));
#endif
}
