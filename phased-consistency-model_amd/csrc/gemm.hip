// pcm_gemm_bf16 — the contraction workhorse of the UNet hot path on gfx950.
//
//   out[m][n] = act(alpha * sum_seg sum_k A_seg[m][k] * W_seg[n][k] + bias[n] + rowvec[m/rpb][n]) + res[m][n]
//
// One kernel serves Linear, conv1x1, conv3x3 (implicit im2col over channels-last activations,
// stride 1/2, nearest-2x-upsampled or zero-inserted sources) and the LoRA injection: the low-rank
// branch s*B(A(x)) is a SECOND K-segment accumulated into the same fp32 MFMA accumulators
// (K' = K + r), so base + LoRA are rounded to bf16 once.  The same kernel runs dgrad with
// pre-transposed / tap-flipped weight operands (weights are frozen, so both orientations stay
// resident in HBM).
//
// Structure (CDNA4): 256 threads = 4 waves (2x2), block tile (64*TM) x (64*TN), BK = 64;
// tiles are staged HBM -> LDS by LDS-DMA (global_load_lds, 16 B/lane, two stages); the LDS image
// is lane-linear, so the bank-conflict swizzle (16-B chunk ^ ((row>>1)&7)) is applied to the
// per-lane SOURCE address and to the ds_read_b128 fragment reads.  Out-of-range rows / padding
// taps / K tails fetch a 16-byte zero page.  MFMA: v_mfma_f32_32x32x16_bf16 with the WEIGHT tile
// as the A operand, so every lane owns 4 consecutive output channels per accumulator quad
// (8-byte bf16 stores, channel-contiguous).
#include <stdlib.h>

#include "gemm_dev.h"

__device__ __attribute__((aligned(16))) static const uint4 pcm_zero_page = {0u, 0u, 0u, 0u};

// NW waves as WGM x (NW/WGM); each wave owns a (32*TM) x (32*TN) sub-tile: block tile BM = 32*TM*WGM, BN = 32*TN*(NW/WGM)
template <int NW, int WGM, int TM, int TN>
__global__ __launch_bounds__(64 * NW) void pcm_gemm_kernel(GemmDev g) {
  constexpr int WGN = NW / WGM;
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  constexpr int STAGE = (BM + BN) * 128;
  PCM_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases stay in SGPRs
  const int wm = wave % WGM, wn = wave / WGM;
  // XCD-aware bijective remap: consecutive logical tiles stay on one XCD (private L2)
  int nwg = g.tiles_m * g.tiles_n, bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = bid % g.tiles_n, tile_m = bid / g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- loader state.  Each thread owns AI rows of the activation tile and WI rows of the weight tile
  // (fixed over the K loop).  Source pointers are rebuilt once per (segment, conv tap) and then only
  // advanced by 128 B per K-tile, so the steady-state issue is {select, add, LDS-DMA} per row.
  constexpr int AI = BM / (8 * NW), WI = BN / (8 * NW);
  const int lrow = lane >> 3, lchunk = lane & 7;
  const char* zero = (const char*)&pcm_zero_page;
  int a_m[AI], a_b[AI], a_y[AI], a_x[AI], a_c[AI];
  const bool any_conv = g.seg[0].mode == PCM_SEG_CONV3X3 || (g.nseg > 1 && g.seg[1].mode == PCM_SEG_CONV3X3);
  const int HoWo = g.Ho * g.Wo;
#pragma unroll
  for (int j = 0; j < AI; j++) {
    int row = 8 * (wave + NW * j) + lrow;
    a_c[j] = lchunk ^ ((row >> 1) & 7);
    int m = m0 + row;
    a_m[j] = m < g.M ? m : -1;
    a_b[j] = 0; a_y[j] = 0; a_x[j] = 0;
    if (any_conv && m < g.M) {
      int bb = m / HoWo, rem = m - bb * HoWo;
      a_b[j] = bb; a_y[j] = rem / g.Wo; a_x[j] = rem - a_y[j] * g.Wo;
    }
  }
  int w_n[WI], w_c[WI];
#pragma unroll
  for (int j = 0; j < WI; j++) {
    int row = 8 * (wave + NW * j) + lrow;
    w_c[j] = lchunk ^ ((row >> 1) & 7);
    int n = n0 + row;
    w_n[j] = n < g.N ? n : -1;
  }
  const char* a_cur[AI];
  const char* w_cur[WI];
  int a_inc[AI], w_inc[WI];   // bytes to advance per K-tile (0 for rows that read the zero page)
  // iterator (wave-uniform)
  int seg_i = 0, tap = 0, chunk = 0;
  SegDev cs = g.seg[0];
  int nchunk = ((cs.mode == PCM_SEG_CONV3X3 ? cs.C : cs.K) + 63) >> 6;
  int ntap = cs.mode == PCM_SEG_CONV3X3 ? 9 : 1;

  auto prepare_tap = [&]() {
    const int ty = tap / 3, tx = tap - ty * 3;
#pragma unroll
    for (int j = 0; j < AI; j++) {
      const char* p = zero;
      bool ok = a_m[j] >= 0;
      if (cs.mode == PCM_SEG_PLAIN) {
        p = (const char*)(cs.a + (size_t)(ok ? a_m[j] : 0) * cs.lda + 8 * a_c[j]);
      } else {
        int vy = a_y[j] * cs.stride + ty - 1, vx = a_x[j] * cs.stride + tx - 1;
        int sh = cs.src_mode != PCM_SRC_DIRECT;
        ok = ok && vy >= 0 && vy < (cs.Hs << sh) && vx >= 0 && vx < (cs.Ws << sh);
        if (cs.src_mode == PCM_SRC_ZEROINS2) ok = ok && !((vy | vx) & 1);
        int sy = ok ? (vy >> sh) : 0, sx = ok ? (vx >> sh) : 0;
        p = (const char*)(cs.a + ((size_t)(a_b[j] * cs.Hs + sy) * cs.Ws + sx) * cs.C + 8 * a_c[j]);
      }
      a_cur[j] = ok ? p : zero;
      a_inc[j] = ok ? 128 : 0;
    }
#pragma unroll
    for (int j = 0; j < WI; j++) {
      bool ok = w_n[j] >= 0;
      const char* p = (const char*)(cs.w + (size_t)(ok ? w_n[j] : 0) * cs.K + (size_t)tap * cs.C + 8 * w_c[j]);
      w_cur[j] = ok ? p : zero;
      w_inc[j] = ok ? 128 : 0;
    }
  };
  auto issue = [&](int stage) {
    char* base = smem + stage * STAGE;
    // K tail of a plain segment (K % 64 != 0): lanes beyond K read zeros
    const bool tail = cs.mode == PCM_SEG_PLAIN && (chunk + 1) * 64 > cs.K;
    char* wbase = base + BM * 128;
    if (!tail) {   // steady state: one LDS-DMA + one 64-bit add per row
#pragma unroll
      for (int j = 0; j < AI; j++) {
        __builtin_amdgcn_global_load_lds(PCM_AS1(a_cur[j]), PCM_AS3(base + (wave + NW * j) * 1024), 16, 0, 0);
        a_cur[j] += a_inc[j];
      }
#pragma unroll
      for (int j = 0; j < WI; j++) {
        __builtin_amdgcn_global_load_lds(PCM_AS1(w_cur[j]), PCM_AS3(wbase + (wave + NW * j) * 1024), 16, 0, 0);
        w_cur[j] += w_inc[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < AI; j++) {
        const char* src = chunk * 64 + 8 * a_c[j] >= cs.K ? zero : a_cur[j];
        __builtin_amdgcn_global_load_lds(PCM_AS1(src), PCM_AS3(base + (wave + NW * j) * 1024), 16, 0, 0);
        a_cur[j] += a_inc[j];
      }
#pragma unroll
      for (int j = 0; j < WI; j++) {
        const char* src = chunk * 64 + 8 * w_c[j] >= cs.K ? zero : w_cur[j];
        __builtin_amdgcn_global_load_lds(PCM_AS1(src), PCM_AS3(wbase + (wave + NW * j) * 1024), 16, 0, 0);
        w_cur[j] += w_inc[j];
      }
    }
    // advance the iterator
    chunk++;
    if (chunk == nchunk) {
      chunk = 0; tap++;
      if (tap == ntap) {
        tap = 0; seg_i++;
        if (seg_i < g.nseg) {
          cs = g.seg[1];
          nchunk = ((cs.mode == PCM_SEG_CONV3X3 ? cs.C : cs.K) + 63) >> 6;
          ntap = cs.mode == PCM_SEG_CONV3X3 ? 9 : 1;
        }
      }
      if (seg_i < g.nseg) prepare_tap();
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; i++)
#pragma unroll
    for (int j = 0; j < TM; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  int total_kt = g.seg[0].ktiles + (g.nseg > 1 ? g.seg[1].ktiles : 0);
  if (g.splitk > 1) {  // this block's K slice: position the iterator at its first K-tile
    int kt0 = blockIdx.y * g.kt_per_split;
    int kt1 = kt0 + g.kt_per_split; if (kt1 > total_kt) kt1 = total_kt;
    total_kt = kt1 - kt0;
    if (kt0 >= g.seg[0].ktiles) {
      kt0 -= g.seg[0].ktiles; seg_i = 1; cs = g.seg[1];
      nchunk = ((cs.mode == PCM_SEG_CONV3X3 ? cs.C : cs.K) + 63) >> 6;
      ntap = cs.mode == PCM_SEG_CONV3X3 ? 9 : 1;
    }
    tap = kt0 / nchunk; chunk = kt0 - tap * nchunk;
  }
  prepare_tap();
#pragma unroll
  for (int j = 0; j < AI; j++) a_cur[j] += (size_t)a_inc[j] * chunk;
#pragma unroll
  for (int j = 0; j < WI; j++) w_cur[j] += (size_t)w_inc[j] * chunk;
  issue(0);
  const int frow = lane & 31, hi = lane >> 5;
  for (int kt = 0; kt < total_kt; kt++) {
    __syncthreads();  // tile kt landed (hipcc drains the LDS-DMA queue before the barrier); stage (kt+1)&1 free
    if (kt + 1 < total_kt) issue((kt + 1) & 1);
    const char* At = smem + (kt & 1) * STAGE;
    const char* Wt = At + BM * 128;
    // fragment reads are software-pipelined one 16-wide sub-step ahead of the MFMAs
    bf16x8 wf[2][TN], af[2][TM];
#pragma unroll
    for (int i = 0; i < TN; i++) wf[0][i] = *(const bf16x8*)(Wt + lds_off(wn * 32 * TN + i * 32 + frow, hi));
#pragma unroll
    for (int j = 0; j < TM; j++) af[0][j] = *(const bf16x8*)(At + lds_off(wm * 32 * TM + j * 32 + frow, hi));
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < TN; i++) wf[nxt][i] = *(const bf16x8*)(Wt + lds_off(wn * 32 * TN + i * 32 + frow, 2 * (ks + 1) + hi));
#pragma unroll
        for (int j = 0; j < TM; j++) af[nxt][j] = *(const bf16x8*)(At + lds_off(wm * 32 * TM + j * 32 + frow, 2 * (ks + 1) + hi));
      }
#pragma unroll
      for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][i], af[cur][j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns column m, 4 consecutive channels n per accumulator quad ----
  if (g.splitk > 1) {  // raw partial sums to this slice's slab; pcm_gemm_finalize_kernel applies the epilogue
    float* slab = g.ws + (size_t)blockIdx.y * g.M * g.N;
#pragma unroll
    for (int j = 0; j < TM; j++) {
      int m = m0 + wm * 32 * TM + j * 32 + frow;
      if (m >= g.M) continue;
#pragma unroll
      for (int i = 0; i < TN; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          int n = n0 + wn * 32 * TN + i * 32 + 8 * q + 4 * hi;
          if (n < g.N) *(float4*)(slab + (size_t)m * g.N + n) = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        }
    }
    return;
  }
  // bf16 output: stage the fp32 tile through LDS (the K-loop stages are dead now) so that the global side is
  // whole-row 16-byte accesses: residual / rowvec loads and the output stores of a wave cover contiguous
  // 256-B row segments instead of 8-B pieces of 32 different rows.
  if (!g.out_f32 && (g.N % 8) == 0 && (g.ldo % 8) == 0 && (!g.res || ((g.ldr % 8) == 0 && (((uintptr_t)g.res) & 15) == 0)) &&
      (!g.rowvec || (((uintptr_t)g.rowvec) & 15) == 0)) {
    constexpr int CH = BN / 4;   // 16-B fp32 chunks per tile row; chunk index XOR-swizzled with the row
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TM; j++) {
      const int row = wm * 32 * TM + j * 32 + frow;
#pragma unroll
      for (int i = 0; i < TN; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int ch = (wn * 32 * TN + i * 32 + 8 * q + 4 * hi) >> 2;
          *(float4*)(smem + ((size_t)row * CH + (ch ^ (row & (CH - 1)))) * 16) =
              make_float4(acc[i][j][4 * q] * g.alpha, acc[i][j][4 * q + 1] * g.alpha, acc[i][j][4 * q + 2] * g.alpha, acc[i][j][4 * q + 3] * g.alpha);
        }
    }
    __syncthreads();
    constexpr int C8 = BN / 8;   // 16-B bf16 output chunks per tile row
    for (int idx = tid; idx < BM * C8; idx += 64 * NW) {
      const int row = idx / C8, c8 = idx - row * C8;
      const int m = m0 + row, n = n0 + 8 * c8;
      if (m >= g.M || n >= g.N) continue;
      const float4 lo = *(const float4*)(smem + ((size_t)row * CH + ((2 * c8) ^ (row & (CH - 1)))) * 16);
      const float4 hi4 = *(const float4*)(smem + ((size_t)row * CH + ((2 * c8 + 1) ^ (row & (CH - 1)))) * 16);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
      if (g.bias) {
        const float4 b0 = *(const float4*)(g.bias + n), b1 = *(const float4*)(g.bias + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (g.rowvec) {
        const uint4 t = *(const uint4*)(g.rowvec + (size_t)(m / g.rpb) * g.N + n);
        const unsigned tw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; e++) { v[2 * e] += bf2f((bf16_t)(tw[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(tw[e] >> 16)); }
      }
      if (g.act == PCM_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = silu_f(v[e]);
      }
      if (g.res) {
        const uint4 t = *(const uint4*)(g.res + (size_t)m * g.ldr + n);
        const unsigned tw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; e++) { v[2 * e] += bf2f((bf16_t)(tw[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(tw[e] >> 16)); }
      }
      const uint4 o = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
      *(uint4*)((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
      if (g.out2) *(uint4*)(g.out2 + (size_t)m * g.ldo2 + n) = o;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TM; j++) {
    int m = m0 + wm * 32 * TM + j * 32 + frow;
    if (m >= g.M) continue;
    const bf16_t* rv = g.rowvec ? g.rowvec + (size_t)(m / g.rpb) * g.N : nullptr;
#pragma unroll
    for (int i = 0; i < TN; i++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        int n = n0 + wn * 32 * TN + i * 32 + 8 * q + 4 * hi;
        if (n >= g.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * q + e] * g.alpha;
        if (g.bias) {
          float4 b4 = *(const float4*)(g.bias + n);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
        if (rv) {
          uint2 t = *(const uint2*)(rv + n);
          v[0] += bf2f((bf16_t)(t.x & 0xffff)); v[1] += bf2f((bf16_t)(t.x >> 16));
          v[2] += bf2f((bf16_t)(t.y & 0xffff)); v[3] += bf2f((bf16_t)(t.y >> 16));
        }
        if (g.act == PCM_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = silu_f(v[e]);
        }
        if (g.res) {
          uint2 t = *(const uint2*)(g.res + (size_t)m * g.ldr + n);
          v[0] += bf2f((bf16_t)(t.x & 0xffff)); v[1] += bf2f((bf16_t)(t.x >> 16));
          v[2] += bf2f((bf16_t)(t.y & 0xffff)); v[3] += bf2f((bf16_t)(t.y >> 16));
        }
        if (g.out_f32) {
          *(float4*)((float*)g.out + (size_t)m * g.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          const uint2 o = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
          *(uint2*)((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
          if (g.out2) *(uint2*)(g.out2 + (size_t)m * g.ldo2 + n) = o;
        }
      }
    }
  }
}

// split-K finalize: out = act(alpha * sum_slices ws + bias + rowvec) + residual
__global__ __launch_bounds__(256) void pcm_gemm_finalize_kernel(GemmDev g) {
  const long nq = (long)g.M * (g.N / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (long)gridDim.x * blockDim.x) {
    int m = (int)(i / (g.N / 4)), n = (int)(i % (g.N / 4)) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < g.splitk; s++) {
      float4 p = *(const float4*)(g.ws + ((size_t)s * g.M + m) * g.N + n);
      v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] *= g.alpha;
    if (g.bias) {
      float4 b4 = *(const float4*)(g.bias + n);
      v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    }
    if (g.rowvec) {
      uint2 t = *(const uint2*)(g.rowvec + (size_t)(m / g.rpb) * g.N + n);
      v[0] += bf2f((bf16_t)(t.x & 0xffff)); v[1] += bf2f((bf16_t)(t.x >> 16));
      v[2] += bf2f((bf16_t)(t.y & 0xffff)); v[3] += bf2f((bf16_t)(t.y >> 16));
    }
    if (g.act == PCM_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = silu_f(v[e]);
    }
    if (g.res) {
      uint2 t = *(const uint2*)(g.res + (size_t)m * g.ldr + n);
      v[0] += bf2f((bf16_t)(t.x & 0xffff)); v[1] += bf2f((bf16_t)(t.x >> 16));
      v[2] += bf2f((bf16_t)(t.y & 0xffff)); v[3] += bf2f((bf16_t)(t.y >> 16));
    }
    if (g.out_f32) *(float4*)((float*)g.out + (size_t)m * g.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
    else {
      const uint2 o = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
      *(uint2*)((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
      if (g.out2) *(uint2*)(g.out2 + (size_t)m * g.ldo2 + n) = o;
    }
  }
}

// the same finalize for an output that a GroupNorm reads next (abi 5, pcm_gemm_epi.chstats): a block owns `rpb` consecutive rows of ONE sample
// (rpb divides stats_rows); a thread keeps a fixed group of 4 channels and walks the block's rows in steps of the row lanes (blockDim.x / (N/4),
// four rows' slab loads in flight), so the per-channel sums of the STORED values stay in registers; the row lanes meet in LDS and the block
// issues N x 2 fp64 atomics -- the statistics pass over the tensor is not needed.
__global__ __launch_bounds__(320) void pcm_gemm_finalize_stats_kernel(GemmDev g, int rpb, int nq_per_block) {
  __shared__ float red[320][8];                    // [row lane * nq_per_block + channel quad of the block][4 sums, 4 sums of squares]
  const int NQ = g.N / 4;
  const int q0 = blockIdx.y * nq_per_block;
  int nqb = NQ - q0; if (nqb > nq_per_block) nqb = nq_per_block;
  const int RL = blockDim.x / nq_per_block;        // row lanes (>= 1, <= 8)
  const int ql = threadIdx.x % nq_per_block, rl = threadIdx.x / nq_per_block;
  const int r0 = blockIdx.x * rpb;
  int r1 = r0 + rpb; if (r1 > g.M) r1 = g.M;
  const bool on = ql < nqb && rl < RL;
  const int n = 4 * (q0 + (on ? ql : 0));
  float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  if (on) {
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) b4 = *(const float4*)(g.bias + n);
    constexpr int U = 4;
    for (int mb = r0 + rl; mb < r1; mb += U * RL) {
      float4 acc[U];
      uint2 rv[U], rs[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        int m = mb + u * RL; if (m > r1 - 1) m = r1 - 1;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s_ = 0; s_ < g.splitk; s_++) {
          const float4 p = *(const float4*)(g.ws + ((size_t)s_ * g.M + m) * g.N + n);
          a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
        }
        acc[u] = a;
        rv[u] = g.rowvec ? *(const uint2*)(g.rowvec + (size_t)(m / g.rpb) * g.N + n) : make_uint2(0u, 0u);
        rs[u] = g.res ? *(const uint2*)(g.res + (size_t)m * g.ldr + n) : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int m = mb + u * RL;
        if (m >= r1) continue;
        float v[4] = {acc[u].x * g.alpha + b4.x, acc[u].y * g.alpha + b4.y, acc[u].z * g.alpha + b4.z, acc[u].w * g.alpha + b4.w};
        if (g.rowvec) {
          v[0] += bf2f((bf16_t)(rv[u].x & 0xffff)); v[1] += bf2f((bf16_t)(rv[u].x >> 16));
          v[2] += bf2f((bf16_t)(rv[u].y & 0xffff)); v[3] += bf2f((bf16_t)(rv[u].y >> 16));
        }
        if (g.act == PCM_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = silu_f(v[e]);
        }
        if (g.res) {
          v[0] += bf2f((bf16_t)(rs[u].x & 0xffff)); v[1] += bf2f((bf16_t)(rs[u].x >> 16));
          v[2] += bf2f((bf16_t)(rs[u].y & 0xffff)); v[3] += bf2f((bf16_t)(rs[u].y >> 16));
        }
        const uint2 o = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        *(uint2*)((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
        if (g.out2) *(uint2*)(g.out2 + (size_t)m * g.ldo2 + n) = o;
        const float x[4] = {bf2f((bf16_t)(o.x & 0xffff)), bf2f((bf16_t)(o.x >> 16)), bf2f((bf16_t)(o.y & 0xffff)), bf2f((bf16_t)(o.y >> 16))};
#pragma unroll
        for (int e = 0; e < 4; e++) { sm[e] += x[e]; sq[e] = fmaf(x[e], x[e], sq[e]); }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) { red[threadIdx.x][e] = sm[e]; red[threadIdx.x][4 + e] = sq[e]; }
  }
  __syncthreads();
  // channel quad ql, value e (0..7: 4 sums, 4 sums of squares): one fp64 atomic per (channel, statistic) and block
  for (int i = threadIdx.x; i < nqb * 8; i += blockDim.x) {
    const int qq = i >> 3, e = i & 7;
    float t = 0.f;
    for (int l = 0; l < RL; l++) t += red[l * nq_per_block + qq][e];
    double* dst = g.chstats + ((size_t)(r0 / g.stats_rows) * g.N + 4 * (q0 + qq) + (e & 3)) * 2 + (e >> 2);
    atomicAdd(dst, (double)t);
  }
}
// geometry of the kernel above: rows per block (divides stats_rows; >= ~256 blocks where M allows), channel quads per block (<= 320), threads
static void finalize_stats_geometry(int M, int N, int stats_rows, int* rpb_, int* nqpb_, int* threads_, int* ny_) {
  const int NQ = N / 4;
  int ny = (NQ + 319) / 320;
  const int nqpb = (NQ + ny - 1) / ny;
  int rl = 320 / nqpb; if (rl > 8) rl = 8; if (rl < 1) rl = 1;
  int rpb = 64;
  while (rpb > 8 && ((stats_rows % rpb) || (long)((M + rpb - 1) / rpb) * ny < 256)) rpb >>= 1;
  *rpb_ = rpb; *nqpb_ = nqpb; *threads_ = nqpb * rl; *ny_ = ny;
}

// tile / split-K plan shared by pcm_gemm_bf16 and pcm_gemm_workspace_bytes
struct GemmPlan { int BM, BN, tiles_m, tiles_n, splitk, kt_per_split; size_t ws_bytes; int big_fn; int w4_fn; };
// Everything from here to gemm_plan is the TOOLS build's knob set (pcm_common.h): in the product build each is its shipped constant.
PCM_KNOB int g_force_bm = 0, g_force_bn = 0;
// -1: read PCM_GEMM_BIG once.  0 = 4-wave tiles of this file only, 1 = planner (default), 2 = gemm8p wherever eligible (no gemm4w),
// 3 = gemm4w wherever eligible (else as 2), 4 = planner without gemm4w; tuning / A-B only
PCM_LAZY_KNOB(big_mode, g_big_mode, "PCM_GEMM_BIG", 1)
PCM_TOOLS_ONLY(extern "C" void pcm_debug_gemm_big_mode(int mode) { g_big_mode = mode; })
PCM_KNOB int g_conv_co = -1, g_conv_md = -1;    // -1: PCM_GEMM_CONV_CO / PCM_GEMM_CONV_MD env if set, else 0 (the shipped tap-outer / re-key path)
PCM_TOOLS_ONLY(extern "C" void pcm_debug_gemm_conv_order(int chunk_outer) { g_conv_co = chunk_outer < 0 ? -1 : (chunk_outer > 1 ? 2 : (chunk_outer ? 1 : 0)); }   // -1: default; 2: by shape
               extern "C" void pcm_debug_gemm_conv_md(int mask_delta) { g_conv_md = mask_delta < 0 ? -1 : (mask_delta ? 1 : 0); })
PCM_KNOB int g_ablate = 0;      // see PCM_ABL in gemm_dev.h (-DPCM_ABLATE probe builds); bits 8.. = gemm4w start stagger override + 1
PCM_TOOLS_ONLY(extern "C" void pcm_debug_gemm_ablate(int mask) { g_ablate = mask; }
               static int g_last_plan = 0;   // the plan code (pcm_gemm_plan_code, include/pcm_hip.h) of the most recent pcm_gemm_bf16 launch
               extern "C" int pcm_debug_last_gemm_plan(void) { return g_last_plan; })
// A/B only: PCM_GEMM_PLAN_LEGACY=1 = the planner as it was before the round-4 re-fit (flat 15 % price of a K split, K split for every
// under-filled small-tile grid)
PCM_LAZY_KNOB(plan_legacy_knob, g_plan_legacy, "PCM_GEMM_PLAN_LEGACY", 0)
static bool plan_legacy() { return plan_legacy_knob() != 0; }
// tuning hook (tools/ only): force the block tile (bm | ksplit << 16, bn) of subsequent pcm_gemm_bf16 calls; (0,0) restores the planner
PCM_TOOLS_ONLY(extern "C" void pcm_debug_force_gemm_tile(int bm, int bn) { g_force_bm = bm; g_force_bn = bn; })

// big_ok: the call satisfies gemm8p's preconditions (K%64==0 per segment, bf16 output with 16-B rows, < 2 GiB operands)
// gemm4w.hip (two workgroups per CU, 128-row tiles).  Measured against gemm8p on every plain-segment launch of the bs-16 step
// (profiles/r04_d_gemm_ab_libs_*.txt): with the shared epilogue it wins only on the fused-GEGLU feed-forward projections with K <= 384
// (x0.91-0.93: their tile is epilogue-heavy and N = 8 x 320 gives 8192 small tiles to interleave); it loses 2-30 % elsewhere (twice the
// weight traffic per flop, and the HBM-bound projections are bound by bytes, not by the missing overlap).  The planner follows that;
// PCM_GEMM_4W_MAXKT (64-wide K-tiles incl. the LoRA segment, default 6) moves the boundary, PCM_GEMM_BIG=3 / 4 force it on / off.
PCM_LAZY_KNOB(w4_max_kt, g_w4_max_kt, "PCM_GEMM_4W_MAXKT", 6)
PCM_LAZY_KNOB(w4_stagger_default, g_w4_stagger, "PCM_GEMM_4W_STAGGER", 0)
static GemmPlan gemm_plan(int M, int N, int total_kt, bool allow_split, bool big_ok, bool must_big = false, bool w4_ok = false) {
  GemmPlan p;
  p.big_fn = 0; p.w4_fn = 0;
  if (must_big) allow_split = false;   // fused-GEGLU epilogue lives in gemm8p / gemm4w only, on complete sums
  if (w4_ok && big_ok && !g_force_bm && (big_mode() == 1 || big_mode() == 3) && M >= 128) {
    int best_fn = 0; double best = 0.0;
    for (int fn = 5; fn >= 4; fn--) {
      const int bn = 64 * fn;
      const long tm = (M + 127) / 128, tn = (N + bn - 1) / bn;
      const double useful = ((double)M * N) / ((double)tm * 128 * tn * bn);
      if (useful > best + 1e-9) { best = useful; best_fn = fn; }
    }
    const bool take = big_mode() == 3 ? true : (must_big && total_kt <= w4_max_kt() && best >= 0.8 && (long)((M + 127) / 128) * ((N + 64 * best_fn - 1) / (64 * best_fn)) >= 1024);
    if (best_fn && take) {
      p.w4_fn = best_fn; p.BM = 128; p.BN = 64 * best_fn;
      p.tiles_m = (M + 127) / 128; p.tiles_n = (N + p.BN - 1) / p.BN;
      p.splitk = 1; p.kt_per_split = total_kt; p.ws_bytes = 0;
      return p;
    }
  }
  if (big_ok && ((!g_force_bm && big_mode() > 0 && M >= 256) || must_big)) {
    // candidates 256x320 / 256x256 (+ split-K); pick by useful work per block-round of the 256 CUs
    int best_fn = 0, best_s = 1; double best = 0.0;
    for (int fn = 5; fn >= 4; fn--) {
      const int bn = 64 * fn;
      const long tm = (M + 255) / 256, tn = (N + bn - 1) / bn, tiles = tm * tn;
      const double useful = ((double)M * N) / ((double)tm * 256 * tn * bn);
      for (int s = 1; s <= 8; s++) {
        if (s > 1 && (!allow_split || total_kt / s < 12)) break;
        const long blocks = tiles * s, rounds = (blocks + 255) / 256;
        double eff = useful * (double)blocks / (double)(rounds * 256);
        if (s > 1) eff *= 0.85;                       // slab write + finalize pass
        if (eff > best + 1e-9) { best = eff; best_fn = fn; best_s = s; }
      }
    }
    const double need = (big_mode() == 2 || big_mode() == 3 || must_big) ? 0.0 : 0.62;
    if (best_fn && best >= need) {
      // WHICH phased-tile plan: a time model fitted to the plan sweep of round 4 (tools/gemm_small_m.py, profiles/r04_n_*): a block-round
      // costs 10 us + 1.4 us (256x256) / 1.75 us (256x320) per 64-wide K step; a K split adds the slab round trip (s fp32 slabs written
      // and read back at ~3.5 TB/s) and the finalize launch.  The occupancy score above valued a split at a flat 15 %: on
      // (16384, 1536, K 1536) it took 2 slabs (143 us) where no split runs 94 us, on (8192, 1536, K 6144) 4 slabs (182 us) against 134.
      double best_t = 1e30;
      for (int fn = 5; fn >= 4 && !plan_legacy(); fn--) {
        const int bn = 64 * fn;
        const long tiles = (long)((M + 255) / 256) * ((N + bn - 1) / bn);
        for (int sp = 1; sp <= 8; sp++) {
          if (sp > 1 && (!allow_split || total_kt / sp < 12)) break;
          const long rounds = (tiles * sp + 255) / 256;
          double t = (double)rounds * (((total_kt + sp - 1) / sp) * (fn == 5 ? 1.75 : 1.4) + 10.0);
          if (sp > 1) t += ((double)sp * M * N * 4.0 + (double)M * N * 2.0) / 3.5e6 + 3.0;
          if (t < best_t - 1e-9) { best_t = t; best_fn = fn; best_s = sp; }
        }
      }
      p.big_fn = best_fn; p.BM = 256; p.BN = 64 * best_fn;
      p.tiles_m = (M + 255) / 256; p.tiles_n = (N + p.BN - 1) / p.BN;
      p.kt_per_split = (total_kt + best_s - 1) / best_s;
      p.splitk = (total_kt + p.kt_per_split - 1) / p.kt_per_split;
      p.ws_bytes = p.splitk > 1 ? (size_t)p.splitk * M * N * sizeof(float) : 0;
      return p;
    }
  }
  if (g_force_bm && (g_force_bm & 0xffff) == 256 && big_ok && (g_force_bn == 256 || g_force_bn == 320)) {   // tuning: the phased tile, forced
    const int fs = (g_force_bm >> 16) > 1 && allow_split && !must_big ? (g_force_bm >> 16) : 1;
    p.big_fn = g_force_bn / 64; p.BM = 256; p.BN = g_force_bn;
    p.tiles_m = (M + 255) / 256; p.tiles_n = (N + p.BN - 1) / p.BN;
    p.kt_per_split = (total_kt + fs - 1) / fs;
    p.splitk = (total_kt + p.kt_per_split - 1) / p.kt_per_split;
    p.ws_bytes = p.splitk > 1 ? (size_t)p.splitk * M * N * sizeof(float) : 0;
    return p;
  }
  if (g_force_bm) {
    p.splitk = 1; p.kt_per_split = total_kt; p.ws_bytes = 0; p.BM = g_force_bm & 0xffff; p.BN = g_force_bn;
    p.tiles_m = (M + p.BM - 1) / p.BM; p.tiles_n = (N + p.BN - 1) / p.BN;
    const int fs = g_force_bm >> 16;          // tuning: bits 16.. of the forced BM = K split
    if (fs > 1 && allow_split && total_kt >= 2 * fs) {
      p.kt_per_split = (total_kt + fs - 1) / fs;
      p.splitk = (total_kt + p.kt_per_split - 1) / p.kt_per_split;
      p.ws_bytes = (size_t)p.splitk * M * N * sizeof(float);
    }
    return p;
  }
  p.splitk = 1; p.kt_per_split = total_kt; p.ws_bytes = 0;
  // 128x128 when N is a multiple of 128; otherwise the tall 256x64 tile (same 64x64 per-wave tile, no N waste at N=320/64)
  if ((N % 128) == 0) { p.BM = 128; p.BN = 128; } else { p.BM = 256; p.BN = 64; }
  auto ntiles = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  long tiles = ntiles(p.BM, p.BN);
  if (tiles < 256 && p.BM == 256) { p.BM = 128; tiles = ntiles(128, 64); }
  // short K (<= 1536 incl. the LoRA segment): a smaller tile and no K split beats the slab round trip + finalize launch, and the grid
  // should hold ~2 tiles per CU (tools/gemm_small_m.py, profiles/r04_n_small_m_plan_sweep.txt: (2048, 1280, K 1280) 128x128 / 3 slabs
  // 25.9 us, 64x64 16.3; (1024, 1280, 1280) 20.5 -> 14.4; (8192, 640, 640) 128x128 20.5, 128x64 16.9; (4096, 1280, 1280) 29.7 -> 27.1)
  const bool short_k = total_kt <= 24 && ntiles(64, 64) >= 256 && !plan_legacy();      // (grids that even 64x64 tiles do not fill keep the K split)
  if (!short_k && tiles < 256 && allow_split && total_kt >= 16) {
    // under-filled grid with a long K loop: slice K across blockIdx.y (slab reduction, no atomics)
    int s = (int)((384 + tiles - 1) / tiles);
    if (s > total_kt / 4) s = total_kt / 4;
    if (s > 16) s = 16;
    if (s >= 2) {
      p.kt_per_split = (total_kt + s - 1) / s;
      p.splitk = (total_kt + p.kt_per_split - 1) / p.kt_per_split;
      p.ws_bytes = (size_t)p.splitk * M * N * sizeof(float);
    }
  }
  if (p.splitk == 1) {  // otherwise shrink the tile until the grid fills the chip
    const long want = short_k ? 512 : 256;
    if (tiles < want && p.BN == 128) { p.BN = 64; tiles = ntiles(128, 64); }
    if (tiles < want && p.BM <= 128) { p.BM = 64; p.BN = 64; }
  }
  p.tiles_m = (M + p.BM - 1) / p.BM;
  p.tiles_n = (N + p.BN - 1) / p.BN;
  return p;
}
static void launch_finalize(const GemmDev& g, const pcm_gemm_epi* e, void* stream) {
  if (g.chstats) {
    int rpb, nqpb, threads, ny;
    finalize_stats_geometry(e->M, e->N, g.stats_rows, &rpb, &nqpb, &threads, &ny);
    PCM_LAUNCH(pcm_gemm_finalize_stats_kernel, dim3((e->M + rpb - 1) / rpb, ny), dim3(threads), 0, stream, g, rpb, nqpb);
    return;
  }
  long nq = (long)e->M * (e->N / 4);
  long fb = (nq + 255) / 256; if (fb > PCM_GRID_CAP(2048)) fb = PCM_GRID_CAP(2048);
  PCM_LAUNCH(pcm_gemm_finalize_kernel, dim3((int)fb), dim3(256), 0, stream, g);
}
static int gemm_total_kt(const pcm_gemm_seg* segs, int nseg) {
  int t = 0;
  for (int i = 0; i < nseg; i++) t += (segs[i].K + 63) / 64;
  return t;
}
static bool gemm_big_ok(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* e) {
  for (int i = 0; i < nseg; i++) {
    const pcm_gemm_seg& s = segs[i];
    if (s.K % 64) return false;
    const size_t a_bytes = s.mode == PCM_SEG_CONV3X3 ? (size_t)(e->M / (e->Ho * e->Wo > 0 ? e->Ho * e->Wo : 1)) * s.Hs * s.Ws * s.C * 2 : (size_t)e->M * s.lda * 2;
    if (a_bytes >= 0x7ff00000u || (size_t)e->N * s.K * 2 >= 0x7ff00000u) return false;
    if (s.mode == PCM_SEG_CONV3X3 && (e->Ho > 1023 || e->Wo > 1023 || e->M / (e->Ho * e->Wo) > 2047)) return false;
  }
  if (e->out_dtype == PCM_F32 || (e->N % 8) || (e->ldo % 8)) return false;
  if (e->residual && ((e->ldr % 8) || (((uintptr_t)e->residual) & 15))) return false;
  if (e->rowvec && (((uintptr_t)e->rowvec) & 15)) return false;
  return true;
}
// gemm4w.hip: plain segments only (K % 64 == 0 is part of big_ok)
static bool gemm_w4_ok(const pcm_gemm_seg* segs, int nseg) {
  for (int i = 0; i < nseg; i++)
    if (segs[i].mode != PCM_SEG_PLAIN) return false;
  return true;
}
// rank-64 projection that the streaming kernel (gemm_n64.hip) takes
static bool gemm_n64_ok(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* e) {
  return big_mode() > 0 && !g_force_bm && nseg == 1 && segs[0].mode == PCM_SEG_PLAIN && e->N == 64 && (segs[0].K % 64) == 0 &&
         e->out_dtype != PCM_F32 && !e->bias && !e->rowvec && !e->residual && e->act == PCM_ACT_NONE;
}
// batch-row projections (M <= 16: time embedding, time_emb_proj, adaLN modulation) that the weight-streaming kernel (gemm_smallm.hip) takes;
// PCM_GEMM_SMALLM=0 switches it off (A/B)
PCM_LAZY_KNOB(smallm_on, g_smallm_on, "PCM_GEMM_SMALLM", 1)
static bool gemm_smallm_ok(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* e) {
  if (!smallm_on() || big_mode() <= 0 || g_force_bm || e->M > 16 || (e->N % 4) || e->rowvec || e->residual) return false;
  if (e->act != PCM_ACT_NONE && e->act != PCM_ACT_SILU) return false;
  for (int i = 0; i < nseg; i++)
    if (segs[i].mode != PCM_SEG_PLAIN || (segs[i].K % 32) || (segs[i].lda % 8)) return false;
  return true;
}
// weights-stationary kernel (gemm_ws.hip): the short-K projections of the 64x64 level -- N a multiple of 320 (one 320-column weight slice per
// workgroup, held in registers), K total 320 or 384 (K + rank-64 LoRA), plain segments, enough 64-row tiles for every CU to amortise its weight
// fill.  BUILT, BIT-IDENTICAL TO THE PHASED TILE, MEASURED SLOWER (x1.06-1.35 on every N = 320 shape, profiles/r06_l_*: with ONE wave per SIMD the
// LDS-DMA issue slots, the LDS waits and the epilogue's ~700 instructions per 64 rows all ADD to the MFMA time -- 5.1 us per 64-row step against
// 1.4 us of MFMAs): it exists in the TOOLS build only (PCM_GEMM_WS=1 / pcm_debug_gemm_ws), the product planner never takes it.  Re-measured with the
// epilogue's LDS accesses untracked and as an eight-wave form (PCM_GEMM_WS=2): on par with the phased tile at best -- the shapes run at the fabric's
// 3.4-5 TB/s (header of gemm_ws.hip, profiles/r06_q_*).
PCM_LAZY_KNOB(ws_on, g_ws_on, "PCM_GEMM_WS", 0)
PCM_TOOLS_ONLY(extern "C" void pcm_debug_gemm_ws(int on) { g_ws_on = on < 0 ? -1 : (on > 2 ? 1 : on); })      // 1: four-wave form, 2: eight-wave form
static bool gemm_ws_ok(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* e) {
  if (!ws_on() || big_mode() != 1 || g_force_bm) return false;
  if (e->out_dtype == PCM_F32 || e->act != PCM_ACT_NONE || e->rowvec || e->out2 || e->chstats || e->pre_out) return false;
  if ((e->N % 320) || e->N > 960 || e->M < 16384 || (e->ldo % 8) || (((uintptr_t)e->out) & 15)) return false;
  if (e->residual && ((e->ldr % 8) || (((uintptr_t)e->residual) & 15))) return false;
  if (e->bias && (((uintptr_t)e->bias) & 15)) return false;
  int kt = 0;
  for (int i = 0; i < nseg; i++) {
    const pcm_gemm_seg& s = segs[i];
    if (s.mode != PCM_SEG_PLAIN || (s.K % 64) || (s.lda % 8) || (((uintptr_t)s.a) & 15) || (((uintptr_t)s.w) & 15)) return false;
    kt += s.K;
  }
  return kt == 320 || (kt == 384 && ws_on() != 2);      // (the eight-wave form exists for K = 320 only)
}
extern "C" size_t pcm_gemm_workspace_bytes(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* e) {
  if (!segs || !e || nseg < 1 || nseg > 2 || e->M <= 0 || e->N <= 0) return 0;
  if (gemm_smallm_ok(segs, nseg, e) || gemm_n64_ok(segs, nseg, e) || gemm_ws_ok(segs, nseg, e)) return 0;
  // the workspace is sized for the plan that would be used WITH a workspace; pcm_gemm_bf16 re-plans identically
  return gemm_plan(e->M, e->N, gemm_total_kt(segs, nseg), true, gemm_big_ok(segs, nseg, e), e->act == PCM_ACT_GEGLU, gemm_w4_ok(segs, nseg)).ws_bytes;
}

// validation + kernel choice + launch; plan_only: stop after the choice.  *code = the plan code of include/pcm_hip.h pcm_gemm_plan_code
static int gemm_run(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* e, void* stream, bool plan_only, int* code);
extern "C" int pcm_gemm_bf16(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* e, void* stream) {
  int code = 0;
  const int rc = gemm_run(segs, nseg, e, stream, false, &code);
  PCM_TOOLS_ONLY(if (rc == 0) g_last_plan = code;)
  return rc;
}
extern "C" int pcm_gemm_plan_code(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* e) {
  int code = 0;
  const int rc = gemm_run(segs, nseg, e, nullptr, true, &code);
  return rc ? rc : code;
}
static bool gemm_plan_emits_chstats(const GemmPlan& pl, const pcm_gemm_epi* e) {
  if (e->out_dtype == PCM_F32 || e->act == PCM_ACT_GEGLU || e->stats_rows <= 0 || (e->stats_rows % 64) || (e->M % e->stats_rows) || (e->N % 8)) return false;
  return pl.splitk > 1 || (pl.big_fn != 0 && pl.w4_fn == 0);
}
extern "C" int pcm_gemm_emits_chstats(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* e) {
  int code = 0;
  const int rc = gemm_run(segs, nseg, e, nullptr, true, &code);
  if (rc) return rc;
  if (code == 32 || code == 64 || code == 65 || code >= 10000) return 0;      // (batch-row, rank-64, gemm4w, weights-stationary: no statistics)
  pcm_gemm_epi probe = *e;
  GemmPlan pl; memset(&pl, 0, sizeof(pl));
  pl.big_fn = code / 1000; pl.splitk = code % 1000;
  return gemm_plan_emits_chstats(pl, &probe) ? 1 : 0;
}
static int gemm_run(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* e, void* stream, bool plan_only, int* code) {
  PCM_CHECK(segs && e && nseg >= 1 && nseg <= 2, PCM_EINVAL, "pcm_gemm_bf16: nseg must be 1 or 2");
  PCM_CHECK(e->M > 0 && e->N > 0 && (e->N % 4) == 0, PCM_EINVAL, "pcm_gemm_bf16: M>0, N>0, N%%4==0 required (M=%d N=%d)", e->M, e->N);
  GemmDev g;
  memset(&g, 0, sizeof(g));
  bool any_conv = false;
  for (int i = 0; i < nseg; i++) {
    const pcm_gemm_seg& s = segs[i];
    SegDev& d = g.seg[i];
    PCM_CHECK(s.a && s.w, PCM_EINVAL, "pcm_gemm_bf16: null operand in segment %d", i);
    PCM_CHECK(PCM_ALIGNED16(s.a) && PCM_ALIGNED16(s.w), PCM_EALIGN, "pcm_gemm_bf16: operands must be 16-byte aligned");
    PCM_CHECK(s.K > 0 && (s.K % 8) == 0, PCM_EINVAL, "pcm_gemm_bf16: K%%8 != 0 (K=%d)", s.K);
    d.a = (const bf16_t*)s.a; d.w = (const bf16_t*)s.w; d.K = s.K; d.lda = s.lda; d.mode = s.mode;
    d.Hs = s.Hs; d.Ws = s.Ws; d.C = s.C; d.stride = s.stride; d.src_mode = s.src_mode;
    d.ktiles = (s.K + 63) / 64;
    if (s.mode == PCM_SEG_CONV3X3) {
      any_conv = true;
      PCM_CHECK(s.C > 0 && (s.C % 64) == 0 && s.K == 9 * s.C, PCM_EINVAL, "pcm_gemm_bf16: conv segment needs C%%64==0 and K==9*C (C=%d K=%d)", s.C, s.K);
      PCM_CHECK(s.stride == 1 || s.stride == 2, PCM_EINVAL, "pcm_gemm_bf16: conv stride must be 1 or 2");
      PCM_CHECK(s.Hs > 0 && s.Ws > 0, PCM_EINVAL, "pcm_gemm_bf16: conv source dims");
    } else {
      PCM_CHECK(s.mode == PCM_SEG_PLAIN, PCM_EINVAL, "pcm_gemm_bf16: bad segment mode");
      PCM_CHECK(s.lda >= s.K && (s.lda % 8) == 0, PCM_EALIGN, "pcm_gemm_bf16: lda must be >=K and %%8==0");
    }
  }
  if (any_conv) PCM_CHECK(e->Ho > 0 && e->Wo > 0 && (e->M % (e->Ho * e->Wo)) == 0, PCM_EINVAL, "pcm_gemm_bf16: M must be B*Ho*Wo for conv");
  const bool geglu = e->act == PCM_ACT_GEGLU;
  PCM_CHECK(e->out && PCM_ALIGNED16(e->out) && (e->ldo % 4) == 0 && e->ldo >= (geglu ? e->N / 2 : e->N), PCM_EALIGN, "pcm_gemm_bf16: out/ldo alignment");
  if (geglu)
    PCM_CHECK((e->N % 16) == 0 && (e->ldo % 8) == 0 && e->out_dtype != PCM_F32 && !e->residual && !e->rowvec && gemm_big_ok(segs, nseg, e), PCM_EUNSUPPORTED,
              "pcm_gemm_bf16: PCM_ACT_GEGLU needs N%%16==0, K%%64==0 per segment, bf16 output, no residual / row vector");
  if (e->pre_out)
    PCM_CHECK(geglu && PCM_ALIGNED16(e->pre_out) && (e->ldp % 8) == 0 && e->ldp >= e->N && e->pre_rows >= 0, PCM_EINVAL,
              "pcm_gemm_bf16: pre_out needs PCM_ACT_GEGLU, 16-byte alignment, ldp%%8==0, ldp >= N");
  if (e->out2)
    PCM_CHECK(!geglu && e->out_dtype != PCM_F32 && (((uintptr_t)e->out2) & 7) == 0 && (e->ldo2 % 4) == 0 && e->ldo2 >= e->N && e->N != 64 && e->M > 16, PCM_EUNSUPPORTED,
              "pcm_gemm_bf16: out2 needs a bf16 output without PCM_ACT_GEGLU, 8-byte alignment, ldo2%%4==0, ldo2 >= N (not the rank-64 / batch-row kernels)");
  if (e->chstats)
    PCM_CHECK(e->stats_rows > 0 && (e->stats_rows % 64) == 0 && (e->M % e->stats_rows) == 0 && (((uintptr_t)e->chstats) & 7) == 0, PCM_EINVAL,
              "pcm_gemm_bf16: chstats needs stats_rows %% 64 == 0 and M %% stats_rows == 0");
  if (e->residual) PCM_CHECK((((uintptr_t)e->residual) & 7) == 0 && (e->ldr % 4) == 0, PCM_EALIGN, "pcm_gemm_bf16: residual alignment");
  if (e->rowvec) PCM_CHECK(e->rows_per_batch > 0, PCM_EINVAL, "pcm_gemm_bf16: rows_per_batch");
  g.nseg = nseg; g.M = e->M; g.N = e->N; g.Ho = e->Ho > 0 ? e->Ho : 1; g.Wo = e->Wo > 0 ? e->Wo : 1;
  g.bias = e->bias; g.rowvec = (const bf16_t*)e->rowvec; g.rpb = e->rows_per_batch > 0 ? e->rows_per_batch : 1;
  g.res = (const bf16_t*)e->residual; g.ldr = e->ldr; g.out = e->out; g.ldo = e->ldo;
  g.out_f32 = e->out_dtype == PCM_F32; g.act = e->act; g.alpha = e->alpha; g.dbg = g_ablate & 0xff;
  g.w4_stagger = (g_ablate >> 8) ? (g_ablate >> 8) - 1 : w4_stagger_default();
  g.pre_out = (bf16_t*)e->pre_out; g.pre_rows = e->pre_out ? e->pre_rows : 0; g.ldp = e->ldp;
  g.out2 = (bf16_t*)e->out2; g.ldo2 = e->ldo2;
  // conv addressing / K order: explicit choice through the env / debug hooks, otherwise by shape (gemm8p.hip launcher)
  // default: the tap-outer order with the per-tap re-key everywhere.  PCM_GEMM_CONV_CO=2 / pcm_debug_gemm_conv_order(2) = chunk-outer BY SHAPE
  // (8x8 maps only): x1.12 on that launch alone with cold operands (round 2), but 117.5-117.7 vs 117.2-117.3 ms per bs-16 step in the
  // interleaved A/B of round 3 (profiles/r03_k_conv_order_by_shape_ab.txt) -- not taken.
  // the environment is read ONCE (not per launch: ~5400 launches per eager step); -1 in g_conv_* = "the cached environment value", so the
  // debug hooks can still override and reset
#if PCM_HAS_TOOLS
  static const int env_co = pcm_env_int("PCM_GEMM_CONV_CO", 0);
  static const int env_md = pcm_env_int("PCM_GEMM_CONV_MD", 0) ? 1 : 0;
  const int cco = g_conv_co < 0 ? env_co : g_conv_co, cmd = g_conv_md < 0 ? env_md : g_conv_md;
#else
  constexpr int cco = 0, cmd = 0;      // product build: the tap-outer order with the per-tap re-key, the only conv variant instantiated (gemm8p.hip)
#endif
  g.conv_auto = cco == 2;
  g.conv_co = cco == 1; g.conv_md = cmd > 0 || cco == 1;
  if (e->N == 64 && nseg == 1 && segs[0].mode == PCM_SEG_CONV3X3) {   // conv LoRA down-projection: halo-window kernel where the geometry allows
    const int rc = pcm_conv_r64_launch(g, stream, plan_only);
    if (rc < 0) return rc;
    if (rc == 0) { *code = 65; return plan_only ? PCM_OK : pcm_post_launch("pcm_gemm_bf16"); }
  }
  if (gemm_smallm_ok(segs, nseg, e)) {
    *code = 32;
    if (plan_only) return PCM_OK;
    int rc = pcm_gemm_smallm_launch(g, stream);
    if (rc) return rc;
    return pcm_post_launch("pcm_gemm_bf16");
  }
  if (gemm_n64_ok(segs, nseg, e)) {
    *code = 64;
    if (plan_only) return PCM_OK;
    int rc = pcm_gemm_n64_launch(g, stream);
    if (rc) return rc;
    return pcm_post_launch("pcm_gemm_bf16");
  }
  if (gemm_ws_ok(segs, nseg, e)) {
    const int form = ws_on() == 2 ? 2 : 1;
    *code = 30000 + 1000 * (form - 1) + (segs[0].K + (nseg > 1 ? segs[1].K : 0)) / 32;
    if (plan_only) return PCM_OK;
    int rc = pcm_gemm_ws_launch(g, stream, form);
    if (rc) return rc;
    return pcm_post_launch("pcm_gemm_bf16");
  }
  // (plan_only: the plan a call WITH a workspace would take, as pcm_gemm_workspace_bytes sizes it)
  GemmPlan pl = gemm_plan(e->M, e->N, gemm_total_kt(segs, nseg), plan_only || e->workspace != nullptr, gemm_big_ok(segs, nseg, e), geglu, gemm_w4_ok(segs, nseg));
  *code = pl.w4_fn ? 10000 + 1000 * pl.w4_fn + 1 : 1000 * pl.big_fn + pl.splitk;
  if (plan_only) return PCM_OK;
  if (pl.splitk > 1) {
    PCM_CHECK(e->workspace_bytes >= pl.ws_bytes && PCM_ALIGNED16(e->workspace), PCM_EINVAL,
              "pcm_gemm_bf16: workspace too small (%zu < %zu) or unaligned", (size_t)e->workspace_bytes, pl.ws_bytes);
    g.ws = (float*)e->workspace;
  }
  g.tiles_m = pl.tiles_m; g.tiles_n = pl.tiles_n; g.splitk = pl.splitk; g.kt_per_split = pl.kt_per_split;
  // per-channel statistics (abi 5): the unsplit phased tile's epilogue and every split-K finalize emit them; other plans ignore the field
  // (pcm_gemm_emits_chstats tells the caller, who then runs the statistics pass)
  if (e->chstats && gemm_plan_emits_chstats(pl, e)) { g.chstats = e->chstats; g.stats_rows = e->stats_rows; }
  if (pl.w4_fn) {
    int rc = pcm_gemm4w_launch(g, pl.w4_fn, stream);
    if (rc) return rc;
    return pcm_post_launch("pcm_gemm_bf16");
  }
  if (pl.big_fn) {
    int rc = pcm_gemm8p_launch(g, pl.big_fn, stream);
    if (rc) return rc;
    if (pl.splitk > 1) launch_finalize(g, e, stream);
    return pcm_post_launch("pcm_gemm_bf16");
  }
  dim3 grid(g.tiles_m * g.tiles_n, pl.splitk);
  size_t smem = 2 * (size_t)(pl.BM + pl.BN) * 128;
  // tiles above 64 KB of dynamic LDS need the per-function cap raised once
#define PCM_GEMM_LAUNCH(NW, WGM, TM, TN)                                                                          \
  do {                                                                                                             \
    static bool lds_ok = false;                                                                                    \
    if (!lds_ok && smem > 65536) {                                                                                 \
      hipError_t er = hipFuncSetAttribute((const void*)pcm_gemm_kernel<NW, WGM, TM, TN>,                           \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                  \
      PCM_CHECK(er == hipSuccess, PCM_EHIP, "pcm_gemm_bf16: hipFuncSetAttribute(LDS %zu): %s", smem, hipGetErrorString(er)); \
      lds_ok = true;                                                                                               \
    }                                                                                                              \
    PCM_LAUNCH((pcm_gemm_kernel<NW, WGM, TM, TN>), grid, dim3(64 * NW), smem, stream, g);                          \
  } while (0)
  if (pl.BM == 256 && pl.BN == 128) PCM_GEMM_LAUNCH(8, 4, 2, 2);
  else if (pl.BM == 128 && pl.BN == 128) PCM_GEMM_LAUNCH(4, 2, 2, 2);
  else if (pl.BM == 256 && pl.BN == 64) PCM_GEMM_LAUNCH(4, 4, 2, 2);
  else if (pl.BM == 128 && pl.BN == 64) PCM_GEMM_LAUNCH(4, 2, 2, 1);
  else PCM_GEMM_LAUNCH(4, 2, 1, 1);
  if (pl.splitk > 1) launch_finalize(g, e, stream);
  return pcm_post_launch("pcm_gemm_bf16");
}
