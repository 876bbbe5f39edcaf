// Fused epilogue of the 128-row-per-wave tiles (gemm8p.hip: 256 x (64*FN), 8 waves as 2 x 4; gemm4w.hip: 128 x (64*FN), 4 waves as 1 x 4).
// A lane holds, per 16x16 accumulator (i, f), pixel row 16*i + (lane & 15) of its wave's 128 rows and channels 16*f + 4*(lane >> 4) .. +3.
// The fp32 tile goes through LDS in four passes of 32*WM rows (the K-loop stages are dead) so that the global side is whole 16-B pieces of
// rows: bias, per-image row vector (time embedding), SiLU, residual, fused GEGLU (+ optional pre-activation output), bf16 stores.
//
// Round 4 (measured with the stamps / ablations of tools/gemm4w_ablate.py, profiles/r04_b_*): the 4 passes of a short-K tile took 2-3x its
// K loop, and the stores were NOT the cost ("no global stores" changed nothing on the GEGLU projections).  What was:
//   * bias was re-loaded from global memory for every 8 / 16-column piece (16 scalar loads per GEGLU piece: 160 gather instructions per
//     thread and tile through the texture path).  Now BN/4 threads load the tile's bias row ONCE into LDS (behind the staging area) and the
//     pieces read it from there (alpha * acc, then + bias: the same fp32 operations in the same order as before).
//   * residual pieces were loaded two at a time inside the piece loop: ~3 exposed HBM round trips per pass, 12 per tile.  Now piece `it` of
//     pass q+1 is requested as soon as piece `it` of pass q has been consumed (ONE register set of 5 pieces per thread: a second set
//     spilled the 256 x 320 tile): the load flies under the rest of the pass and the next staging.  x0.90-0.97 on the residual-carrying
//     projections; epilogues without a residual keep the plain piece loop (the unrolled form cost them 3-15 %: profiles/r04_c_*).
#pragma once
#include "gemm_dev.h"

// one residual OR row-vector operand piece (the epilogues that carry both keep the in-loop loads for the second one)
template <int FN, int WM>
struct PcmEpi {
  static constexpr int WNC = 16 * FN, BN = 4 * WNC, ROWS = 32 * WM, NT = 256 * WM;
  static constexpr int CH = BN / 4, C8 = BN / 8, C16 = BN / 16;
  static constexpr int IT = ROWS * C8 / NT;            // 16-B output pieces per thread and pass
  static_assert(IT * NT == ROWS * C8, "store loop covers the pass exactly");

  // row / column of piece `it` of this thread in pass q
  static __device__ __forceinline__ void piece(int tid, int it, int q, int m0, int n0, int& lr, int& c8, int& m, int& n) {
    const int idx = tid + NT * it;
    lr = idx / C8; c8 = idx - lr * C8;
    m = m0 + 128 * (lr >> 5) + 32 * q + (lr & 31); n = n0 + 8 * c8;
  }

  static __device__ __forceinline__ uint4 fetch(const GemmDev& g, const bf16_t* src, bool is_res, int tid, int it, int q, int m0, int n0) {
    int lr, c8, m, n;
    piece(tid, it, q, m0, n0, lr, c8, m, n);
    if (m >= g.M || n >= g.N) return make_uint4(0u, 0u, 0u, 0u);
    return is_res ? *(const uint4*)(src + (size_t)m * g.ldr + n) : *(const uint4*)(src + (size_t)(m / g.rpb) * g.N + n);
  }

  // ---- fused GEGLU, in registers.  The packed weight rows are interleaved in groups of 2 (pcm_hip.h PCM_ACT_GEGLU): columns 4p .. 4p+3 of
  // the GEMM are [value 2p, value 2p+1, gate 2p, gate 2p+1], i.e. ONE lane's four accumulator elements hold both values and both gates of
  // output pair p -- no LDS round trip to bring them together.  out = v * gelu_erf(g) is packed to bf16 at once; the bf16 tile (half the
  // columns, half the bytes per element: 1/4 of the fp32 staging traffic) goes through LDS in two 64-row passes for 16-B coalesced
  // stores, together with the bf16 pre-activation tile when the caller keeps it for the backward.
  // (round 4: the previous form staged fp32, then ran value * gelu(gate) per 16-column piece in a serial per-thread loop: 2/3 of the
  //  tile time of the K = 320 feed-forward projection, tools/gemm4w_ablate.py.)
  static constexpr int OUT_STRIDE = BN + 16, PRE_STRIDE = 2 * BN + 16;      // bytes per staged row (+16: bank spread of the 4-B / 8-B writes)
  static constexpr int GROWS = 64 * WM;                                     // rows per GEGLU pass
  static constexpr size_t geglu_lds_bytes() { return (size_t)GROWS * (OUT_STRIDE + PRE_STRIDE); }
  template <typename Acc>
  static __device__ __forceinline__ void run_geglu(const GemmDev& g, char* smem, const Acc& acc, int tid, int wm, int wn, int m0, int n0, bool sync_first) {
    const int lane = tid & 63, frow = lane & 15, fk = lane >> 4;
    char* pre_lds = smem + (size_t)GROWS * OUT_STRIDE;
    f32x4 bq[FN];
#pragma unroll
    for (int f = 0; f < FN; f++) {
      const int n = n0 + WNC * wn + 16 * f + 4 * fk;
      const float4 b4 = (g.bias && n < g.N) ? *(const float4*)(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      bq[f] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
    const bool keep_pre = g.pre_out != nullptr;
    constexpr int OC = BN / 16, PC = BN / 8;          // 16-B pieces per staged row: outputs / pre-activations
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (q || sync_first) __syncthreads();
#pragma unroll
      for (int ii = 0; ii < 4; ii++) {
        const int lr = 64 * wm + 16 * ii + frow;
#pragma unroll
        for (int f = 0; f < FN; f++) {
          const f32x4 a = acc[4 * q + ii][f] * g.alpha + bq[f];
          const unsigned o = pack_bf2(a[0] * gelu_erf_f(a[2]), a[1] * gelu_erf_f(a[3]));
          *(unsigned*)(smem + (size_t)lr * OUT_STRIDE + 2 * (WNC / 2) * wn + 16 * f + 4 * fk) = o;
          if (keep_pre) *(uint2*)(pre_lds + (size_t)lr * PRE_STRIDE + 2 * WNC * wn + 32 * f + 8 * fk) = make_uint2(pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]));
        }
      }
      __syncthreads();
      for (int idx = tid; idx < GROWS * OC; idx += NT) {
        const int lr = idx / OC, c = idx - lr * OC;
        const int m = m0 + 128 * (lr >> 6) + 64 * q + (lr & 63), n = n0 + 16 * c;
        if (m >= g.M || n >= g.N) continue;
        if (PCM_ABL(1)) { if (g.alpha == 123456.f) continue; }
        *(uint4*)((bf16_t*)g.out + (size_t)m * g.ldo + (n >> 1)) = *(const uint4*)(smem + (size_t)lr * OUT_STRIDE + 16 * c);
      }
      if (keep_pre)      // what the backward of GEGLU needs: the pre-activation of rows < pre_rows, in the interleaved column order
        for (int idx = tid; idx < GROWS * PC; idx += NT) {
          const int lr = idx / PC, c = idx - lr * PC;
          const int m = m0 + 128 * (lr >> 6) + 64 * q + (lr & 63), n = n0 + 8 * c;
          if (m >= g.M || m >= g.pre_rows || n >= g.N) continue;
          *(uint4*)(g.pre_out + (size_t)m * g.ldp + n) = *(const uint4*)(pre_lds + (size_t)lr * PRE_STRIDE + 16 * c);
        }
    }
  }

  // X ("extras", abi 5): a second copy of the output rows (g.out2) and / or the per-channel statistics of the tile (g.chstats).  A
  // COMPILE-TIME flag selected once per tile: the common tile runs exactly the piece loops it ran before (run-time tests of the two
  // pointers inside them cost the unswitched form, see finish below; measured +2.5 ms on the two-timestep forward: profiles/r06_b_*)
  template <typename Acc>
  static __device__ __forceinline__ void run(const GemmDev& g, char* smem, const Acc& acc, int tid, int wm, int wn, int m0, int n0, bool sync_first) {
    if (g.out2 || (WM == 2 && g.chstats)) run_impl<true>(g, smem, acc, tid, wm, wn, m0, n0, sync_first);
    else run_impl<false>(g, smem, acc, tid, wm, wn, m0, n0, sync_first);
  }
  template <bool X, typename Acc>
  static __device__ __forceinline__ void run_impl(const GemmDev& g, char* smem, const Acc& acc, int tid, int wm, int wn, int m0, int n0, bool sync_first) {
    // the piece geometry below is a function of tid only: opaque copies keep hipcc from computing it (or anything shared with it) ahead of
    // the K loop and carrying it through the loop -- the 256 x 320 tile has no register to spare there (measured: 60+ spills, reloads inside
    // the K loop, without this)
    PCM_PIN_V(tid);
    PCM_PIN_S(m0);
    PCM_PIN_S(n0);
    if (g.act == PCM_ACT_GEGLU) { run_geglu(g, smem, acc, tid, wm, wn, m0, n0, sync_first); return; }
    const int lane = tid & 63, frow = lane & 15, fk = lane >> 4;
    // residual pieces are requested a pass ahead of their use (one register set of IT pieces; a row vector rides in the loop: L2-resident)
#ifdef PCM_EPI_NO_RES_PATH      // (A/B build only: tools/jobs/r04_g_*.sh)
    const bool pf_on = false;
#else
    const bool pf_on = g.res != nullptr;
#endif
    uint4 pf[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) pf[it] = pf_on ? fetch(g, g.res, true, tid, it, 0, m0, n0) : make_uint4(0u, 0u, 0u, 0u);
    // the tile's BN bias values: ONE global load by BN/4 threads, kept in LDS behind the staging area for all four passes
    float4* bias_lds = (float4*)(smem + (size_t)ROWS * CH * 16);
    // statistics image behind the bias row (WM == 2 only: the 256-row tile's dead K-loop stages have the room), see stats_pass
    char* sb = (X && WM == 2 && g.chstats) ? smem + (size_t)ROWS * CH * 16 + BN * 4 : nullptr;
    const bool st_merge = (g.stats_rows % (128 * WM)) == 0;
    ColAcc st;
    stats_zero(st);
    auto after_pass = [&](int q) {       // the pass's stored pieces are parked in sb: add them up (workgroup-uniform branch)
      if (!X || !sb) return;
      __syncthreads();
      stats_pass(sb, st, tid);
      if (!st_merge || q == 3) stats_flush(g, st, tid, n0, m0 + 32 * q, m0 + 128 + 32 * q, st_merge);
    };
    float4 b_mine = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < BN / 4 && g.bias && n0 + 4 * tid < g.N) b_mine = *(const float4*)(g.bias + n0 + 4 * tid);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (q || sync_first) __syncthreads();
#pragma unroll
      for (int ii = 0; ii < 2; ii++) {
        const int lr = 32 * wm + 16 * ii + frow;
#pragma unroll
        for (int f = 0; f < FN; f++) {
          const int ch = (WNC / 4) * wn + 4 * f + fk;
          const f32x4 a = acc[2 * q + ii][f] * g.alpha;
          *(float4*)(smem + ((size_t)lr * CH + (ch ^ (lr & 15))) * 16) = make_float4(a[0], a[1], a[2], a[3]);
        }
      }
      if (q == 0 && tid < BN / 4) bias_lds[tid] = b_mine;
      __syncthreads();
      if (pf_on) {
#pragma unroll
        for (int it = 0; it < IT; it++) {
          int lr, c8, m, n;
          piece(tid, it, q, m0, n0, lr, c8, m, n);
          const uint4 px = pf[it];
          if (q < 3) pf[it] = fetch(g, g.res, true, tid, it, q + 1, m0, n0);    // the same piece of the next pass, a pass ahead
          char* sbp = (X && sb) ? sb + (size_t)(tid + NT * it) * 16 : nullptr;
          if (m >= g.M || n >= g.N) { if (X && sbp) *(uint4*)sbp = make_uint4(0u, 0u, 0u, 0u); continue; }
          const float4 lo = *(const float4*)(smem + ((size_t)lr * CH + ((2 * c8) ^ (lr & 15))) * 16);
          const float4 hi4 = *(const float4*)(smem + ((size_t)lr * CH + ((2 * c8 + 1) ^ (lr & 15))) * 16);
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
          if (PCM_ABL(1)) { if (v[0] != 123456.f) continue; }
          finish_any<X>(g, bias_lds, m, n, c8, v, px, sbp);
        }
        after_pass(q);
        continue;
      }
      // no residual: bias-free (q / k / v, LoRA-down style projections), bias only, bias + time-embedding row (resnet conv1), generic
      if (g.act != PCM_ACT_SILU) {
        if (!g.rowvec) {
          if (g.bias) plain_pass<true, false, X>(g, smem, bias_lds, tid, q, m0, n0, sb); else plain_pass<false, false, X>(g, smem, bias_lds, tid, q, m0, n0, sb);
          after_pass(q);
          continue;
        }
        if (g.bias) { plain_pass<true, true, X>(g, smem, bias_lds, tid, q, m0, n0, sb); after_pass(q); continue; }
      }
      for (int idx = tid; idx < ROWS * C8; idx += NT) {
        const int lr = idx / C8, c8 = idx - lr * C8;
        const int m = m0 + 128 * (lr >> 5) + 32 * q + (lr & 31), n = n0 + 8 * c8;
        char* sbp = (X && sb) ? sb + (size_t)idx * 16 : nullptr;
        if (m >= g.M || n >= g.N) { if (X && sbp) *(uint4*)sbp = make_uint4(0u, 0u, 0u, 0u); continue; }
        const float4 lo = *(const float4*)(smem + ((size_t)lr * CH + ((2 * c8) ^ (lr & 15))) * 16);
        const float4 hi4 = *(const float4*)(smem + ((size_t)lr * CH + ((2 * c8 + 1) ^ (lr & 15))) * 16);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
        if (PCM_ABL(1)) { if (v[0] != 123456.f) continue; }
        finish_any<X>(g, bias_lds, m, n, c8, v, make_uint4(0u, 0u, 0u, 0u), sbp);
      }
      after_pass(q);
    }
  }

  // bias (LDS copy of the tile's bias row), row vector, SiLU, residual (already loaded: ``res``), one 16-B bf16 store.  The flags are
  // COMPILE-TIME: the piece loops are instantiated per flag set and selected once per tile (hipcc unswitched the old single loop by itself;
  // it does not do so for this one: measured as 96 instead of 56 instructions per piece and +3-6 % on the bias-free projections)
  template <bool BIAS, bool RV, bool SILU, bool RES, bool X>
  static __device__ __forceinline__ void finish(const GemmDev& g, const float4* bias_lds, int m, int n, int c8, float (&v)[8], const uint4 res, char* sbp = nullptr) {
    if (BIAS) {
      const float4 b0 = bias_lds[2 * c8], b1 = bias_lds[2 * c8 + 1];
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (RV) {
      const uint4 t = *(const uint4*)(g.rowvec + (size_t)(m / g.rpb) * g.N + n);
      const unsigned tw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int e = 0; e < 4; e++) { v[2 * e] += bf2f((bf16_t)(tw[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(tw[e] >> 16)); }
    }
    if (SILU) {
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = silu_f(v[e]);
    }
    if (RES) {
      const unsigned tw[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
      for (int e = 0; e < 4; e++) { v[2 * e] += bf2f((bf16_t)(tw[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(tw[e] >> 16)); }
    }
    const uint4 o = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    *(uint4*)((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
    if (X) {
      if (g.out2) *(uint4*)(g.out2 + (size_t)m * g.ldo2 + n) = o;    // (wave-uniform: a skip tensor's second home, its concat buffer)
      if (sbp) *(uint4*)sbp = o;                                      // the STORED values, for the per-channel statistics pass (stats_pass)
    }
  }
  // the generic form (flags read at run time): the residual path and the rare flag sets
  template <bool X>
  static __device__ __forceinline__ void finish_any(const GemmDev& g, const float4* bias_lds, int m, int n, int c8, float (&v)[8], const uint4 res, char* sbp = nullptr) {
    if (g.bias) {
      const float4 b0 = bias_lds[2 * c8], b1 = bias_lds[2 * c8 + 1];
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
      const unsigned tw[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
      for (int e = 0; e < 4; e++) { v[2 * e] += bf2f((bf16_t)(tw[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(tw[e] >> 16)); }
    }
    const uint4 o = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    *(uint4*)((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
    if (X) {
      if (g.out2) *(uint4*)(g.out2 + (size_t)m * g.ldo2 + n) = o;    // (wave-uniform: a skip tensor's second home, its concat buffer)
      if (sbp) *(uint4*)sbp = o;                                      // the STORED values, for the per-channel statistics pass (stats_pass)
    }
  }
  // one pass of the plain piece loop with compile-time flags
  template <bool BIAS, bool RV, bool X>
  static __device__ __forceinline__ void plain_pass(const GemmDev& g, const char* smem, const float4* bias_lds, int tid, int q, int m0, int n0, char* sb) {
    for (int idx = tid; idx < ROWS * C8; idx += NT) {
      const int lr = idx / C8, c8 = idx - lr * C8;
      const int m = m0 + 128 * (lr >> 5) + 32 * q + (lr & 31), n = n0 + 8 * c8;
      char* sbp = (X && sb) ? sb + (size_t)idx * 16 : nullptr;
      if (m >= g.M || n >= g.N) { if (X && sbp) *(uint4*)sbp = make_uint4(0u, 0u, 0u, 0u); continue; }
      const float4 lo = *(const float4*)(smem + ((size_t)lr * CH + ((2 * c8) ^ (lr & 15))) * 16);
      const float4 hi4 = *(const float4*)(smem + ((size_t)lr * CH + ((2 * c8 + 1) ^ (lr & 15))) * 16);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
      if (PCM_ABL(1)) { if (v[0] != 123456.f) continue; }
      finish<BIAS, RV, false, false, X>(g, bias_lds, m, n, c8, v, make_uint4(0u, 0u, 0u, 0u), sbp);
    }
  }

  // ---- per-channel statistics of the tile (abi 5, pcm_gemm_epi.chstats): the GroupNorm that reads this output next needs sum / sum of squares
  // per (sample, group) of the STORED values.  The piece loops park the packed 16-bit pieces of a pass in an LDS image [ROWS][BN]; here
  // thread t < BN/2 adds up columns 2t, 2t+1 over the pass's two 32-row runs (rows 32q.. of each 128-row half of the tile).  A tile that
  // lies inside one sample (stats_rows % (128 * WM) == 0) keeps the sums in registers and issues its BN x 2 fp64 atomics once; smaller feature maps
  // (8x8, 16x8: a run of 32 rows is still inside one sample) flush after every pass.
  struct ColAcc { float s[2][2], q[2][2]; };       // [row run][column of the pair]
  static __device__ __forceinline__ void stats_zero(ColAcc& a) {
#pragma unroll
    for (int h = 0; h < 2; h++) { a.s[h][0] = a.s[h][1] = a.q[h][0] = a.q[h][1] = 0.f; }
  }
  static __device__ __forceinline__ void stats_flush(const GemmDev& g, ColAcc& a, int tid, int n0, int m_run0, int m_run1, bool merge) {
    const int n = n0 + 2 * tid;
    if (tid < BN / 2 && n < g.N) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        if (merge && h == 1) break;
        const int mrow = h ? m_run1 : m_run0;
        if (mrow >= g.M) continue;
        double* dst = g.chstats + ((size_t)(mrow / g.stats_rows) * g.N + n) * 2;
        const float s0 = merge ? a.s[0][0] + a.s[1][0] : a.s[h][0], s1 = merge ? a.s[0][1] + a.s[1][1] : a.s[h][1];
        const float q0 = merge ? a.q[0][0] + a.q[1][0] : a.q[h][0], q1 = merge ? a.q[0][1] + a.q[1][1] : a.q[h][1];
        atomicAdd(dst, (double)s0); atomicAdd(dst + 1, (double)q0);
        atomicAdd(dst + 2, (double)s1); atomicAdd(dst + 3, (double)q1);
      }
    }
    stats_zero(a);
  }
  static __device__ __forceinline__ void stats_pass(const char* sb, ColAcc& a, int tid) {
    if (tid < BN / 2) {
#pragma unroll
      for (int h = 0; h < WM; h++)
#pragma unroll 8
        for (int r = 0; r < 32; r++) {
          const unsigned w = *(const unsigned*)(sb + (size_t)(32 * h + r) * (BN * 2) + 4 * tid);
          const float x0 = bf2f((bf16_t)(w & 0xffff)), x1 = bf2f((bf16_t)(w >> 16));
          a.s[h][0] += x0; a.q[h][0] = fmaf(x0, x0, a.q[h][0]);
          a.s[h][1] += x1; a.q[h][1] = fmaf(x1, x1, a.q[h][1]);
        }
    }
  }
};
