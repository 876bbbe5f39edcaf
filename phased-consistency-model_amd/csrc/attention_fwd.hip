// (no pcm-build-flags line on purpose: this file is built with hipcc's default accumulator form; its VGPR / ACCVGPR variants are selected per
// kernel inside the source, profiles/r03_b_attention_fwd_variants.txt)
// Software-pipelined attention forward (SDPA over latent tokens; replaces F.scaled_dot_product_attention / xformers,
// train_pcm_lora_sd15.py:947-957, for every forward of the step).
//
// Why a second forward kernel.  The first one (attention.hip) runs each 64-key tile as a dependent chain inside a wave:
//   QK^T MFMAs -> softmax VALU -> PV MFMAs, and relies on three co-resident waves per SIMD to overlap one wave's VALU with another's MFMAs.
// This one makes the overlap available INSIDE every wave's instruction stream:
//
//   body j (one basic block in the steady state):   S_{j+1} = K_{j+1} Q^T          6 MFMAs  \   independent of each other:
//                                                    O     += V_{j-1}^T P_{j-1}^T    8 MFMAs   >  sched_group_barrier interleaves them
//                                                    P_j    = softmax-numerators(S_j)  ~100 VALU /   (one MFMA, 3 exp, 4 VALU, ...)
//
// i.e. the score MFMAs run one tile AHEAD and the PV MFMAs one tile BEHIND the softmax.  K and V tiles are double-buffered in LDS (K_{j+1}
// and V_{j-1} are read while K_{j+2} and V_j are stored), one barrier per tile instead of two; global loads run two tiles ahead in
// registers.  The lazy softmax reference (attention.hip) is kept; a reference move rescales O AFTER the body's PV MFMAs (P_{j-1} is still
// in the old scale), which only costs anything on the rare tiles that move it.
// What it bought is measured, not assumed: see the table at the launcher below (3-5 % at head dims 64 / 80, nothing at 40).
#include "attn_dev.h"

// softmax numerators of one 64-key tile held as S^T fragments (lane = query row l31, 16 + 16 scores): raw-domain exp2 against the lazily
// moved reference.  Returns the packed bf16 P fragments; ``alpha`` (per lane) and ``moved`` (wave-uniform) describe a reference move.
template <int D, bool MASK>
__device__ __forceinline__ void softmax_tile(f32x16 (&s_)[2], float sc, float& m_run, float& l_run, bf16x8 (&pf)[4], float& alpha,
                                             bool& moved, int kv0, int Lk, int hi) {
  using C = AttnCfg<D>;
  constexpr bool ONES = C::DV * 32 > D && C::RKU > C::DG;
  if constexpr (MASK) {
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        int kv = kv0 + 32 * t + 4 * hi + (r & 3) + 8 * (r >> 2);
        if (kv >= Lk) s_[t][r] = -1e30f;
      }
  }
  float mxa[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) mxa[r & 3] = fmaxf(mxa[r & 3], s_[t][r]);
  const float mx = pcm_xhalf_max(fmaxf(fmaxf(mxa[0], mxa[1]), fmaxf(mxa[2], mxa[3])));
  const float m_new = fmaxf(m_run, mx * sc);
  // branch-free (the body stays ONE basic block): without a move m_use = m_run and alpha = exp2(0) = 1 exactly
  moved = !__all(m_new <= m_run + 8.0f);
  const float m_use = moved ? m_new : m_run;
  alpha = PCM_EXP2F(m_run - m_use);
  l_run *= alpha;
  m_run = m_use;
  float psum = 0.f;
  {
    // two scores per v_pk_fma_f32: with VGPR-form accumulators VALU issue slots are what the tile time is made of (PMC, round 3: 8.3
    // VALU per MFMA with scalar fmas against 7.1 in the first kernel)
    const f32x2 sc2 = {sc, sc}, nm2 = {-m_run, -m_run};
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 x = pcm_pk_fma(f32x2{s_[t][r], s_[t][r + 1]}, sc2, nm2);
        const float p0 = PCM_EXP2F(x[0]), p1 = PCM_EXP2F(x[1]);
        s_[t][r] = p0; s_[t][r + 1] = p1;
        if (!ONES) psum += p0 + p1;
      }
  }
  l_run += psum;
#pragma unroll
  for (int ss = 0; ss < 4; ss++) pf[ss] = pack_frag(s_[ss >> 1], ss & 1);
}

template <int D>
__device__ __forceinline__ void qk_tile(const char* Kt, const bf16x8 (&qf)[AttnCfg<D>::DK16], f32x16 (&s_)[2], int l31, int hi) {
  using C = AttnCfg<D>;
#pragma unroll
  for (int t = 0; t < 2; t++) {
#pragma unroll
    for (int r = 0; r < 16; r++) s_[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < C::DK16; s++) {
      const bf16x8 kf = *(const bf16x8*)(Kt + ((32 * t + l31) * C::RKU + 2 * s + hi) * 16);
      s_[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s_[t], 0, 0, 0);
    }
  }
}
template <int D>
__device__ __forceinline__ void pv_tile(const char* Vt, const TrFrag<D>& trf, const bf16x8 (&pf)[4], f32x16 (&acc_o)[AttnCfg<D>::DV]) {
  using C = AttnCfg<D>;
#pragma unroll
  for (int i = 0; i < C::DV; i++)
#pragma unroll
    for (int ss = 0; ss < 4; ss++) acc_o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trf.get(Vt, i, ss), pf[ss], acc_o[i], 0, 0, 0);
}

// Instruction order of the steady-state body, as sched_group_barrier groups (LLVM SchedGroupMask: VALU 0x2, MFMA 0x8, VMEM read 0x20,
// DS read 0x100, DS write 0x200, TRANS 0x400): staging stores and the next tiles' global loads first, then ALL fragment reads of the tile
// (their latency is covered by the row-maximum chain, which needs no LDS data), then one MFMA per group with its share of the softmax
// VALU / transcendental work behind it.
template <int D>
__device__ __forceinline__ void fwd_sched_pipeline() {
  using C = AttnCfg<D>;
  constexpr int NM = 2 * C::DK16 + 4 * C::DV;           // MFMAs per body
  constexpr int NR = 2 * C::DK16 + 8 * C::DV;           // LDS fragment reads per body
  __builtin_amdgcn_sched_group_barrier(0x200, 2 * RowGeom<D, 64>::N, 0);
  __builtin_amdgcn_sched_group_barrier(0x020, 2 * RowGeom<D, 64>::N, 0);
  if constexpr (C::DV >= 3) {
    // wide heads (d = 80: 22 MFMAs, 34 fragment reads): all reads up front do not fit the register budget -- the K fragments first, the
    // V^T fragments two per MFMA group as the stream goes
    constexpr int NK = 2 * C::DK16, NV = 8 * C::DV;
    __builtin_amdgcn_sched_group_barrier(0x100, NK, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
    pcm_static_for<0, 3>([&](auto) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    });
    pcm_static_for<3, NM>([&](auto it) {
      constexpr int g = decltype(it)::value;
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      if constexpr (2 * g < NV) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    });
    return;
  }
  __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
  // row maximum (no exponential can start before it): ~26 VALU spread under the first three score MFMAs
  __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
  pcm_static_for<0, 3>([&](auto) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
  });
  pcm_static_for<3, NM>([&](auto) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x400, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
  });
}

// One pipelined tile body (see the file header): stage K_{j+2} / V_j into the free LDS images, issue the loads of K_{j+3} / V_{j+1},
// S_{j+1} = K_{j+1} Q^T into ``sn``, O += V_{j-1}^T P_{j-1}^T from ``pf``, P_j = softmax numerators of ``s_`` into ``pn``, barrier,
// deferred rescale of O when the softmax reference moved.
template <int D, bool FULL>
__device__ __forceinline__ void fwd_body(int j, char* Ks, char* Vs, const RowGeom<D, 64>& geo, RowStage<D, 64>& kst, RowStage<D, 64>& vst,
                                         const bf16_t* kb, const bf16_t* vb, int ldk, int Lk, int tid, int l31, int hi, const TrFrag<D>& trf,
                                         const bf16x8 (&qf)[AttnCfg<D>::DK16], float sc, f32x16 (&s_)[2], f32x16 (&sn)[2], const bf16x8 (&pf)[4],
                                         bf16x8 (&pn)[4], f32x16 (&acc_o)[AttnCfg<D>::DV], float& m_run, float& l_run) {
  using C = AttnCfg<D>;
  constexpr int TB = TileBytes<D>::value;
  char* Kj = Ks + (j & 1) * TB;            // K_{j+2} goes where K_j was
  char* Vj = Vs + (j & 1) * TB;            // V_j goes where V_{j-2} was
  const char* Kn = Ks + ((j + 1) & 1) * TB;
  const char* Vp = Vs + ((j + 1) & 1) * TB;
  if constexpr (FULL) {
    kst.store_full(geo, Kj, tid);
    vst.store_full(geo, Vj, tid);
    kst.load_full(geo, kb, ldk, (j + 3) * 64);
    vst.load_full(geo, vb, ldk, (j + 1) * 64);
  } else {
    kst.store(geo, Kj, Lk, tid);
    vst.store(geo, Vj, Lk, tid);
    kst.load(geo, kb, ldk, (j + 3) * 64, Lk, tid);
    vst.load(geo, vb, ldk, (j + 1) * 64, Lk, tid);
  }
  qk_tile<D>(Kn, qf, sn, l31, hi);
  pv_tile<D>(Vp, trf, pf, acc_o);
  float alpha;
  bool moved;
  softmax_tile<D, false>(s_, sc, m_run, l_run, pn, alpha, moved, j * 64, Lk, hi);
  // P_j is only consumed by the NEXT body: without this pin hipcc sinks the 32 exponentials below the branch at the end of this body, out
  // of the block that holds the MFMAs they are meant to run under
  PCM_HW_ONLY(asm volatile("" : "+v"(pn[0]), "+v"(pn[1]), "+v"(pn[2]), "+v"(pn[3])));
  if constexpr (FULL) fwd_sched_pipeline<D>();
  __syncthreads();
  if (moved) {
#pragma unroll
    for (int i = 0; i < C::DV; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc_o[i][r] *= alpha;
  }
}

// WPS = waves per SIMD the register budget is declared for (2: <= 256 registers; 1: the whole 512-entry file)
template <int D, bool AG, int WPS>
__global__ __launch_bounds__(256, WPS) void attn_fwd_pipe_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o, float* lse, int H,
                                                                  int Lq, int Lk, int ldq, int ldk, int ldo, float scale) {
  using C = AttnCfg<D>;
  constexpr int TB = TileBytes<D>::value;
  __shared__ __attribute__((aligned(16))) char Ks[2 * TB];
  __shared__ __attribute__((aligned(16))) char Vs[2 * TB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128 + wave * 32;
  const bf16_t* qb = q + (size_t)b * Lq * ldq + h * D;
  const bf16_t* kb = k + (size_t)b * Lk * ldk + h * D;
  const bf16_t* vb = v + (size_t)b * Lk * ldk + h * D;
  PCM_HW_ONLY(if constexpr (AG) asm volatile("; AccVGPR accumulators requested" : : "a"(0)));   // an 'a' operand makes the function use the AGPR MFMA forms
  bf16x8 qf[C::DK16];
#pragma unroll
  for (int s = 0; s < C::DK16; s++) qf[s] = gfrag<D>(qb, ldq, q0 + l31, Lq, s, hi);
  f32x16 acc_o[C::DV];
#pragma unroll
  for (int i = 0; i < C::DV; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc_o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float sc = scale * LOG2E;
  constexpr bool ONES = C::DV * 32 > D && C::RKU > C::DG;
  const int nt = (Lk + 63) >> 6, nfull = Lk >> 6;
  // LDS images: K_0, K_1 staged here; the V image of "tile -1" (read by body 0 with P = 0) must hold finite numbers: zeros
  fill_pad_chunks<D, 64>(Ks, tid, false);
  fill_pad_chunks<D, 64>(Ks + TB, tid, false);
  fill_pad_chunks<D, 64>(Vs, tid, ONES);
  fill_pad_chunks<D, 64>(Vs + TB, tid, ONES);
  for (int u = tid; u < 64 * C::DG; u += 256) {
    const int r = u / C::DG, c = u - r * C::DG;
    *(uint4*)(Vs + TB + (r * C::RKU + c) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  const TrFrag<D> trf(lane);
  const RowGeom<D, 64> geo(ldk, tid);
  RowStage<D, 64> kst, vst;
  kst.load(geo, kb, ldk, 0, Lk, tid);
  vst.load(geo, kb, ldk, 64, Lk, tid);            // K_1 through the second stage
  kst.store(geo, Ks, Lk, tid);
  vst.store(geo, Ks + TB, Lk, tid);
  kst.load(geo, kb, ldk, 128, Lk, tid);           // in flight: K_2 and V_0, stored by body 0
  vst.load(geo, vb, ldk, 0, Lk, tid);
  __syncthreads();
  f32x16 s_[2], sn[2];
  bf16x8 pf[4] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}}, pn[4];
  qk_tile<D>(Ks, qf, s_, l31, hi);
  __syncthreads();          // body 0 overwrites the K_0 image with K_2: every wave must have read its K_0 fragments
  // ---- steady state: every tile touched is a full one -> no branch inside the body; two bodies per trip with the register sets swapped
  // (scores S_j / S_{j+1} and probabilities P_{j-1} / P_j ping-pong between two names: no register copies at the loop edge)
  const int jf = nfull - 3 > 0 ? nfull - 3 : 0;      // bodies j < jf only touch full tiles (K_{j+3}, V_{j+1}) and have a next tile
  int j = 0;
  for (; j + 1 < jf; j += 2) {
    fwd_body<D, true>(j, Ks, Vs, geo, kst, vst, kb, vb, ldk, Lk, tid, l31, hi, trf, qf, sc, s_, sn, pf, pn, acc_o, m_run, l_run);
    fwd_body<D, true>(j + 1, Ks, Vs, geo, kst, vst, kb, vb, ldk, Lk, tid, l31, hi, trf, qf, sc, sn, s_, pn, pf, acc_o, m_run, l_run);
  }
  // ---- the remaining tiles before the final one: same body with clamped / predicated staging
  for (; j < nt - 1; j++) {
    fwd_body<D, false>(j, Ks, Vs, geo, kst, vst, kb, vb, ldk, Lk, tid, l31, hi, trf, qf, sc, s_, sn, pf, pn, acc_o, m_run, l_run);
#pragma unroll
    for (int t = 0; t < 2; t++) s_[t] = sn[t];
#pragma unroll
    for (int ss = 0; ss < 4; ss++) pf[ss] = pn[ss];
  }
  // ---- final tile j = nt - 1 (possibly ragged): no next scores; PV of the previous tile, softmax, then its own PV
  {
    vst.store(geo, Vs + (j & 1) * TB, Lk, tid);
    pv_tile<D>(Vs + ((j + 1) & 1) * TB, trf, pf, acc_o);
    float alpha; bool moved;
    softmax_tile<D, true>(s_, sc, m_run, l_run, pn, alpha, moved, j * 64, Lk, hi);
    __syncthreads();
    if (moved) {
#pragma unroll
      for (int i = 0; i < C::DV; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc_o[i][r] *= alpha;
    }
    pv_tile<D>(Vs + (j & 1) * TB, trf, pn, acc_o);
  }
  float l_tot = pcm_xhalf_sum(l_run);
  if constexpr (ONES) {
    constexpr int LOC = D % 32;
    static_assert(((LOC >> 2) & 1) == 0, "ones row must sit in the low lane half");
    l_tot = __shfl(acc_o[D / 32][(LOC & 3) + 4 * (LOC >> 3)], l31);
  }
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + l31;
  if (qrow < Lq) {
    if (hi == 0 && lse) lse[((size_t)b * H + h) * Lq + qrow] = m_run + log2f(l_tot);
    bf16_t* orow = o + ((size_t)b * Lq + qrow) * ldo + h * D;
#pragma unroll
    for (int i = 0; i < C::DV; i++)
#pragma unroll
      for (int qd = 0; qd < 4; qd++) {
        const int dcol = 32 * i + 8 * qd + 4 * hi;
        if (dcol < D)
          *(uint2*)(orow + dcol) = make_uint2(pack_bf2(acc_o[i][4 * qd] * inv, acc_o[i][4 * qd + 1] * inv),
                                              pack_bf2(acc_o[i][4 * qd + 2] * inv, acc_o[i][4 * qd + 3] * inv));
      }
  }
}

// Which forward runs (pcm_debug_attn_fwd_variant): -1 = by shape (default, the round-3 measurement below), 0 = always the first kernel
// (attention.hip), 1 = pipelined / VGPR accumulators, 2 = pipelined / AccVGPR accumulators, 3 = pipelined / AccVGPR / one wave per SIMD.
//
// MEASURED on MI355X (profiles/r03_b_attention_fwd_variants.txt, TFLOP/s, interleaved timing of the four variants of one build):
//   shape (B, H, L, Lk, d)        first kernel   pipelined VGPR   pipelined AGPR (2 / SIMD)   AGPR (1 / SIMD)
//   32, 8, 4096, 4096, 40              614            594 (x0.97)        552 (x0.90)              503 (x0.82)
//   32, 8, 1024, 1024, 80              620            648 (x1.05)        489                      468
//    4, 10, 4096, 4096, 64             776            800 (x1.03)        690                      495
//    2, 24, 4250, 4250, 64             720            756 (x1.05)        646                      455
//   32, 8, 4096, 77, 40                206            137 (x0.66: the pipeline's prologue is most of a two-tile stream)
// i.e. giving every wave independent VALU work next to its MFMAs buys 3-5 % at head dims 64 / 80 and nothing at 40.  With VGPR
// accumulators the MFMA and the VALU of one SIMD contend for the VGPR ports whichever wave they come from (same throughput from three
// dependent-chain waves per SIMD as from two pipelined ones: the per-tile time stays the SUM of ~450 MFMA-issue and ~550 VALU-issue
// cycles), and the AccVGPR forms hipcc generates pay for the separation with 64-94 v_accvgpr copies per tile.  The d = 40 forward is
// VALU-bound on this chip: 32 v_exp_f32 (quarter rate) + ~65 other VALU per 14 MFMAs, against the ~4 VALU a 32x32x16 MFMA can hide.
// After the second pass (packed fma in the softmax: 81 instead of 95 VALU per d = 40 body; V^T reads streamed with the MFMA groups at
// d = 80; profiles/r03_e_attention_fwd_variants.txt): d = 40 x0.99-1.00, d = 80 x1.11, d = 64 x1.01 (L 4096) / x1.05 (L 4250) / x0.98 (L 1024).
// Default: the pipelined VGPR form where it measured faster (d = 80 from 8 key tiles, d = 64 from 32 key tiles), the first kernel elsewhere.
PCM_KNOB int g_attn_fwd_variant = -1;
PCM_TOOLS_ONLY(extern "C" void pcm_debug_attn_fwd_variant(int v) { g_attn_fwd_variant = v; }
               extern "C" int pcm_debug_attn_fwd_variant_get() { return g_attn_fwd_variant; })

// returns false when the first kernel (attention.hip) is to run: by choice of the variant, or no pipelined instantiation for the head dim
bool pcm_attn_fwd_pipe_launch(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk, int d, int ldq,
                              int ldk, int ldo, float scale, void* stream) {
  int var = g_attn_fwd_variant;
  if (var < 0) var = ((d == 80 && Lk >= 512) || (d == 64 && Lk >= 2048)) ? 1 : 0;
  if (var < 1 || var > 3 || d > 80) return false;
  dim3 grid((Lq + 127) / 128, H, B), block(256);
#define PIPE_CALL(DD, AGF, WPS)                                                                                                     \
  PCM_LAUNCH((attn_fwd_pipe_kernel<DD, AGF, WPS>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, \
             H, Lq, Lk, ldq, ldk, ldo, scale)
  // the two AccVGPR forms (measured 10-37 % slower, table above; the d = 80 one spills) are only built for A/B probes (-DPCM_ATTN_AGPR_VARIANTS)
#ifdef PCM_ATTN_AGPR_VARIANTS
#define PIPE_VAR(DD)                                        \
  if (var == 1) { PIPE_CALL(DD, false, 2); }                \
  else if (var == 2) { PIPE_CALL(DD, true, 2); }            \
  else { PIPE_CALL(DD, true, 1); }
#else
  if (var != 1) return false;
#define PIPE_VAR(DD) PIPE_CALL(DD, false, 2);
#endif
  switch (d) {
    case 32: PIPE_VAR(32) break;
    case 40: PIPE_VAR(40) break;
    case 64: PIPE_VAR(64) break;
    default: PIPE_VAR(80) break;
  }
  return true;
}
