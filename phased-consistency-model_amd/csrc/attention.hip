// pcm-build-flags: -mllvm -amdgpu-mfma-vgpr-form
// (read by pcm_amd/build.py: MFMA results stay in VGPRs -- the softmax reads every accumulator; the AGPR form costs a v_accvgpr copy per value per tile)
// Scaled-dot-product attention over latent tokens for the SD1.5 UNet (8 heads, head_dim 40/80/160,
// Lq in {4096,1024,256,64}, Lk = Lq (self) or 77 (text)), flash-style: scores never leave the CU.
//
// Forward / dQ kernels: a workgroup owns 128 query rows (4 waves x 32) and streams 64-key tiles.
//   S^T = K Q^T is computed "swapped" (MFMA A = K tile from LDS, B = Q fragments in registers), so a
//   lane holds 16+16 scores of ONE query row: row max/sum need a single cross-half shuffle.
//   P^T feeds the second MFMA (O^T = V^T P^T) straight from the accumulator registers: the MFMA
//   contraction slot (hi, e) of step ss is DEFINED as key 16ss + 8(e>>2) + 4hi + (e&3) — exactly
//   what the lane already holds — and the V^T operand is read in that same key order out of the
//   ROW-MAJOR V tile by LDS transpose reads (ds_read_b64_tr_b16: 4 consecutive keys of one column
//   per read), so no permlane / LDS round trip for P and no transposed copy of V anywhere.
// dK/dV kernel: a workgroup owns 128 key rows and streams 64-query tiles with the roles swapped
//   (S = Q K^T, lane = one key), the same slot trick on the query index.
// head_dim 40 is padded to 48 for QK^T (K-step 16) and to 64 for the PV tile (32-row output tiles);
// padded rows of the accumulators are never stored.
#define ATTN_STAMPS_OWNER
#include "attn_dev.h"

// ============================================================================ forward
// second launch-bound = minimum waves per SIMD (HIP): three workgroups per CU (<= 168 VGPRs) hide the barrier / LDS latencies of a tile
// measurably better than two (tools/attn_occupancy.py).  Head dims 32 / 40 are held there; 64 lands at 166 VGPRs on its own.
template <int D>
__global__ __launch_bounds__(256, (D <= 40 ? 3 : 1)) void attn_fwd_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o,
                                                       float* lse, int H, int Lq, int Lk, int ldq, int ldk, int ldo, float scale ATTN_DBG_PARAM) {
  using C = AttnCfg<D>;
  __shared__ __attribute__((aligned(16))) char Ks[TileBytes<D>::value];
  __shared__ __attribute__((aligned(16))) char Vs[TileBytes<D>::value];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (xcd_remap) {
    // XCD-aware block -> (query block, head, image) map (opt-in A/B, pcm_debug_attn_xcd_remap): workgroup n runs on XCD n % 8, so with x
    // fastest the query blocks of ONE (image, head) spread over all eight L2s and each of them fetches that head's K / V.  Here every run
    // of 8 * gridDim.x workgroups serves eight (image, head) pairs, pair i entirely on XCD i; a ragged last group keeps the plain map.
    const int nx = gridDim.x, nbh = gridDim.y * gridDim.z;
    const int n = bx + nx * (by + (int)gridDim.y * bz);
    const int grp = n / (8 * nx), r = n - grp * 8 * nx;
    if (grp * 8 + 8 <= nbh) {
      const int pair = grp * 8 + (r & 7);
      bx = r >> 3; by = pair % (int)gridDim.y; bz = pair / (int)gridDim.y;
    }
  }
  const int b = bz, h = by, q0 = bx * 128 + wave * 32;
  const bf16_t* qb = q + (size_t)b * Lq * ldq + h * D;
  const bf16_t* kb = k + (size_t)b * Lk * ldk + h * D;
  const bf16_t* vb = v + (size_t)b * Lk * ldk + h * D;
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
  // head dims with a padded column (40 -> 48, 80 -> 96): column D of the V tile is set to ones, so the PV MFMA (V^T P^T) also
  // produces the softmax denominator sum_k p[k] in accumulator row D (rescaled with O for free, no VALU row sums)
  constexpr bool ONES = C::DV * 32 > D && C::RKU > C::DG;
  fill_pad_chunks<D, 64>(Ks, tid, false);
  fill_pad_chunks<D, 64>(Vs, tid, ONES);
  const TrFrag<D> trf(lane);
  const RowGeom<D, 64> geo(ldk, tid);
  RowStage<D, 64> kst, vst;
  if (AttnPrefetch<D>::value) { kst.load(geo, kb, ldk, 0, Lk, tid); vst.load(geo, vb, ldk, 0, Lk, tid); }
#ifdef PCM_ABLATE
  const bool stamp_on = blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == gridDim.z / 2;
#endif
  for (int kv0 = 0; kv0 < Lk; kv0 += 64) {
    ATTN_STAMP(0);
    if (!ATTN_ABL(16)) __syncthreads();
    ATTN_STAMP(1);
    if (AttnPrefetch<D>::value) {
      if (!ATTN_ABL(2) || kv0 == 0) { kst.store(geo, Ks, Lk, tid); vst.store(geo, Vs, Lk, tid); }
    } else {
      load_rowmajor<D, 64>(Ks, kb, ldk, kv0, Lk, tid);
      load_rowmajor<D, 64>(Vs, vb, ldk, kv0, Lk, tid);
    }
    ATTN_STAMP(2);
    if (!ATTN_ABL(16)) __syncthreads();
    ATTN_STAMP(3);
    if (AttnPrefetch<D>::value && kv0 + 64 < Lk && !ATTN_ABL(2)) { kst.load(geo, kb, ldk, kv0 + 64, Lk, tid); vst.load(geo, vb, ldk, kv0 + 64, Lk, tid); }
    f32x16 s_[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++) s_[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < C::DK16; s++) {
        bf16x8 kf = *(const bf16x8*)(Ks + ((32 * t + l31) * C::RKU + 2 * s + hi) * 16);
        if (ATTN_ABL(8)) { s_[t][s] += (float)kf[0]; continue; }
        s_[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s_[t], 0, 0, 0);
      }
    }
    // V^T fragments of this tile: in flight under the softmax (head dims whose 4*DV fragments fit the register budget)
    constexpr bool V_EARLY = D <= 80;
    TrQuad<D> vq[V_EARLY ? C::DV : 1];
    if constexpr (V_EARLY) pcm_static_for<0, C::DV>([&](auto it) { vq[decltype(it)::value].template issue<decltype(it)::value>(Vs, trf); });
    // online softmax in the log2 domain on RAW scores: p = exp2(s*sc - m); masks only on the tail tile
    if (kv0 + 64 > Lk) {
      asm volatile("" ::: "memory");   // keep this a real (wave-uniform) branch: if-converted it costs 3 VALU ops per score in every tile
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          int kv = kv0 + 32 * t + 4 * hi + (r & 3) + 8 * (r >> 2);
          if (kv >= Lk) s_[t][r] = -1e30f;
        }
    }
    // row maximum: four independent chains (a single chain of 16 dependent v_max3 sits on the critical path between the QK^T MFMAs and
    // the exponentials), halves combined by a VALU lane swap
    float mxa[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) mxa[r & 3] = fmaxf(mxa[r & 3], s_[t][r]);
    float mx = pcm_xhalf_max(fmaxf(fmaxf(mxa[0], mxa[1]), fmaxf(mxa[2], mxa[3])));
    const float m_new = fmaxf(m_run, mx * sc);
    // Lazy reference update: the running reference m_run only has to keep exp2(s*sc - m_run) inside the fp32 / bf16 range, it need not be
    // the exact row maximum (softmax is shift invariant; the final 1/l and the LSE use the same reference).  While no row of the wave
    // has grown by more than 2^8 the old reference stays (p <= 256, same relative precision in bf16) and the rescale of the O
    // accumulators -- 16 v_pk_mul_f32 + an exp per tile, needed on most tiles when the test was "m_new == m_run" -- is skipped.
    if (!__all(m_new <= m_run + 8.0f)) {
      const float alpha = PCM_EXP2F(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < C::DV; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc_o[i][r] *= alpha;
    }
    float psum = 0.f;
    {
      const f32x2 sc2 = {sc, sc}, nm2 = {-m_run, -m_run};
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {       // two scores per v_pk_fma_f32
          const f32x2 x = pcm_pk_fma(f32x2{s_[t][r], s_[t][r + 1]}, sc2, nm2);
          const float p0 = ATTN_ABL(1) ? x[0] : PCM_EXP2F(x[0]), p1 = ATTN_ABL(1) ? x[1] : PCM_EXP2F(x[1]);
          s_[t][r] = p0; s_[t][r + 1] = p1;
          if (!ONES) psum += p0 + p1;
        }
    }
    l_run += psum;
    bf16x8 pf[4];
#pragma unroll
    for (int ss = 0; ss < 4; ss++) pf[ss] = pack_frag(s_[ss >> 1], ss & 1);
    ATTN_STAMP(4);
    if constexpr (V_EARLY) {
      vq[0].wait();
      pcm_static_for<1, C::DV>([&](auto it) { vq[decltype(it)::value].keep(); });
    }
    pcm_static_for<0, C::DV>([&](auto it) {
      constexpr int i = decltype(it)::value;
      TrQuad<D>& v4 = vq[V_EARLY ? i : 0];
      if constexpr (!V_EARLY) { v4.template issue<i>(Vs, trf); v4.wait(); }
#pragma unroll
      for (int ss = 0; ss < 4; ss++) acc_o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v4.frag(ss), pf[ss], acc_o[i], 0, 0, 0);
    });
    ATTN_STAMP(5);
  }
  float l_tot = pcm_xhalf_sum(l_run);
  if constexpr (ONES) {   // accumulator row D: tile D/32, local row D%32 -> lane half (loc>>2)&1 (= 0 for 40 / 80), register (loc&3) + 4*(loc>>3)
    constexpr int LOC = D % 32;
    static_assert(((LOC >> 2) & 1) == 0, "ones row must sit in the low lane half");
    l_tot = __shfl(acc_o[D / 32][(LOC & 3) + 4 * (LOC >> 3)], l31);
  }
  float inv = 1.0f / l_tot;
  int qrow = q0 + l31;
  if (qrow < Lq) {
    if (hi == 0 && lse) lse[((size_t)b * H + h) * Lq + qrow] = m_run + log2f(l_tot);
    bf16_t* orow = o + ((size_t)b * Lq + qrow) * ldo + h * D;
#pragma unroll
    for (int i = 0; i < C::DV; i++)
#pragma unroll
      for (int qd = 0; qd < 4; qd++) {
        int dcol = 32 * i + 8 * qd + 4 * hi;
        if (dcol < D)
          *(uint2*)(orow + dcol) = make_uint2(pack_bf2(acc_o[i][4 * qd] * inv, acc_o[i][4 * qd + 1] * inv),
                                              pack_bf2(acc_o[i][4 * qd + 2] * inv, acc_o[i][4 * qd + 3] * inv));
      }
  }
}

// ============================================================================ backward: delta
// delta[b][h][q] = sum_d dO[q][d] * O[q][d]
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* o, const bf16_t* dO, float* delta, int H, int Lq, int d, int ldo) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*Lq*H
  long total = (long)gridDim.y * Lq * H;
  (void)total;
  int b = blockIdx.y;
  if (idx >= (long)Lq * H) return;
  int qrow = (int)(idx / H), h = (int)(idx % H);
  const bf16_t* op = o + ((size_t)b * Lq + qrow) * ldo + h * d;
  const bf16_t* dp = dO + ((size_t)b * Lq + qrow) * ldo + h * d;
  float acc = 0.f;
  for (int c = 0; c < d; c += 8) {
    uint4 a = *(const uint4*)(op + c), g = *(const uint4*)(dp + c);
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
      acc += bf2f((bf16_t)(aw[e] & 0xffff)) * bf2f((bf16_t)(gw[e] & 0xffff));
      acc += bf2f((bf16_t)(aw[e] >> 16)) * bf2f((bf16_t)(gw[e] >> 16));
    }
  }
  delta[((size_t)b * H + h) * Lq + qrow] = acc;
}

// ============================================================================ backward: dQ
template <int D>
__global__ __launch_bounds__(256, (D <= 40 ? 3 : 1)) void attn_bwd_dq_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* dO,
                                                          const float* lse, float* delta, const bf16_t* o, bf16_t* dq, int H, int Lq,
                                                          int Lk, int ldq, int ldk, int ldo, float scale) {
  using C = AttnCfg<D>;
  __shared__ __attribute__((aligned(16))) char Ks[TileBytes<D>::value];
  __shared__ __attribute__((aligned(16))) char Vs[TileBytes<D>::value];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128 + wave * 32;
  const bf16_t* qb = q + (size_t)b * Lq * ldq + h * D;
  const bf16_t* dob = dO + (size_t)b * Lq * ldo + h * D;
  const bf16_t* kb = k + (size_t)b * Lk * ldk + h * D;
  const bf16_t* vb = v + (size_t)b * Lk * ldk + h * D;
  bf16x8 qf[C::DK16], dof[C::DK16];
#pragma unroll
  for (int s = 0; s < C::DK16; s++) {
    qf[s] = gfrag<D>(qb, ldq, q0 + l31, Lq, s, hi);
    dof[s] = gfrag<D>(dob, ldo, q0 + l31, Lq, s, hi);
  }
  const int qrow = q0 + l31;
  const float L2 = qrow < Lq ? lse[((size_t)b * H + h) * Lq + qrow] : 0.f;
  // delta[q] = sum_d dO[q][d] * O[q][d]: with ``o`` given it is computed HERE from the dO fragments the lane already holds plus the matching O
  // fragments (two half-row partial sums, one lane swap) and published for the dK/dV kernel, which runs after this one -- the separate
  // attn_delta pass (one launch and one more read of O and dO per attention) is then skipped
  float dl;
  if (o) {
    const bf16_t* ob = o + (size_t)b * Lq * ldo + h * D;
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < C::DK16; s++) {
      const bf16x8 of = gfrag<D>(ob, ldo, qrow, Lq, s, hi);
#pragma unroll
      for (int e = 0; e < 8; e++) part += bf2f((bf16_t)of[e]) * bf2f((bf16_t)dof[s][e]);
    }
    dl = pcm_xhalf_sum(part);
    if (hi == 0 && qrow < Lq) delta[((size_t)b * H + h) * Lq + qrow] = dl;
  } else {
    dl = qrow < Lq ? delta[((size_t)b * H + h) * Lq + qrow] : 0.f;
  }
  f32x16 acc[C::DV];
#pragma unroll
  for (int i = 0; i < C::DV; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  const float sc = scale * LOG2E;
  fill_pad_chunks<D, 64>(Ks, tid, false);
  fill_pad_chunks<D, 64>(Vs, tid, false);
  const TrFrag<D> trf(lane);
  const RowGeom<D, 64> geo(ldk, tid);
  RowStage<D, 64> kst, vst;
  if (AttnPrefetch<D>::value) { kst.load(geo, kb, ldk, 0, Lk, tid); vst.load(geo, vb, ldk, 0, Lk, tid); }
  const f32x2 sc2 = {sc, sc}, nl2 = {-L2, -L2}, scl2 = {scale, scale}, ndl2 = {-dl * scale, -dl * scale};
  for (int kv0 = 0; kv0 < Lk; kv0 += 64) {
    __syncthreads();
    if (AttnPrefetch<D>::value) {
      kst.store(geo, Ks, Lk, tid); vst.store(geo, Vs, Lk, tid);
    } else {
      load_rowmajor<D, 64>(Ks, kb, ldk, kv0, Lk, tid);
      load_rowmajor<D, 64>(Vs, vb, ldk, kv0, Lk, tid);
    }
    __syncthreads();
    if (AttnPrefetch<D>::value && kv0 + 64 < Lk) {
      kst.load(geo, kb, ldk, kv0 + 64, Lk, tid); vst.load(geo, vb, ldk, kv0 + 64, Lk, tid);
    }
    f32x16 s_[2], dp[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++) { s_[t][r] = 0.f; dp[t][r] = 0.f; }
#pragma unroll
      for (int s = 0; s < C::DK16; s++) {
        bf16x8 kf = *(const bf16x8*)(Ks + ((32 * t + l31) * C::RKU + 2 * s + hi) * 16);
        bf16x8 vf = *(const bf16x8*)(Vs + ((32 * t + l31) * C::RKU + 2 * s + hi) * 16);
        s_[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s_[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s], dp[t], 0, 0, 0);
      }
    }
    // dS^T = p * (dP - delta) * scale, two scores per packed instruction: v_pk_fma (exponent), 2 x v_exp, v_pk_fma ((dP - delta) * scale),
    // v_pk_mul -- 2.5 VALU per score instead of 5 (VALU time is not hidden behind the MFMAs on this chip)
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 x = pcm_pk_fma(f32x2{s_[t][r], s_[t][r + 1]}, sc2, nl2);
        const f32x2 p = {PCM_EXP2F(x[0]), PCM_EXP2F(x[1])};
        const f32x2 y = pcm_pk_fma(f32x2{dp[t][r], dp[t][r + 1]}, scl2, ndl2) * p;
        s_[t][r] = y[0]; s_[t][r + 1] = y[1];
      }
    if (kv0 + 64 > Lk) {
      asm volatile("" ::: "memory");   // keep this a real (wave-uniform) branch: if-converted it costs 3 VALU ops per score in every tile
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          int kv = kv0 + 32 * t + 4 * hi + (r & 3) + 8 * (r >> 2);
          if (kv >= Lk) s_[t][r] = 0.f;
        }
    }
    bf16x8 df[4];
#pragma unroll
    for (int ss = 0; ss < 4; ss++) df[ss] = pack_frag(s_[ss >> 1], ss & 1);
    // K^T fragments: the tile of accumulator row group i+1 is in flight while group i multiplies
    TrQuad<D> kq[2];
    kq[0].template issue<0>(Ks, trf);
    pcm_static_for<0, C::DV>([&](auto it) {
      constexpr int i = decltype(it)::value;
      TrQuad<D>& k4 = kq[i & 1];
      k4.wait();
      if constexpr (i + 1 < C::DV) kq[(i + 1) & 1].template issue<i + 1>(Ks, trf);
#pragma unroll
      for (int ss = 0; ss < 4; ss++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k4.frag(ss), df[ss], acc[i], 0, 0, 0);
    });
  }
  if (qrow < Lq) {
    bf16_t* orow = dq + ((size_t)b * Lq + qrow) * ldq + h * D;
#pragma unroll
    for (int i = 0; i < C::DV; i++)
#pragma unroll
      for (int qd = 0; qd < 4; qd++) {
        int dcol = 32 * i + 8 * qd + 4 * hi;
        if (dcol < D)
          *(uint2*)(orow + dcol) = make_uint2(pack_bf2(acc[i][4 * qd], acc[i][4 * qd + 1]), pack_bf2(acc[i][4 * qd + 2], acc[i][4 * qd + 3]));
      }
  }
}

// ============================================================================ backward: dK, dV
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* dO,
                                                            const float* lse, const float* delta, bf16_t* dk, bf16_t* dv, int H,
                                                            int Lq, int Lk, int ldq, int ldk, int ldo, float scale) {
  using C = AttnCfg<D>;
  __shared__ __attribute__((aligned(16))) char Qs[TileBytes<D>::value];
  __shared__ __attribute__((aligned(16))) char Os[TileBytes<D>::value];
  __shared__ __attribute__((aligned(16))) float L2s[64], dls[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, kv0 = blockIdx.x * 128 + wave * 32;
  const bf16_t* qb = q + (size_t)b * Lq * ldq + h * D;
  const bf16_t* dob = dO + (size_t)b * Lq * ldo + h * D;
  const bf16_t* kb = k + (size_t)b * Lk * ldk + h * D;
  const bf16_t* vb = v + (size_t)b * Lk * ldk + h * D;
  bf16x8 kf[C::DK16], vf[C::DK16];
#pragma unroll
  for (int s = 0; s < C::DK16; s++) {
    kf[s] = gfrag<D>(kb, ldk, kv0 + l31, Lk, s, hi);
    vf[s] = gfrag<D>(vb, ldk, kv0 + l31, Lk, s, hi);
  }
  f32x16 acc_k[C::DV], acc_v[C::DV];
#pragma unroll
  for (int i = 0; i < C::DV; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) { acc_k[i][r] = 0.f; acc_v[i][r] = 0.f; }
  const float sc = scale * LOG2E;
  const bool kv_ok = (kv0 + l31) < Lk;
  const bool blk_full = ((int)blockIdx.x * 128 + 128) <= Lk;
  fill_pad_chunks<D, 64>(Qs, tid, false);
  fill_pad_chunks<D, 64>(Os, tid, false);
  const TrFrag<D> trf(lane);
  const RowGeom<D, 64> geq(ldq, tid), geo(ldo, tid);
  RowStage<D, 64> qst, ost;
  float l2r = 0.f, dlr = 0.f;
  const float* lse_bh = lse + ((size_t)b * H + h) * Lq;
  const float* dl_bh = delta + ((size_t)b * H + h) * Lq;
  auto stage_load = [&](int q0_) {
    qst.load(geq, qb, ldq, q0_, Lq, tid); ost.load(geo, dob, ldo, q0_, Lq, tid);
    if (q0_ + 64 <= Lq) {          // uniform base + lane offset; only the ragged last tile clamps
      asm volatile("" ::: "memory");
      l2r = lse_bh[q0_ + (tid & 63)];
      dlr = dl_bh[q0_ + (tid & 63)];
    } else {
      int qr = q0_ + (tid & 63);
      if (qr >= Lq) qr = Lq - 1;
      l2r = lse_bh[qr];
      dlr = dl_bh[qr];
    }
  };
  if (AttnPrefetch<D>::value) stage_load(0);
  for (int qq0 = 0; qq0 < Lq; qq0 += 64) {
    __syncthreads();
    if (AttnPrefetch<D>::value) {
      qst.store(geq, Qs, Lq, tid); ost.store(geo, Os, Lq, tid);
      if (tid < 64) { L2s[tid] = -l2r; dls[tid] = -dlr * scale; }
    } else {
      load_rowmajor<D, 64>(Qs, qb, ldq, qq0, Lq, tid);
      load_rowmajor<D, 64>(Os, dob, ldo, qq0, Lq, tid);
      if (tid < 64) {
        int qr = qq0 + tid;
        L2s[tid] = qr < Lq ? -lse_bh[qr] : 0.f;
        dls[tid] = qr < Lq ? -dl_bh[qr] * scale : 0.f;
      }
    }
    __syncthreads();
    if (AttnPrefetch<D>::value && qq0 + 64 < Lq) stage_load(qq0 + 64);
    f32x16 s_[2], dp[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++) { s_[t][r] = 0.f; dp[t][r] = 0.f; }
#pragma unroll
      for (int s = 0; s < C::DK16; s++) {
        bf16x8 qfr = *(const bf16x8*)(Qs + ((32 * t + l31) * C::RKU + 2 * s + hi) * 16);
        bf16x8 ofr = *(const bf16x8*)(Os + ((32 * t + l31) * C::RKU + 2 * s + hi) * 16);
        s_[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[s], s_[t], 0, 0, 0);   // S[q][kv]
        dp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ofr, vf[s], dp[t], 0, 0, 0);   // dP[q][kv]
      }
    }
    // L2s / dls hold -lse and -delta*scale of the tile's 64 queries: the addends of the two packed fmas (see the dQ kernel)
    {
      const f32x2 sc2 = {sc, sc}, scl2 = {scale, scale};
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int ql = 32 * t + 4 * hi + (r & 3) + 8 * (r >> 2);     // r even -> ql even: the pair (ql, ql + 1) is one 8-byte LDS read
          const f32x2 nl = *(const f32x2*)&L2s[ql], nd = *(const f32x2*)&dls[ql];
          const f32x2 x = pcm_pk_fma(f32x2{s_[t][r], s_[t][r + 1]}, sc2, nl);
          const f32x2 p = {PCM_EXP2F(x[0]), PCM_EXP2F(x[1])};
          const f32x2 y = pcm_pk_fma(f32x2{dp[t][r], dp[t][r + 1]}, scl2, nd) * p;   // dS[q][kv]
          dp[t][r] = y[0]; dp[t][r + 1] = y[1];
          s_[t][r] = p[0]; s_[t][r + 1] = p[1];
        }
    }
    if (!blk_full || qq0 + 64 > Lq) {
      asm volatile("" ::: "memory");   // real branch, see the forward kernel
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          int ql = 32 * t + 4 * hi + (r & 3) + 8 * (r >> 2);
          if (!kv_ok || (qq0 + ql) >= Lq) { s_[t][r] = 0.f; dp[t][r] = 0.f; }
        }
    }
    bf16x8 pf[4], df[4];
#pragma unroll
    for (int ss = 0; ss < 4; ss++) { pf[ss] = pack_frag(s_[ss >> 1], ss & 1); df[ss] = pack_frag(dp[ss >> 1], ss & 1); }
    // dO^T and Q^T fragments, one 32-row accumulator group at a time: dO^T(i), Q^T(i) in flight while the previous group multiplies
    TrQuad<D> oq, qq;
    oq.template issue<0>(Os, trf);
    qq.template issue<0>(Qs, trf);
    pcm_static_for<0, C::DV>([&](auto it) {
      constexpr int i = decltype(it)::value;
      oq.wait(); qq.keep();
      bf16x8 of[4], qf4[4];
#pragma unroll
      for (int ss = 0; ss < 4; ss++) { of[ss] = oq.frag(ss); qf4[ss] = qq.frag(ss); }
      if constexpr (i + 1 < C::DV) { oq.template issue<i + 1>(Os, trf); qq.template issue<i + 1>(Qs, trf); }
#pragma unroll
      for (int ss = 0; ss < 4; ss++) {
        acc_v[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[ss], pf[ss], acc_v[i], 0, 0, 0);   // dV^T[d][kv]
        acc_k[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf4[ss], df[ss], acc_k[i], 0, 0, 0);  // dK^T[d][kv]
      }
    });
  }
  if (kv_ok) {
    bf16_t* krow = dk + ((size_t)b * Lk + kv0 + l31) * ldk + h * D;
    bf16_t* vrow = dv + ((size_t)b * Lk + kv0 + l31) * ldk + h * D;
#pragma unroll
    for (int i = 0; i < C::DV; i++)
#pragma unroll
      for (int qd = 0; qd < 4; qd++) {
        int dcol = 32 * i + 8 * qd + 4 * hi;
        if (dcol < D) {
          *(uint2*)(krow + dcol) = make_uint2(pack_bf2(acc_k[i][4 * qd], acc_k[i][4 * qd + 1]), pack_bf2(acc_k[i][4 * qd + 2], acc_k[i][4 * qd + 3]));
          *(uint2*)(vrow + dcol) = make_uint2(pack_bf2(acc_v[i][4 * qd], acc_v[i][4 * qd + 1]), pack_bf2(acc_v[i][4 * qd + 2], acc_v[i][4 * qd + 3]));
        }
      }
  }
}

PCM_KNOB int g_attn_xcd_remap = 0;
PCM_TOOLS_ONLY(extern "C" void pcm_debug_attn_xcd_remap(int on) { g_attn_xcd_remap = on ? 1 : 0; })
// attention_fwd.hip: the software-pipelined forward (false: no instantiation for this variant / head dim -> the kernel above runs)
bool pcm_attn_fwd_pipe_launch(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk, int d, int ldq,
                              int ldk, int ldo, float scale, void* stream);

static int attn_check(const char* what, const void* q, const void* k, const void* v, int B, int H, int Lq, int Lk, int d,
                      int ldq, int ldk, int ldo) {
  PCM_CHECK(q && k && v && B > 0 && H > 0 && Lq > 0 && Lk > 0, PCM_EINVAL, "%s: null/empty", what);
  PCM_CHECK(d == 40 || d == 80 || d == 160 || d == 32 || d == 64, PCM_EUNSUPPORTED, "%s: head_dim %d not in {32,40,64,80,160}", what, d);
  PCM_CHECK((ldq % 8) == 0 && (ldk % 8) == 0 && (ldo % 8) == 0 && PCM_ALIGNED16(q) && PCM_ALIGNED16(k) && PCM_ALIGNED16(v), PCM_EALIGN,
            "%s: strides must be %%8 and pointers 16-byte aligned", what);
  return PCM_OK;
}

#define ATTN_DISPATCH(d, CALL)          \
  switch (d) {                          \
    case 32: { CALL(32); } break;       \
    case 40: { CALL(40); } break;       \
    case 64: { CALL(64); } break;       \
    case 80: { CALL(80); } break;       \
    default: { CALL(160); } break;      \
  }

// The packed-operand pre-pass of round 1 is gone (the k-along-rows operands are LDS transpose reads of the row-major tiles), so no call
// needs a workspace any more; the *_ws entry points and the size query stay in the ABI and accept / report an unused workspace.
PCM_TOOLS_ONLY(extern "C" void pcm_debug_attn_pack_min_len(int) {})
extern "C" size_t pcm_attn_workspace_bytes(int, int, int, int, int, int) { return 0; }

extern "C" int pcm_attn_fwd_ws(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk,
                               int d, int ldq, int ldk, int ldo, float scale, void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  if (int rc = attn_check("pcm_attn_fwd", q, k, v, B, H, Lq, Lk, d, ldq, ldk, ldo)) return rc;
  PCM_CHECK(o && PCM_ALIGNED16(o), PCM_EALIGN, "pcm_attn_fwd: o");
  PCM_CHECK(B <= 65535 && H <= 65535, PCM_EUNSUPPORTED, "pcm_attn_fwd: batch / head count beyond the grid limit");
  if (pcm_attn_fwd_pipe_launch(q, k, v, o, lse, B, H, Lq, Lk, d, ldq, ldk, ldo, scale, stream)) return pcm_post_launch("pcm_attn_fwd");
  dim3 grid((Lq + 127) / 128, H, B), block(256);
#define FWD_CALL(DD)                                                                                                        \
  PCM_LAUNCH((attn_fwd_kernel<DD>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, \
             H, Lq, Lk, ldq, ldk, ldo, scale ATTN_DBG_ARG)
  ATTN_DISPATCH(d, FWD_CALL)
  return pcm_post_launch("pcm_attn_fwd");
}
extern "C" int pcm_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk,
                            int d, int ldq, int ldk, int ldo, float scale, void* stream) {
  return pcm_attn_fwd_ws(q, k, v, o, lse, B, H, Lq, Lk, d, ldq, ldk, ldo, scale, nullptr, 0, stream);
}

extern "C" int pcm_attn_bwd_ws(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                               float* delta, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int d, int ldq, int ldk,
                               int ldo, float scale, void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  if (int rc = attn_check("pcm_attn_bwd", q, k, v, B, H, Lq, Lk, d, ldq, ldk, ldo)) return rc;
  PCM_CHECK(o && dO && lse && delta && PCM_ALIGNED16(o) && PCM_ALIGNED16(dO), PCM_EALIGN, "pcm_attn_bwd: o/dO/lse/delta");
  PCM_CHECK(B <= 65535 && H <= 65535, PCM_EUNSUPPORTED, "pcm_attn_bwd: batch / head count beyond the grid limit");
  // delta: by the dQ kernel when there is one (it holds the dO rows anyway), by its own pass otherwise
  if (!dq) PCM_LAUNCH(attn_delta_kernel, dim3((Lq * H + 255) / 256, B), dim3(256), 0, stream, (const bf16_t*)o, (const bf16_t*)dO, delta, H, Lq, d, ldo);
  if (dq) {
    dim3 grid((Lq + 127) / 128, H, B), block(256);
#define DQ_CALL(DD)                                                                                                          \
  PCM_LAUNCH((attn_bwd_dq_kernel<DD>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dO, \
             lse, delta, (const bf16_t*)o, (bf16_t*)dq, H, Lq, Lk, ldq, ldk, ldo, scale)
    ATTN_DISPATCH(d, DQ_CALL)
  }
  if (dk && dv) {
    dim3 grid((Lk + 127) / 128, H, B), block(256);
#define DKV_CALL(DD)                                                                                                         \
  PCM_LAUNCH((attn_bwd_dkdv_kernel<DD>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dO, \
             lse, delta, (bf16_t*)dk, (bf16_t*)dv, H, Lq, Lk, ldq, ldk, ldo, scale)
    ATTN_DISPATCH(d, DKV_CALL)
  }
  return pcm_post_launch("pcm_attn_bwd");
}
extern "C" int pcm_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                            float* delta, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int d, int ldq, int ldk,
                            int ldo, float scale, void* stream) {
  return pcm_attn_bwd_ws(q, k, v, o, dO, lse, delta, dq, dk, dv, B, H, Lq, Lk, d, ldq, ldk, ldo, scale, nullptr, 0, stream);
}
