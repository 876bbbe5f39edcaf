// pcm-build-flags: -mllvm -amdgpu-mfma-vgpr-form
// Scaled-dot-product attention with a PRE-SCALED query (round 5): q' = q * (head_dim^-1/2 * log2 e), folded by the caller into the to_q
// projection weights (pcm_amd/model.py packs W_q and the LoRA factor s*B_q times that constant: ONE rounding of the fp32 master
// values, as before), so that the scores leave the QK^T MFMA already in the log2 domain:  S' = q' k^T.  Same tiling, LDS images and
// transpose-read operands as attention.hip (a workgroup = 4 waves x 32 query rows, 64-key tiles, S^T = K Q^T so that a lane owns one
// query row); what changes is the VALU stream of the softmax, which on this chip is what the d = 40 tile's time is made of
// (attention.hip / DESIGN section 9: 14 MFMAs carry ~96 VALU + 32 v_exp, and VALU issue does not hide under the MFMAs of the same SIMD):
//
//  1. The subtraction of the softmax reference rides the MFMA.  A head dim that is not a multiple of 16 (40 -> 48) leaves spare
//     contraction slots in the padded QK^T: the K tile carries a ONE in its pad columns D, D+1 and the lane's query fragment carries
//     -m in slot D (forward: the reference m is kept bf16-representable, so nothing is lost -- softmax is shift invariant and the LSE
//     output uses the same m) or the (hi, lo) bf16 split of -lse in slots D, D+1 (backward: 16 significand bits).  The MFMA then
//     returns s' - m: the exponentials read the accumulators directly -- no v_pk_fma per score pair.  The backward's dP - delta
//     is formed the same way (ones in V's pad columns, -delta as (hi, lo) in the dO fragment).
//  2. No running maximum after the first key tile.  The first tile fixes the reference (its true row maximum); later tiles only
//     exponentiate.  exp2(s' - m) may then exceed 1 -- by up to 2^127 before fp32 overflows, and the bf16 P / fp32 accumulators keep
//     their RELATIVE precision at any magnitude.  Rows whose later scores exceed the first tile's maximum by more than ~2^120 overflow
//     to inf: the row sum (accumulated by the PV MFMA's ones column) is checked ONCE after the loop, and a workgroup that sees a
//     non-finite sum repeats its tiles with the classic per-tile maximum tracking (lazy reference, threshold 2^8).  Real
//     activations never take that path; tests force it (tests/kernel_cases.py::case_attention, spike_overflow).
//  3. The 1/sqrt(d) of dS is folded into the epilogue (dq', dK are scaled by ln 2 once per output element).
// Per 64-key tile and wave, d = 40 forward: 14 MFMAs + 32 v_exp + 16 v_cvt_pk + staging, against + 16 v_pk_fma + 24 v_max3 + compares.
//
// Head dims without spare slots (32, 64, 80, 160) and the IEEE-half build (-DPCM_ACT_F16: -delta of loss-scaled gradients can leave the
// half range) keep the subtraction on the VALU (s' + (-m), one packed add per score pair) and gain only items 2 and 3.
// Replaces F.scaled_dot_product_attention / xformers (train_pcm_lora_sd15.py:947-957) for every attention of the UNet passes.
#include "attn_dev.h"

#define LN2 0.6931471805599453f
#ifndef PCM_ATTN_PS_DMA_DEFAULT      // (A/B builds: tools/probes/build_variant.py nodma -DPCM_ATTN_PS_DMA_DEFAULT=0)
#define PCM_ATTN_PS_DMA_DEFAULT 1
#endif

template <int D>
struct SlotCfg {
  using C = AttnCfg<D>;
#ifdef PCM_ACT_F16
  static constexpr bool ON = false;
#else
  static constexpr bool ON = (16 * C::DK16 - D) >= 2 && (D % 2) == 0;
#endif
  // column D of a row-major operand = k-step S, lane half H, elements E, E+1 of the MFMA fragment (gfrag: row-major [16s + 8hi + e])
  static constexpr int S = D / 16, H = (D % 16) / 8, E = D % 8;
};
// the two 16-bit values of slots (D, D+1) of a register fragment: only the lanes of half H hold those columns (the other half's
// elements E, E+1 of the same k-step are real data: columns D - 8 .. of the row)
template <int D>
__device__ __forceinline__ void set_slot(bf16x8 (&f)[AttnCfg<D>::DK16], int hi, unsigned two) {
  using SC = SlotCfg<D>;
  if (hi == SC::H) {
    f[SC::S][SC::E] = (short)(two & 0xffffu);
    f[SC::S][SC::E + 1] = (short)(two >> 16);
  }
}
// (hi, lo) split of a float in the 16-bit format: hi = rn(v), lo = rn(v - hi); hi in the low half-word (column D), lo in the high one
__device__ __forceinline__ unsigned split_hi_lo(float v) {
  const float h = bf2f(f2bf(v));
  return pack_bf2(h, v - h);
}
#define PCM_TWO_ONES (PCM_ONE_BITS | (PCM_ONE_BITS << 16))
// pad chunks of a tile image with the given first dword in the chunk that holds column D
template <int D, int ROWS>
__device__ __forceinline__ void fill_pad_chunks_w(char* dst, int tid, unsigned first_dword) {
  using C = AttnCfg<D>;
  constexpr int NP = C::RKU - C::DG;
  for (int u = tid; u < ROWS * NP; u += 256) {
    int r = u / NP, c = C::DG + (u - r * NP);
    *(uint4*)(dst + (r * C::RKU + c) * 16) = make_uint4(c == C::DG ? first_dword : 0u, 0u, 0u, 0u);
  }
}

// ============================================================================ forward
// DMA: the K / V tiles are staged by LDS-DMA into a double-buffered pair of images (attn_dev.h DmaTile): tile j+1 is in flight while tile j is
// computed on, ONE barrier per tile, no staging registers and no ds_write pass; otherwise the register staging of attention.hip (two barriers)
template <int D, bool DMA>
__global__ __launch_bounds__(256, (D <= 40 ? 3 : 1)) void attn_fwd_ps_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o,
                                                                              float* lse, int H, int Lq, int Lk, int ldq, int ldk, int ldo,
                                                                              int force_track) {
  using C = AttnCfg<D>;
  using SC = SlotCfg<D>;
  constexpr bool SLOT = SC::ON;
  constexpr int TB = TileBytes<D>::value, NBUF = DMA ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char KVs[NBUF * 2 * TB];      // [buffer][K, V]
  char* Ks = KVs;
  char* Vs = KVs + TB;
  __shared__ int s_redo;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128 + wave * 32;
  const bf16_t* qb = q + (size_t)b * Lq * ldq + h * D;
  const bf16_t* kb = k + (size_t)b * Lk * ldk + h * D;
  const bf16_t* vb = v + (size_t)b * Lk * ldk + h * D;
  bf16x8 qf[C::DK16];
#pragma unroll
  for (int s = 0; s < C::DK16; s++) qf[s] = gfrag<D>(qb, ldq, q0 + l31, Lq, s, hi);
  // column D of the V tile = ones: the PV MFMA also produces the softmax denominator in accumulator row D (attention.hip)
  constexpr bool ONES = C::DV * 32 > D && C::RKU > C::DG;
#pragma unroll
  for (int bf = 0; bf < NBUF; bf++) {
    fill_pad_chunks_w<D, 64>(KVs + (2 * bf) * TB, tid, SLOT ? PCM_TWO_ONES : 0u);
    fill_pad_chunks_w<D, 64>(KVs + (2 * bf + 1) * TB, tid, ONES ? (unsigned)PCM_ONE_BITS : 0u);
  }
  if (tid == 0) s_redo = 0;
  const TrFrag<D> trf(lane);
  const RowGeom<D, 64> geo(ldk, tid);
  RowStage<D, 64> kst, vst;
  const DmaTile<D, 64> dma(ldk, lane, wave);
  if constexpr (DMA) __syncthreads();     // the pad fills above are ordinary stores: they must land before a DMA'd tile next to them is read
  f32x16 acc_o[C::DV];
  float m_run = 0.f, l_run = 0.f;
  bool track = force_track != 0;         // workgroup-uniform: per-tile maximum tracking (the fallback of item 2; tests force it)
  for (int pass = 0; pass < 2; pass++) {
#pragma unroll
    for (int i = 0; i < C::DV; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc_o[i][r] = 0.f;
    m_run = 0.f; l_run = 0.f;
    if constexpr (SLOT) set_slot<D>(qf, hi, 0u);
    if constexpr (DMA) {
      dma.issue(kb, ldk, 0, Lk, KVs, wave); dma.issue(vb, ldk, 0, Lk, KVs + TB, wave);
    } else if (AttnPrefetch<D>::value) { kst.load(geo, kb, ldk, 0, Lk, tid); vst.load(geo, vb, ldk, 0, Lk, tid); }
    for (int kv0 = 0; kv0 < Lk; kv0 += 64) {
      if constexpr (DMA) {
        // this wave's pieces of tile kv0 have landed; past the barrier every wave's have, and every wave is done reading the OTHER buffer
        // (tile kv0 - 64): the next tile goes there while this one is computed on
        PCM_WAIT_VMCNT(0);
        __syncthreads();
        const int cur = (kv0 >> 6) & 1;
        Ks = KVs + (2 * cur) * TB; Vs = Ks + TB;
      } else {
        __syncthreads();
        if (AttnPrefetch<D>::value) {
          kst.store(geo, Ks, Lk, tid); vst.store(geo, Vs, Lk, tid);
        } else {
          load_rowmajor<D, 64>(Ks, kb, ldk, kv0, Lk, tid);
          load_rowmajor<D, 64>(Vs, vb, ldk, kv0, Lk, tid);
        }
        __syncthreads();
        if (AttnPrefetch<D>::value && kv0 + 64 < Lk) { kst.load(geo, kb, ldk, kv0 + 64, Lk, tid); vst.load(geo, vb, ldk, kv0 + 64, Lk, tid); }
      }
      f32x16 s_[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) s_[t][r] = 0.f;
#pragma unroll
        for (int s = 0; s < C::DK16; s++) {
          bf16x8 kf = *(const bf16x8*)(Ks + ((32 * t + l31) * C::RKU + 2 * s + hi) * 16);
          s_[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s_[t], 0, 0, 0);      // SLOT: s' - m_run;  else s'
        }
      }
      if constexpr (DMA) {
        // the next tile's DMA is issued AFTER this tile's last compiler-tracked LDS read (the K fragments above): hipcc orders every tracked
        // ds_read behind pending LDS-DMA with s_waitcnt vmcnt(0), which in front of the K reads would drain the prefetch it just issued
        // (pcm_common.h PCM_TR16_ISSUE); the V^T reads below are untracked asm reads of the CURRENT buffer
        if (kv0 + 64 < Lk) {
          char* nx = KVs + (2 * (((kv0 >> 6) & 1) ^ 1)) * TB;
          dma.issue(kb, ldk, kv0 + 64, Lk, nx, wave); dma.issue(vb, ldk, kv0 + 64, Lk, nx + TB, wave);
        }
      }
      constexpr bool V_EARLY = D <= 80;
      TrQuad<D> vq[V_EARLY ? C::DV : 1];
      if constexpr (V_EARLY) pcm_static_for<0, C::DV>([&](auto it) { vq[decltype(it)::value].template issue<decltype(it)::value>(Vs, trf); });
      if (kv0 + 64 > Lk) {
        asm volatile("" ::: "memory");   // a real (wave-uniform) branch: if-converted it costs 3 VALU ops per score in every tile
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            int kv = kv0 + 32 * t + 4 * hi + (r & 3) + 8 * (r >> 2);
            if (kv >= Lk) s_[t][r] = -1e30f;
          }
      }
      const bool first = kv0 == 0;
      if (first || track) {
        asm volatile("" ::: "memory");   // wave-uniform branch: the steady state of the fast path never enters
        float mxa[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int r = 0; r < 16; r++) mxa[r & 3] = fmaxf(mxa[r & 3], s_[t][r]);
        const float mx = pcm_xhalf_max(fmaxf(fmaxf(mxa[0], mxa[1]), fmaxf(mxa[2], mxa[3])));
        const float m_abs = SLOT ? m_run + mx : mx;               // SLOT: the accumulators are relative to m_run
        float m_new = first ? m_abs : fmaxf(m_run, m_abs);
        if (first || !__all(m_new <= m_run + 8.0f)) {
          if constexpr (SLOT) m_new = bf2f(f2bf(m_new));          // the reference must be representable in the fragment's slot
          const float delta = m_new - m_run;                      // 0 for rows that did not move (m_run is representable already)
          if (!first) {
            const float alpha = PCM_EXP2F(-delta);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < C::DV; i++)
#pragma unroll
              for (int r = 0; r < 16; r++) acc_o[i][r] *= alpha;
          }
          m_run = m_new;
          if constexpr (SLOT) {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
              for (int r = 0; r < 16; r++) s_[t][r] -= delta;     // this tile's scores were formed against the old reference
            set_slot<D>(qf, hi, (unsigned)f2bf(-m_run));
          }
        }
      }
      float psum = 0.f;
      if constexpr (SLOT) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int r = 0; r < 16; r++) s_[t][r] = PCM_EXP2F(s_[t][r]);
      } else {
        const f32x2 nm2 = {-m_run, -m_run};
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 x = f32x2{s_[t][r], s_[t][r + 1]} + nm2;
            const float p0 = PCM_EXP2F(x[0]), p1 = PCM_EXP2F(x[1]);
            s_[t][r] = p0; s_[t][r + 1] = p1;
            if (!ONES) psum += p0 + p1;
          }
      }
      l_run += psum;
      bf16x8 pf[4];
#pragma unroll
      for (int ss = 0; ss < 4; ss++) pf[ss] = pack_frag(s_[ss >> 1], ss & 1);
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
    }
    float l_chk = pcm_xhalf_sum(l_run);
    if constexpr (ONES) {
      constexpr int LOC = D % 32;
      static_assert(((LOC >> 2) & 1) == 0, "ones row must sit in the low lane half");
      l_chk = __shfl(acc_o[D / 32][(LOC & 3) + 4 * (LOC >> 3)], l31);
    }
    l_run = l_chk;
    if (track) break;
    // item 2: a row whose later scores outgrew the first tile's maximum by ~2^120 has a non-finite sum -> the workgroup repeats with tracking.
    // The check also reads every OUTPUT accumulator (round 6): where the row sum is the fp32 VALU sum (head dims without the ones column:
    // 32 / 64 / 160) P is packed to the 16-bit format AFTER that sum -- in the IEEE-half build a p above 65504 (a late score ~16 above the
    // first tile's maximum in the log2 domain) is inf in the PV MFMA while the fp32 sum stays finite, and a finite sum near the threshold
    // times |v| can overflow the accumulators on its own: the threshold is 1e30 (was 1e37), which leaves |v| * Lk 2^27 of headroom in every
    // build.  x * 0 is NaN for inf / NaN and 0 otherwise.
    float chk = l_chk;
#ifdef PCM_ACT_F16
    // (half build, sums on the VALU only: the bfloat16 build's P cannot overflow before the fp32 sum does, and reading the accumulators here
    //  cost its d = 40 forward 17 % -- 4.67 -> 5.48 ms per two-timestep forward, profiles/r06_c_* -- although the code sits behind the loop)
    if constexpr (!ONES) {
#pragma unroll
      for (int i = 0; i < C::DV; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) chk += acc_o[i][r] * 0.f;
    }
#endif
    __syncthreads();
    if (!(chk < 1e30f)) s_redo = 1;
    __syncthreads();
    if (!s_redo) break;
    track = true;
  }
  const float l_tot = l_run;
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

// ============================================================================ backward: dQ'  (gradient with respect to the pre-scaled q')
template <int D, bool DMA>
__global__ __launch_bounds__(256, (D <= 40 ? 3 : 1)) void attn_bwd_dq_ps_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* dO,
                                                                                 const float* lse, float* delta, const bf16_t* o, bf16_t* dq, int H,
                                                                                 int Lq, int Lk, int ldq, int ldk, int ldo) {
  using C = AttnCfg<D>;
  using SC = SlotCfg<D>;
  constexpr bool SLOT = SC::ON;
  constexpr int TB = TileBytes<D>::value, NBUF = DMA ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char KVs[NBUF * 2 * TB];      // [buffer][K, V]  (DMA: double-buffered, see the forward)
  char* Ks = KVs;
  char* Vs = KVs + TB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
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
  float dl;      // delta[q] = sum_d dO[q][d] * O[q][d], computed here from the fragments the lane holds (attention.hip) and published for dK/dV
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
  if constexpr (SLOT) {     // -lse and -delta as (hi, lo) pairs in the spare contraction slots, against ones in the K / V pad columns
    set_slot<D>(qf, hi, split_hi_lo(-L2));
    set_slot<D>(dof, hi, split_hi_lo(-dl));
  }
  f32x16 acc[C::DV];
#pragma unroll
  for (int i = 0; i < C::DV; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
#pragma unroll
  for (int bf = 0; bf < 2 * NBUF; bf++) fill_pad_chunks_w<D, 64>(KVs + bf * TB, tid, SLOT ? PCM_TWO_ONES : 0u);
  const TrFrag<D> trf(lane);
  const RowGeom<D, 64> geo(ldk, tid);
  RowStage<D, 64> kst, vst;
  const DmaTile<D, 64> dma(ldk, lane, wave);
  if constexpr (DMA) {
    __syncthreads();
    dma.issue(kb, ldk, 0, Lk, KVs, wave); dma.issue(vb, ldk, 0, Lk, KVs + TB, wave);
  } else if (AttnPrefetch<D>::value) { kst.load(geo, kb, ldk, 0, Lk, tid); vst.load(geo, vb, ldk, 0, Lk, tid); }
  const f32x2 nl2 = {-L2, -L2}, ndl2 = {-dl, -dl};
  for (int kv0 = 0; kv0 < Lk; kv0 += 64) {
    if constexpr (DMA) {
      PCM_WAIT_VMCNT(0);
      __syncthreads();
      Ks = KVs + (2 * ((kv0 >> 6) & 1)) * TB; Vs = Ks + TB;
    } else {
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
        s_[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s_[t], 0, 0, 0);     // SLOT: s' - lse
        dp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s], dp[t], 0, 0, 0);    // SLOT: dP - delta
      }
    }
    if constexpr (DMA) {      // after the tile's last compiler-tracked LDS reads (see the forward); the K^T reads below are untracked asm reads
      if (kv0 + 64 < Lk) {
        char* nx = KVs + (2 * (((kv0 >> 6) & 1) ^ 1)) * TB;
        dma.issue(kb, ldk, kv0 + 64, Lk, nx, wave); dma.issue(vb, ldk, kv0 + 64, Lk, nx + TB, wave);
      }
    }
    // dS'^T = p * (dP - delta): the 1/sqrt(d) (times ln 2 for the log2-domain q') multiplies the output once, in the epilogue
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 x = {s_[t][r], s_[t][r + 1]}, g = {dp[t][r], dp[t][r + 1]};
        if constexpr (!SLOT) { x = x + nl2; g = g + ndl2; }
        const f32x2 p = {PCM_EXP2F(x[0]), PCM_EXP2F(x[1])};
        const f32x2 y = g * p;
        s_[t][r] = y[0]; s_[t][r + 1] = y[1];
      }
    if (kv0 + 64 > Lk) {
      asm volatile("" ::: "memory");
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
          *(uint2*)(orow + dcol) = make_uint2(pack_bf2(acc[i][4 * qd] * LN2, acc[i][4 * qd + 1] * LN2),
                                              pack_bf2(acc[i][4 * qd + 2] * LN2, acc[i][4 * qd + 3] * LN2));
      }
  }
}

// ============================================================================ backward: dK, dV
// DMA (only with the slot trick, i.e. d = 40 in the bfloat16 build: without it the softmax reads the -lse / -delta side arrays with tracked
// LDS reads late in the tile, behind which hipcc would drain a DMA issued earlier): Q / dO tiles by LDS-DMA into a double buffer, the
// per-row (-lse, -delta) pairs written into the NEXT buffer's pad columns one tile ahead.
#ifndef PCM_ATTN_DKDV_WAVES      // minimum waves per SIMD of the d <= 40 instantiations (A/B builds: -DPCM_ATTN_DKDV_WAVES=3 spills 38 registers)
#define PCM_ATTN_DKDV_WAVES 1
#endif
template <int D, bool DMA_>
__global__ __launch_bounds__(256, (D <= 40 ? PCM_ATTN_DKDV_WAVES : 1)) void attn_bwd_dkdv_ps_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* dO,
                                                               const float* lse, const float* delta, bf16_t* dk, bf16_t* dv, int H, int Lq,
                                                               int Lk, int ldq, int ldk, int ldo) {
  using C = AttnCfg<D>;
  using SC = SlotCfg<D>;
  constexpr bool SLOT = SC::ON;
  constexpr bool DMA = DMA_ && SLOT;
  constexpr int TB = TileBytes<D>::value, NBUF = DMA ? 2 : 1;
  __shared__ __attribute__((aligned(16))) char QOs[NBUF * 2 * TB];      // [buffer][Q, dO]
  char* Qs = QOs;
  char* Os = QOs + TB;
  __shared__ __attribute__((aligned(16))) float L2s[64], dls[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
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
  if constexpr (SLOT) {     // ones against the (-lse) / (-delta) pairs the Q / dO tile rows carry in their pad columns D, D+1
    set_slot<D>(kf, hi, PCM_TWO_ONES);
    set_slot<D>(vf, hi, PCM_TWO_ONES);
  }
  f32x16 acc_k[C::DV], acc_v[C::DV];
#pragma unroll
  for (int i = 0; i < C::DV; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) { acc_k[i][r] = 0.f; acc_v[i][r] = 0.f; }
  const bool kv_ok = (kv0 + l31) < Lk;
  const bool blk_full = ((int)blockIdx.x * 128 + 128) <= Lk;
#pragma unroll
  for (int bf = 0; bf < 2 * NBUF; bf++) fill_pad_chunks<D, 64>(QOs + bf * TB, tid, false);
  const TrFrag<D> trf(lane);
  const RowGeom<D, 64> geq(ldq, tid), geo(ldo, tid);
  RowStage<D, 64> qst, ost;
  float l2r = 0.f, dlr = 0.f;
  const float* lse_bh = lse + ((size_t)b * H + h) * Lq;
  const float* dl_bh = delta + ((size_t)b * H + h) * Lq;
  auto stage_load = [&](int q0_) {
    qst.load(geq, qb, ldq, q0_, Lq, tid); ost.load(geo, dob, ldo, q0_, Lq, tid);
    if (q0_ + 64 <= Lq) {
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
  // per-row -lse / -delta of a staged tile: SLOT -> (hi, lo) pairs into the tile rows' pad columns; else the fp32 side arrays
  auto publish_rows = [&](char* qdst, char* odst, float l2v, float dlv) {
    if (tid < 64) {
      if constexpr (SLOT) {
        *(unsigned*)(qdst + (tid * C::RKU + C::DG) * 16) = split_hi_lo(-l2v);
        *(unsigned*)(odst + (tid * C::RKU + C::DG) * 16) = split_hi_lo(-dlv);
      } else {
        L2s[tid] = -l2v; dls[tid] = -dlv;
      }
    }
  };
  auto rows_load = [&](int q0_) {          // -lse / -delta of the 64 rows of tile q0_ (clamped on the ragged last tile)
    int qr = q0_ + (tid & 63);
    if (qr >= Lq) qr = Lq - 1;
    l2r = lse_bh[qr];
    dlr = dl_bh[qr];
  };
  const DmaTile<D, 64> dmaq(ldq, lane, wave), dmao(ldo, lane, wave);
  if constexpr (DMA) {
    rows_load(0);
    __syncthreads();                       // the pad fills above (zeros, by other threads) before the row pairs that go into the same chunks
    publish_rows(QOs, QOs + TB, l2r, dlr);
    __syncthreads();                       // ... and both before any DMA'd tile is read
    dmaq.issue(qb, ldq, 0, Lq, QOs, wave); dmao.issue(dob, ldo, 0, Lq, QOs + TB, wave);
    if (64 < Lq) rows_load(64);
  } else if (AttnPrefetch<D>::value) stage_load(0);
  for (int qq0 = 0; qq0 < Lq; qq0 += 64) {
    if constexpr (DMA) {
      PCM_WAIT_VMCNT(0);
      __syncthreads();
      Qs = QOs + (2 * ((qq0 >> 6) & 1)) * TB; Os = Qs + TB;
    } else {
      __syncthreads();
      if (AttnPrefetch<D>::value) {
        qst.store(geq, Qs, Lq, tid); ost.store(geo, Os, Lq, tid);
        publish_rows(Qs, Os, l2r, dlr);
      } else {
        load_rowmajor<D, 64>(Qs, qb, ldq, qq0, Lq, tid);
        load_rowmajor<D, 64>(Os, dob, ldo, qq0, Lq, tid);
        const int qr = qq0 + tid;
        publish_rows(Qs, Os, (tid < 64 && qr < Lq) ? lse_bh[qr] : 0.f, (tid < 64 && qr < Lq) ? dl_bh[qr] : 0.f);
      }
      __syncthreads();
      if (AttnPrefetch<D>::value && qq0 + 64 < Lq) stage_load(qq0 + 64);
    }
    f32x16 s_[2], dp[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++) { s_[t][r] = 0.f; dp[t][r] = 0.f; }
#pragma unroll
      for (int s = 0; s < C::DK16; s++) {
        bf16x8 qfr = *(const bf16x8*)(Qs + ((32 * t + l31) * C::RKU + 2 * s + hi) * 16);
        bf16x8 ofr = *(const bf16x8*)(Os + ((32 * t + l31) * C::RKU + 2 * s + hi) * 16);
        s_[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[s], s_[t], 0, 0, 0);   // S'[q][kv]  (SLOT: - lse[q])
        dp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ofr, vf[s], dp[t], 0, 0, 0);   // dP[q][kv]  (SLOT: - delta[q])
      }
    }
    if constexpr (DMA) {
      // the OTHER buffer is free (every wave is past this tile's barrier, i.e. done with the previous tile): first the next tile's row
      // pairs (ordinary LDS stores, values loaded one tile ago), THEN its DMA -- a tracked LDS access after a DMA issue would make hipcc
      // drain it -- then the global loads of the row values one tile further
      if (qq0 + 64 < Lq) {
        char* nx = QOs + (2 * (((qq0 >> 6) & 1) ^ 1)) * TB;
        publish_rows(nx, nx + TB, l2r, dlr);
        dmaq.issue(qb, ldq, qq0 + 64, Lq, nx, wave); dmao.issue(dob, ldo, qq0 + 64, Lq, nx + TB, wave);
        if (qq0 + 128 < Lq) rows_load(qq0 + 128);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 x = {s_[t][r], s_[t][r + 1]}, g = {dp[t][r], dp[t][r + 1]};
        if constexpr (!SLOT) {
          const int ql = 32 * t + 4 * hi + (r & 3) + 8 * (r >> 2);     // r even -> ql even: the pair (ql, ql + 1) is one 8-byte LDS read
          x = x + *(const f32x2*)&L2s[ql];
          g = g + *(const f32x2*)&dls[ql];
        }
        const f32x2 p = {PCM_EXP2F(x[0]), PCM_EXP2F(x[1])};
        const f32x2 y = g * p;                                          // dS'[q][kv] (unscaled)
        dp[t][r] = y[0]; dp[t][r + 1] = y[1];
        s_[t][r] = p[0]; s_[t][r + 1] = p[1];
      }
    if (!blk_full || qq0 + 64 > Lq) {
      asm volatile("" ::: "memory");
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
        acc_v[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[ss], pf[ss], acc_v[i], 0, 0, 0);   // dV^T[d][kv]  (rows >= D: never stored)
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
          *(uint2*)(krow + dcol) = make_uint2(pack_bf2(acc_k[i][4 * qd] * LN2, acc_k[i][4 * qd + 1] * LN2),
                                              pack_bf2(acc_k[i][4 * qd + 2] * LN2, acc_k[i][4 * qd + 3] * LN2));
          *(uint2*)(vrow + dcol) = make_uint2(pack_bf2(acc_v[i][4 * qd], acc_v[i][4 * qd + 1]), pack_bf2(acc_v[i][4 * qd + 2], acc_v[i][4 * qd + 3]));
        }
      }
  }
}

// Explicit instantiations of every (head dim, staging) pair the launchers below name.  (hipcc 7.2 emitted no host-side launch stub for some
// of the DMA-staged kernels when they were only instantiated implicitly through the launch macros: "undefined symbol __device_stub__..."
// at dlopen; the explicit form is emitted unconditionally.)
#define PCM_ATTN_PS_INST(DD, DM)                                                                                                              \
  template __global__ void attn_fwd_ps_kernel<DD, DM>(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, float*, int, int, int, int, int, int, int); \
  template __global__ void attn_bwd_dq_ps_kernel<DD, DM>(const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, const float*, float*,     \
                                                         const bf16_t*, bf16_t*, int, int, int, int, int, int);                                \
  template __global__ void attn_bwd_dkdv_ps_kernel<DD, DM>(const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, const float*,           \
                                                           const float*, bf16_t*, bf16_t*, int, int, int, int, int, int);
#if PCM_HAS_TOOLS
PCM_ATTN_PS_INST(32, true) PCM_ATTN_PS_INST(40, true) PCM_ATTN_PS_INST(64, true) PCM_ATTN_PS_INST(80, true)
PCM_ATTN_PS_INST(32, false) PCM_ATTN_PS_INST(40, false) PCM_ATTN_PS_INST(64, false) PCM_ATTN_PS_INST(80, false) PCM_ATTN_PS_INST(160, false)
#else
PCM_ATTN_PS_INST(32, (PCM_ATTN_PS_DMA_DEFAULT != 0)) PCM_ATTN_PS_INST(40, (PCM_ATTN_PS_DMA_DEFAULT != 0)) PCM_ATTN_PS_INST(64, (PCM_ATTN_PS_DMA_DEFAULT != 0))
PCM_ATTN_PS_INST(80, (PCM_ATTN_PS_DMA_DEFAULT != 0)) PCM_ATTN_PS_INST(160, false)
#endif

// ============================================================================ C ABI
bool pcm_attn_fwd_pipe_launch(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk, int d, int ldq,
                              int ldk, int ldo, float scale, void* stream);
PCM_KNOB int g_attn_ps_track = 0;     // tests: 1 = run the forward with per-tile maximum tracking from the start (the fallback path of item 2)
PCM_TOOLS_ONLY(extern "C" void pcm_debug_attn_ps_track(int on) { g_attn_ps_track = on ? 1 : 0; })
// K / V staging of the forward: 1 = LDS-DMA double buffer (head dims <= 80), 0 = register staging; A/B hook in the tools build
PCM_KNOB int g_attn_ps_dma = PCM_ATTN_PS_DMA_DEFAULT;
PCM_TOOLS_ONLY(extern "C" void pcm_debug_attn_ps_dma(int on) { g_attn_ps_dma = on ? 1 : 0; })

static int attn_ps_check(const char* what, const void* q, const void* k, const void* v, int B, int H, int Lq, int Lk, int d, int ldq, int ldk,
                         int ldo) {
  PCM_CHECK(q && k && v && B > 0 && H > 0 && Lq > 0 && Lk > 0, PCM_EINVAL, "%s: null/empty", what);
  PCM_CHECK(d == 40 || d == 80 || d == 160 || d == 32 || d == 64, PCM_EUNSUPPORTED, "%s: head_dim %d not in {32,40,64,80,160}", what, d);
  PCM_CHECK((ldq % 8) == 0 && (ldk % 8) == 0 && (ldo % 8) == 0 && PCM_ALIGNED16(q) && PCM_ALIGNED16(k) && PCM_ALIGNED16(v), PCM_EALIGN,
            "%s: strides must be %%8 and pointers 16-byte aligned", what);
  PCM_CHECK(B <= 65535 && H <= 65535, PCM_EUNSUPPORTED, "%s: batch / head count beyond the grid limit", what);
  return PCM_OK;
}
#define ATTN_PS_DISPATCH(d, CALL)       \
  switch (d) {                          \
    case 32: { CALL(32); } break;       \
    case 40: { CALL(40); } break;       \
    case 64: { CALL(64); } break;       \
    case 80: { CALL(80); } break;       \
    default: { CALL(160); } break;      \
  }

extern "C" int pcm_attn_fwd_prescaled(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk, int d,
                                      int ldq, int ldk, int ldo, void* stream) {
  if (int rc = attn_ps_check("pcm_attn_fwd_prescaled", q, k, v, B, H, Lq, Lk, d, ldq, ldk, ldo)) return rc;
  PCM_CHECK(o && PCM_ALIGNED16(o), PCM_EALIGN, "pcm_attn_fwd_prescaled: o");
  // head dims without a spare contraction slot keep the software-pipelined forward where it measured faster (attention_fwd.hip: d = 64 from
  // 32 key tiles); its score scale is then 1: scale * log2(e) = 1
  if (!SlotCfg<64>::ON && d == 64 && !g_attn_ps_track &&
      pcm_attn_fwd_pipe_launch(q, k, v, o, lse, B, H, Lq, Lk, d, ldq, ldk, ldo, 1.0f / LOG2E, stream))
    return pcm_post_launch("pcm_attn_fwd_prescaled");
  dim3 grid((Lq + 127) / 128, H, B), block(256);
#define FWD_PS_CALL_(DD, DM)                                                                                                            \
  PCM_LAUNCH((attn_fwd_ps_kernel<DD, DM>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse, \
             H, Lq, Lk, ldq, ldk, ldo, g_attn_ps_track)
#if PCM_HAS_TOOLS
#define FWD_PS_CALL(DD) if (g_attn_ps_dma && DD <= 80) { FWD_PS_CALL_(DD, (DD <= 80)); } else { FWD_PS_CALL_(DD, false); }
#else
#define FWD_PS_CALL(DD) FWD_PS_CALL_(DD, (PCM_ATTN_PS_DMA_DEFAULT && DD <= 80))
#endif
  ATTN_PS_DISPATCH(d, FWD_PS_CALL)
  return pcm_post_launch("pcm_attn_fwd_prescaled");
}

extern "C" int pcm_attn_bwd_prescaled(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse, float* delta,
                                      void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int d, int ldq, int ldk, int ldo, void* stream) {
  if (int rc = attn_ps_check("pcm_attn_bwd_prescaled", q, k, v, B, H, Lq, Lk, d, ldq, ldk, ldo)) return rc;
  PCM_CHECK(o && dO && lse && delta && dq && PCM_ALIGNED16(o) && PCM_ALIGNED16(dO), PCM_EALIGN, "pcm_attn_bwd_prescaled: o/dO/lse/delta/dq");
  {
    dim3 grid((Lq + 127) / 128, H, B), block(256);
#define DQ_PS_CALL_(DD, DM)                                                                                                        \
  PCM_LAUNCH((attn_bwd_dq_ps_kernel<DD, DM>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dO, \
             lse, delta, (const bf16_t*)o, (bf16_t*)dq, H, Lq, Lk, ldq, ldk, ldo)
#if PCM_HAS_TOOLS
#define DQ_PS_CALL(DD) if (g_attn_ps_dma && DD <= 80) { DQ_PS_CALL_(DD, (DD <= 80)); } else { DQ_PS_CALL_(DD, false); }
#else
#define DQ_PS_CALL(DD) DQ_PS_CALL_(DD, (PCM_ATTN_PS_DMA_DEFAULT && DD <= 80))
#endif
    ATTN_PS_DISPATCH(d, DQ_PS_CALL)
  }
  if (dk && dv) {
    dim3 grid((Lk + 127) / 128, H, B), block(256);
#define DKV_PS_CALL_(DD, DM)                                                                                                       \
  PCM_LAUNCH((attn_bwd_dkdv_ps_kernel<DD, DM>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dO, \
             lse, delta, (bf16_t*)dk, (bf16_t*)dv, H, Lq, Lk, ldq, ldk, ldo)
#if PCM_HAS_TOOLS
#define DKV_PS_CALL(DD) if (g_attn_ps_dma && DD <= 80) { DKV_PS_CALL_(DD, (DD <= 80)); } else { DKV_PS_CALL_(DD, false); }
#else
#define DKV_PS_CALL(DD) DKV_PS_CALL_(DD, (PCM_ATTN_PS_DMA_DEFAULT && DD <= 80))
#endif
    ATTN_PS_DISPATCH(d, DKV_PS_CALL)
  }
  return pcm_post_launch("pcm_attn_bwd_prescaled");
}
