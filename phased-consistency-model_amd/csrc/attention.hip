// Scaled-dot-product attention over latent tokens for the SD1.5 UNet (8 heads, head_dim 40/80/160,
// Lq in {4096,1024,256,64}, Lk = Lq (self) or 77 (text)), flash-style: scores never leave the CU.
//
// Forward / dQ kernels: a workgroup owns 128 query rows (4 waves x 32) and streams 64-key tiles.
//   S^T = K Q^T is computed "swapped" (MFMA A = K tile from LDS, B = Q fragments in registers), so a
//   lane holds 16+16 scores of ONE query row: row max/sum need a single cross-half shuffle.
//   P^T feeds the second MFMA (O^T = V^T P^T) straight from the accumulator registers: the MFMA
//   contraction slot (hi, e) of step ss is DEFINED as key 16ss + 8(e>>2) + 4hi + (e&3) — exactly
//   what the lane already holds — and the V tile is staged transposed into LDS in that same key
//   order (8x8 register transposes), so no permlane / LDS round trip for P.
// dK/dV kernel: a workgroup owns 128 key rows and streams 64-query tiles with the roles swapped
//   (S = Q K^T, lane = one key), the same slot trick on the query index.
// head_dim 40 is padded to 48 for QK^T (K-step 16) and to 64 for the PV tile (32-row output tiles);
// padded rows of the accumulators are never stored.
#include "pcm_common.h"

#define LOG2E 1.4426950408889634f

template <int D>
struct AttnCfg {
  static constexpr int DK16 = (D + 15) / 16;     // QK^T K-steps of 16
  static constexpr int KCH = 2 * DK16;           // 16-B chunks per row-major row (incl. zero pad)
  static constexpr int RKU = KCH | 1;            // row stride in 16-B units (odd -> conflict-free ds_read_b128)
  static constexpr int DV = (D + 31) / 32;       // 32-row tiles of the transposed output
  static constexpr int DG = D / 8;               // 8-wide column groups
};

// row-major tile image: [rows][RKU*16 B]; chunk c of row r at (r*RKU + c)*16
// transposed tile image: [d rows][64 slots] bf16 = 8 chunks of 16 B, XOR-swizzled
__device__ __forceinline__ int tr_off(int drow, int chunk) { return drow * 128 + ((chunk ^ ((drow >> 1) & 7)) << 4); }

// load a [rows x D] row-major tile (row stride ld elements) into the padded row-major LDS image
template <int D, int ROWS>
__device__ __forceinline__ void load_rowmajor(char* dst, const bf16_t* src, int ld, int row0, int nrows_valid, int tid) {
  using C = AttnCfg<D>;
  for (int u = tid; u < ROWS * C::DG; u += 256) {
    int r = u / C::DG, c = u - r * C::DG;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row0 + r < nrows_valid) v = *(const uint4*)(src + (size_t)(row0 + r) * ld + 8 * c);
    *(uint4*)(dst + (r * C::RKU + c) * 16) = v;
  }
}
template <int D, int ROWS>
__device__ __forceinline__ void zero_pad_chunks(char* dst, int tid) {
  using C = AttnCfg<D>;
  constexpr int NP = C::KCH - C::DG;
  if (NP > 0)
    for (int u = tid; u < ROWS * NP; u += 256) {
      int r = u / NP, c = C::DG + (u - r * NP);
      *(uint4*)(dst + (r * C::RKU + c) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
}
// load a [64 x D] tile TRANSPOSED into the [d][64-slot] image; slot (ss, hi, e) <-> row 16ss+8(e>>2)+4hi+(e&3)
template <int D>
__device__ __forceinline__ void load_transposed64(char* dst, const bf16_t* src, int ld, int row0, int nrows_valid, int tid) {
  using C = AttnCfg<D>;
  if (tid < 8 * C::DG) {
    int sh = tid & 7, dg = tid >> 3;  // sh = 2*ss + hi
    int ss = sh >> 1, hi = sh & 1;
    uint4 rr[8], oo[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      int r = row0 + 16 * ss + 8 * (e >> 2) + 4 * hi + (e & 3);
      rr[e] = r < nrows_valid ? *(const uint4*)(src + (size_t)r * ld + 8 * dg) : make_uint4(0u, 0u, 0u, 0u);
    }
    transpose8x8_bf16(rr, oo);
#pragma unroll
    for (int dd = 0; dd < 8; dd++) *(uint4*)(dst + tr_off(8 * dg + dd, sh)) = oo[dd];
  }
}
// ---- register staging (issue the NEXT tile's global loads before computing on the current tile;
// the LDS write happens after the next barrier, so L2/HBM latency hides under the MFMA phase) ----
// Loads are UNCONDITIONAL from clamped addresses (a predicated load + zero select makes hipcc wait
// vmcnt(0) right after issue: WAW on the destination); out-of-range rows are zeroed at store time.
template <int D, int ROWS>
struct RowStage {
  using C = AttnCfg<D>;
  static constexpr int N = (ROWS * C::DG + 255) / 256;
  uint4 r[N];
  int row0_;
  __device__ __forceinline__ void load(const bf16_t* src, int ld, int row0, int nrows_valid, int tid) {
    row0_ = row0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      int u = tid + 256 * i;
      if (u >= ROWS * C::DG) u = ROWS * C::DG - 1;
      int rr = u / C::DG, c = u - rr * C::DG;
      int row = row0 + rr;
      if (row >= nrows_valid) row = nrows_valid - 1;
      r[i] = *(const uint4*)(src + (size_t)row * ld + 8 * c);
    }
  }
  // full (wave-uniform): every row of the tile is valid -> no per-piece validity selects (the common case: only the last
  // tile of a ragged sequence is partial)
  __device__ __forceinline__ void store(char* dst, int nrows_valid, int tid) const {
    const bool full = row0_ + ROWS <= nrows_valid;
#pragma unroll
    for (int i = 0; i < N; i++) {
      int u = tid + 256 * i;
      int rr = u / C::DG, c = u - rr * C::DG;
      if (u < ROWS * C::DG) {
        uint4 v = r[i];
        if (!full && row0_ + rr >= nrows_valid) v = make_uint4(0u, 0u, 0u, 0u);
        *(uint4*)(dst + (rr * C::RKU + c) * 16) = v;
      }
    }
  }
};
// transposing stage in 4x4 blocks: a unit = 4 consecutive rows x 4 columns (4 x 8-B loads, 8 VGPRs),
// transposed in registers and written as 4 x ds_write_b64.  Row group rg = rows 4rg..4rg+3 maps to
// slot chunk 2*ss+hi with ss = rg>>2, hi = rg&1 and element half (rg>>1)&1 (same slot order as above).
template <int D>
struct TransStage {
  using C = AttnCfg<D>;
  static constexpr int DQ = D / 4;
  static constexpr int N = (16 * DQ + 255) / 256;
  uint2 rr[N][4];
  int row0_;
  __device__ __forceinline__ void load(const bf16_t* src, int ld, int row0, int nrows_valid, int tid) {
    row0_ = row0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      int u = tid + 256 * i;
      if (u >= 16 * DQ) u = 16 * DQ - 1;
      int dq = u % DQ, rg = u / DQ;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int r = row0 + 4 * rg + j;
        if (r >= nrows_valid) r = nrows_valid - 1;
        rr[i][j] = *(const uint2*)(src + (size_t)r * ld + 4 * dq);
      }
    }
  }
  __device__ __forceinline__ void store(char* dst, int nrows_valid, int tid) const {
    const bool full = row0_ + 64 <= nrows_valid;
#pragma unroll
    for (int i = 0; i < N; i++) {
      int u = tid + 256 * i;
      if (u < 16 * DQ) {
        int dq = u % DQ, rg = u / DQ;
        int chunk = 2 * (rg >> 2) + (rg & 1), half = (rg >> 1) & 1;
        unsigned a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          bool ok = full || row0_ + 4 * rg + j < nrows_valid;
          a[j] = ok ? rr[i][j].x : 0u; b[j] = ok ? rr[i][j].y : 0u;
        }
        uint2 o[4];
        o[0] = make_uint2((a[0] & 0xffffu) | (a[1] << 16), (a[2] & 0xffffu) | (a[3] << 16));
        o[1] = make_uint2((a[0] >> 16) | (a[1] & 0xffff0000u), (a[2] >> 16) | (a[3] & 0xffff0000u));
        o[2] = make_uint2((b[0] & 0xffffu) | (b[1] << 16), (b[2] & 0xffffu) | (b[3] << 16));
        o[3] = make_uint2((b[0] >> 16) | (b[1] & 0xffff0000u), (b[2] >> 16) | (b[3] & 0xffff0000u));
#pragma unroll
        for (int c = 0; c < 4; c++) *(uint2*)(dst + tr_off(4 * dq + c, chunk) + 8 * half) = o[c];
      }
    }
  }
};
template <int D> struct AttnPrefetch { static constexpr bool value = D <= 80; };

// A workgroup re-stages every 64-row tile of the streamed operand, so the 4x4 register transposes of TransStage are repeated by
// all L/128 workgroups of a (batch, head).  With a PACKED operand (pcm_attn_pack_t: the [d][64-slot] tile images written once to
// HBM, tile-major, tail rows zeroed) the stage is a straight 16-B copy into the swizzled LDS image.
template <int D>
struct PackedTStage {
  typedef unsigned pk_u32x4 __attribute__((ext_vector_type(4)));
  static constexpr int NCH = D * 8;                 // 16-B chunks of one tile image (D rows x 128 B)
  static constexpr int N = (NCH + 255) / 256;
  pk_u32x4 r[N];
  __device__ __forceinline__ void load(const bf16_t* packed_bh, int row0, int tid) {
    const pk_u32x4* t = (const pk_u32x4*)(packed_bh + (size_t)(row0 >> 6) * D * 64);
#pragma unroll
    for (int i = 0; i < N; i++) {
      int u = tid + 256 * i;
      u = u < NCH ? u : NCH - 1;
      r[i] = t[u];
    }
  }
  __device__ __forceinline__ void store(char* dst, int tid) const {
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int u = tid + 256 * i;
      if (u < NCH) *(pk_u32x4*)(dst + tr_off(u >> 3, u & 7)) = r[i];
    }
  }
};
// one interface for both: PK selects the packed copy, otherwise the transposing register stage (separate types: a struct holding
// both stages is not promoted to registers by hipcc and the prefetched tile ends up in scratch)
template <int D>
struct TransStageA : TransStage<D> {
  __device__ __forceinline__ void load(const bf16_t* src, int ld, const bf16_t*, int row0, int nrows_valid, int tid) { TransStage<D>::load(src, ld, row0, nrows_valid, tid); }
};
template <int D>
struct PackedTStageA : PackedTStage<D> {
  __device__ __forceinline__ void load(const bf16_t*, int, const bf16_t* packed_bh, int row0, int, int tid) { PackedTStage<D>::load(packed_bh, row0, tid); }
  __device__ __forceinline__ void store(char* dst, int, int tid) const { PackedTStage<D>::store(dst, tid); }
};
template <int D, bool PK> struct TStageSel { using type = TransStageA<D>; };
template <int D> struct TStageSel<D, true> { using type = PackedTStageA<D>; };
template <int D, bool PK> using TStage = typename TStageSel<D, PK>::type;

// x [B][L][ld] (head h at columns h*D..) -> packed transposed tiles xt[(b*H+h)][tile][D][64 slots]; slot 8*chunk+e of a tile is row
// 16*(chunk>>1) + 8*(e>>2) + 4*(chunk&1) + (e&3) (the contraction-slot order of the PV / dS MFMAs); rows >= L are zero.
__global__ __launch_bounds__(256) void attn_pack_t_kernel(const bf16_t* x, bf16_t* xt, int H, int L, int D, int ld, int HG) {
  // one block = one 64-row tile of a GROUP of HG heads (blockIdx.z; the group is sized by the launcher so that the LDS image stays under
  // 64 KB for any H*D): coalesced 16-B row loads into LDS, transposed 2-byte reads out of LDS, 16-B tile-image stores
  PCM_DYN_SMEM(sm);
  const int tile = blockIdx.x, b = blockIdx.y, nt = gridDim.x, h0 = blockIdx.z * HG;
  const int nh = (H - h0 < HG ? H - h0 : HG), HD = nh * D, CV = HD / 8, RS = HG * D + 8;   // RS: padded LDS row (elements)
  bf16_t* t = (bf16_t*)sm;
  const bf16_t* xb = x + (size_t)b * L * ld + h0 * D;
  for (int u = threadIdx.x; u < 64 * CV; u += blockDim.x) {
    const int r = u / CV, c = u - r * CV, row = 64 * tile + r;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < L) v = *(const uint4*)(xb + (size_t)row * ld + 8 * c);
    *(uint4*)(t + r * RS + 8 * c) = v;
  }
  __syncthreads();
  for (int u = threadIdx.x; u < HD * 8; u += blockDim.x) {
    const int col = u % HD, chunk = u / HD;                 // lanes along the columns: conflict-free 2-byte LDS reads
    const int hl = col / D, drow = col - hl * D;
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int r = 16 * (chunk >> 1) + 8 * (e >> 2) + 4 * (chunk & 1) + (e & 3);
      w[e >> 1] |= (unsigned)t[r * RS + col] << (16 * (e & 1));
    }
    uint4* out = (uint4*)(xt + ((size_t)(b * H + h0 + hl) * nt + tile) * D * 64);
    out[drow * 8 + chunk] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}


// fragment straight from global: row-major [row][16s + 8hi ..]; zero outside [0, D) / invalid rows
template <int D>
__device__ __forceinline__ bf16x8 gfrag(const bf16_t* base, int ld, int row, int nrows_valid, int s, int hi) {
  int c = 16 * s + 8 * hi;
  bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  if (row < nrows_valid && c < D) return *(const bf16x8*)(base + (size_t)row * ld + c);
  return z;
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 pack_frag(const f32x16& p, int half) {
  u32x4_t w = {pack_bf2(p[8 * half + 0], p[8 * half + 1]), pack_bf2(p[8 * half + 2], p[8 * half + 3]),
               pack_bf2(p[8 * half + 4], p[8 * half + 5]), pack_bf2(p[8 * half + 6], p[8 * half + 7])};
  return __builtin_bit_cast(bf16x8, w);
}

// ============================================================================ forward
template <int D, bool PK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* vt, bf16_t* o,
                                                       float* lse, int H, int Lq, int Lk, int ldq, int ldk, int ldo, float scale) {
  using C = AttnCfg<D>;
  __shared__ __attribute__((aligned(16))) char Ks[64 * C::RKU * 16];
  __shared__ __attribute__((aligned(16))) char Vt[C::DV * 32 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128 + wave * 32;
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
  zero_pad_chunks<D, 64>(Ks, tid);
  // head dims whose transposed V image has a spare padded row (40 -> 64, 80 -> 96): row D is set to ones, so the PV MFMA
  // also produces the softmax denominator sum_k p[k] in accumulator row D (rescaled with O for free, no VALU row sums)
  constexpr bool ONES = C::DV * 32 > D;
  if (ONES && tid < 8) *(uint4*)(Vt + tr_off(D, tid)) = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  RowStage<D, 64> kst;
  TStage<D, PK> vst;
  const bf16_t* vtb = PK ? vt + (size_t)(b * H + h) * ((Lk + 63) >> 6) * D * 64 : nullptr;
  if (AttnPrefetch<D>::value) { kst.load(kb, ldk, 0, Lk, tid); vst.load(vb, ldk, vtb, 0, Lk, tid); }
  for (int kv0 = 0; kv0 < Lk; kv0 += 64) {
    __syncthreads();
    if (AttnPrefetch<D>::value) {
      kst.store(Ks, Lk, tid); vst.store(Vt, Lk, tid);
    } else {
      load_rowmajor<D, 64>(Ks, kb, ldk, kv0, Lk, tid);
      load_transposed64<D>(Vt, vb, ldk, kv0, Lk, tid);
    }
    __syncthreads();
    if (AttnPrefetch<D>::value && kv0 + 64 < Lk) { kst.load(kb, ldk, kv0 + 64, Lk, tid); vst.load(vb, ldk, vtb, kv0 + 64, Lk, tid); }
    f32x16 s_[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++) s_[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < C::DK16; s++) {
        bf16x8 kf = *(const bf16x8*)(Ks + ((32 * t + l31) * C::RKU + 2 * s + hi) * 16);
        s_[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s_[t], 0, 0, 0);
      }
    }
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
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, s_[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float m_new = fmaxf(m_run, mx * sc);
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float p = PCM_EXP2F(fmaf(s_[t][r], sc, -m_new));
        s_[t][r] = p;
        if (!ONES) psum += p;
      }
    if (__all(m_new == m_run)) {      // running max unchanged for the whole wave: no rescale pass
      l_run += psum;
    } else {
      float alpha = PCM_EXP2F(m_run - m_new);
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < C::DV; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc_o[i][r] *= alpha;
    }
    bf16x8 pf[4];
#pragma unroll
    for (int ss = 0; ss < 4; ss++) pf[ss] = pack_frag(s_[ss >> 1], ss & 1);
#pragma unroll
    for (int i = 0; i < C::DV; i++)
#pragma unroll
      for (int ss = 0; ss < 4; ss++) {
        bf16x8 vf = *(const bf16x8*)(Vt + tr_off(32 * i + l31, 2 * ss + hi));
        acc_o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ss], acc_o[i], 0, 0, 0);
      }
  }
  float l_tot = l_run + __shfl_xor(l_run, 32);
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
template <int D, bool PK>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* dO, const bf16_t* kt,
                                                          const float* lse, const float* delta, bf16_t* dq, int H, int Lq,
                                                          int Lk, int ldq, int ldk, int ldo, float scale) {
  using C = AttnCfg<D>;
  __shared__ __attribute__((aligned(16))) char Ks[64 * C::RKU * 16];
  __shared__ __attribute__((aligned(16))) char Vs[64 * C::RKU * 16];
  __shared__ __attribute__((aligned(16))) char Kt[C::DV * 32 * 128];
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
  const float dl = qrow < Lq ? delta[((size_t)b * H + h) * Lq + qrow] : 0.f;
  f32x16 acc[C::DV];
#pragma unroll
  for (int i = 0; i < C::DV; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  const float sc = scale * LOG2E;
  zero_pad_chunks<D, 64>(Ks, tid);
  zero_pad_chunks<D, 64>(Vs, tid);
  RowStage<D, 64> kst, vst;
  TStage<D, PK> ktst;
  const bf16_t* ktb = PK ? kt + (size_t)(b * H + h) * ((Lk + 63) >> 6) * D * 64 : nullptr;
  if (AttnPrefetch<D>::value) { kst.load(kb, ldk, 0, Lk, tid); vst.load(vb, ldk, 0, Lk, tid); ktst.load(kb, ldk, ktb, 0, Lk, tid); }
  for (int kv0 = 0; kv0 < Lk; kv0 += 64) {
    __syncthreads();
    if (AttnPrefetch<D>::value) {
      kst.store(Ks, Lk, tid); vst.store(Vs, Lk, tid); ktst.store(Kt, Lk, tid);
    } else {
      load_rowmajor<D, 64>(Ks, kb, ldk, kv0, Lk, tid);
      load_rowmajor<D, 64>(Vs, vb, ldk, kv0, Lk, tid);
      load_transposed64<D>(Kt, kb, ldk, kv0, Lk, tid);
    }
    __syncthreads();
    if (AttnPrefetch<D>::value && kv0 + 64 < Lk) {
      kst.load(kb, ldk, kv0 + 64, Lk, tid); vst.load(vb, ldk, kv0 + 64, Lk, tid); ktst.load(kb, ldk, ktb, kv0 + 64, Lk, tid);
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
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float p = PCM_EXP2F(fmaf(s_[t][r], sc, -L2));
        s_[t][r] = p * (dp[t][r] - dl) * scale;  // dS^T
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
#pragma unroll
    for (int i = 0; i < C::DV; i++)
#pragma unroll
      for (int ss = 0; ss < 4; ss++) {
        bf16x8 ktf = *(const bf16x8*)(Kt + tr_off(32 * i + l31, 2 * ss + hi));
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, df[ss], acc[i], 0, 0, 0);
      }
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
template <int D, bool PK>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, const bf16_t* dO, const bf16_t* qt,
                                                            const bf16_t* ot, const float* lse, const float* delta, bf16_t* dk, bf16_t* dv, int H,
                                                            int Lq, int Lk, int ldq, int ldk, int ldo, float scale) {
  using C = AttnCfg<D>;
  __shared__ __attribute__((aligned(16))) char Qs[64 * C::RKU * 16];
  __shared__ __attribute__((aligned(16))) char Os[64 * C::RKU * 16];
  __shared__ __attribute__((aligned(16))) char Qt[C::DV * 32 * 128];
  __shared__ __attribute__((aligned(16))) char Ot[C::DV * 32 * 128];
  __shared__ float L2s[64], dls[64];
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
  zero_pad_chunks<D, 64>(Qs, tid);
  zero_pad_chunks<D, 64>(Os, tid);
  RowStage<D, 64> qst, ost;
  TStage<D, PK> qtst, otst;
  const bf16_t* qtb = PK ? qt + (size_t)(b * H + h) * ((Lq + 63) >> 6) * D * 64 : nullptr;
  const bf16_t* otb = PK ? ot + (size_t)(b * H + h) * ((Lq + 63) >> 6) * D * 64 : nullptr;
  float l2r = 0.f, dlr = 0.f;
  auto stage_load = [&](int q0_) {
    qst.load(qb, ldq, q0_, Lq, tid); ost.load(dob, ldo, q0_, Lq, tid);
    qtst.load(qb, ldq, qtb, q0_, Lq, tid); otst.load(dob, ldo, otb, q0_, Lq, tid);
    {
      int qr = q0_ + (tid & 63);
      if (qr >= Lq) qr = Lq - 1;
      l2r = lse[((size_t)b * H + h) * Lq + qr];
      dlr = delta[((size_t)b * H + h) * Lq + qr];
    }
  };
  if (AttnPrefetch<D>::value) stage_load(0);
  for (int qq0 = 0; qq0 < Lq; qq0 += 64) {
    __syncthreads();
    if (AttnPrefetch<D>::value) {
      qst.store(Qs, Lq, tid); ost.store(Os, Lq, tid); qtst.store(Qt, Lq, tid); otst.store(Ot, Lq, tid);
      if (tid < 64) { L2s[tid] = l2r; dls[tid] = dlr; }
    } else {
      load_rowmajor<D, 64>(Qs, qb, ldq, qq0, Lq, tid);
      load_rowmajor<D, 64>(Os, dob, ldo, qq0, Lq, tid);
      load_transposed64<D>(Qt, qb, ldq, qq0, Lq, tid);
      load_transposed64<D>(Ot, dob, ldo, qq0, Lq, tid);
      if (tid < 64) {
        int qr = qq0 + tid;
        L2s[tid] = qr < Lq ? lse[((size_t)b * H + h) * Lq + qr] : 0.f;
        dls[tid] = qr < Lq ? delta[((size_t)b * H + h) * Lq + qr] : 0.f;
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
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        int ql = 32 * t + 4 * hi + (r & 3) + 8 * (r >> 2);
        float p = PCM_EXP2F(fmaf(s_[t][r], sc, -L2s[ql]));
        dp[t][r] = p * (dp[t][r] - dls[ql]) * scale;  // dS[q][kv]
        s_[t][r] = p;
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
#pragma unroll
    for (int i = 0; i < C::DV; i++)
#pragma unroll
      for (int ss = 0; ss < 4; ss++) {
        bf16x8 otf = *(const bf16x8*)(Ot + tr_off(32 * i + l31, 2 * ss + hi));
        bf16x8 qtf = *(const bf16x8*)(Qt + tr_off(32 * i + l31, 2 * ss + hi));
        acc_v[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(otf, pf[ss], acc_v[i], 0, 0, 0);  // dV^T[d][kv]
        acc_k[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, df[ss], acc_k[i], 0, 0, 0);  // dK^T[d][kv]
      }
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

// Workspace for the packed transposed operands (pcm_attn_pack_t images): the forward needs V^T, the backward K^T, Q^T and dO^T.
// 0 when the packed path is not used (short sequences: the one-off packing pass does not pay; head dims without register staging).
static int g_attn_pack_min = 1024;
extern "C" void pcm_debug_attn_pack_min_len(int n) { g_attn_pack_min = n; }   // tests only: exercise the packed path on short sequences
static bool attn_use_packed(int Lq, int Lk, int d) { return d <= 80 && Lq >= g_attn_pack_min && Lk >= g_attn_pack_min; }
static size_t attn_packed_bytes(int B, int H, int L, int d) { return (size_t)B * H * ((L + 63) / 64) * d * 128; }
extern "C" size_t pcm_attn_workspace_bytes(int B, int H, int Lq, int Lk, int d, int backward) {
  if (!attn_use_packed(Lq, Lk, d)) return 0;
  return backward ? attn_packed_bytes(B, H, Lk, d) + 2 * attn_packed_bytes(B, H, Lq, d) : attn_packed_bytes(B, H, Lk, d);
}
// heads per pack block: the LDS image of a block is 64 rows x (HG*d + 8) bf16; HG*d <= 504 keeps it at <= 64 KB (no opt-in attribute,
// two blocks per CU) for every head count -- SDXL level 2 (20 x 64) and SD3 (24 x 64) run as 3 / 4 head groups
static int attn_pack_heads_per_block(int H, int d) { int hg = 504 / d; return hg < 1 ? 1 : (hg > H ? H : hg); }
static int attn_pack_launch(const void* x, void* xt, int B, int H, int L, int d, int ld, void* stream) {
  const int HG = attn_pack_heads_per_block(H, d);
  const size_t smem = 64 * (size_t)(HG * d + 8) * 2;
  PCM_CHECK(smem <= 64 * 1024 && (d % 8) == 0, PCM_EUNSUPPORTED, "pcm_attn: packed operand path: head_dim %d needs %zu B of LDS per tile", d, smem);
  PCM_CHECK(B <= 65535 && (H + HG - 1) / HG <= 65535, PCM_EUNSUPPORTED, "pcm_attn: batch %d exceeds the grid limit", B);
  PCM_LAUNCH(attn_pack_t_kernel, dim3((L + 63) / 64, B, (H + HG - 1) / HG), dim3(256), smem, stream, (const bf16_t*)x, (bf16_t*)xt, H, L, d, ld, HG);
  return PCM_OK;
}

extern "C" int pcm_attn_fwd_ws(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk,
                               int d, int ldq, int ldk, int ldo, float scale, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = attn_check("pcm_attn_fwd", q, k, v, B, H, Lq, Lk, d, ldq, ldk, ldo)) return rc;
  PCM_CHECK(o && PCM_ALIGNED16(o), PCM_EALIGN, "pcm_attn_fwd: o");
  dim3 grid((Lq + 127) / 128, H, B), block(256);
  const bool pk = workspace && attn_use_packed(Lq, Lk, d);
  if (pk) {
    PCM_CHECK(PCM_ALIGNED16(workspace) && workspace_bytes >= pcm_attn_workspace_bytes(B, H, Lq, Lk, d, 0), PCM_EINVAL, "pcm_attn_fwd_ws: workspace");
    if (int rc = attn_pack_launch(v, workspace, B, H, Lk, d, ldk, stream)) return rc;
  }
#define FWD_CALL(DD)                                                                                                               \
  if (pk && AttnPrefetch<DD>::value)                                                                                               \
    PCM_LAUNCH((attn_fwd_kernel<DD, true>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,          \
               (const bf16_t*)workspace, (bf16_t*)o, lse, H, Lq, Lk, ldq, ldk, ldo, scale);                                         \
  else                                                                                                                             \
    PCM_LAUNCH((attn_fwd_kernel<DD, false>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,         \
               (const bf16_t*)nullptr, (bf16_t*)o, lse, H, Lq, Lk, ldq, ldk, ldo, scale)
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
  if (int rc = attn_check("pcm_attn_bwd", q, k, v, B, H, Lq, Lk, d, ldq, ldk, ldo)) return rc;
  PCM_CHECK(o && dO && lse && delta && PCM_ALIGNED16(o) && PCM_ALIGNED16(dO), PCM_EALIGN, "pcm_attn_bwd: o/dO/lse/delta");
  PCM_LAUNCH(attn_delta_kernel, dim3((Lq * H + 255) / 256, B), dim3(256), 0, stream, (const bf16_t*)o, (const bf16_t*)dO, delta, H, Lq, d, ldo);
  const bool pk = workspace && attn_use_packed(Lq, Lk, d);
  const bf16_t *kt = nullptr, *qt = nullptr, *ot = nullptr;
  if (pk) {
    PCM_CHECK(PCM_ALIGNED16(workspace) && workspace_bytes >= pcm_attn_workspace_bytes(B, H, Lq, Lk, d, 1), PCM_EINVAL, "pcm_attn_bwd_ws: workspace");
    char* w = (char*)workspace;
    kt = (const bf16_t*)w; qt = (const bf16_t*)(w + attn_packed_bytes(B, H, Lk, d)); ot = (const bf16_t*)(w + attn_packed_bytes(B, H, Lk, d) + attn_packed_bytes(B, H, Lq, d));
    if (dq) { if (int rc = attn_pack_launch(k, (void*)kt, B, H, Lk, d, ldk, stream)) return rc; }
    if (dk && dv) {
      if (int rc = attn_pack_launch(q, (void*)qt, B, H, Lq, d, ldq, stream)) return rc;
      if (int rc = attn_pack_launch(dO, (void*)ot, B, H, Lq, d, ldo, stream)) return rc;
    }
  }
  if (dq) {
    dim3 grid((Lq + 127) / 128, H, B), block(256);
#define DQ_CALL(DD)                                                                                                                  \
  if (pk && AttnPrefetch<DD>::value)                                                                                                 \
    PCM_LAUNCH((attn_bwd_dq_kernel<DD, true>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,         \
               (const bf16_t*)dO, kt, lse, delta, (bf16_t*)dq, H, Lq, Lk, ldq, ldk, ldo, scale);                                      \
  else                                                                                                                               \
    PCM_LAUNCH((attn_bwd_dq_kernel<DD, false>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,        \
               (const bf16_t*)dO, (const bf16_t*)nullptr, lse, delta, (bf16_t*)dq, H, Lq, Lk, ldq, ldk, ldo, scale)
    ATTN_DISPATCH(d, DQ_CALL)
  }
  if (dk && dv) {
    dim3 grid((Lk + 127) / 128, H, B), block(256);
#define DKV_CALL(DD)                                                                                                                 \
  if (pk && AttnPrefetch<DD>::value)                                                                                                 \
    PCM_LAUNCH((attn_bwd_dkdv_kernel<DD, true>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,       \
               (const bf16_t*)dO, qt, ot, lse, delta, (bf16_t*)dk, (bf16_t*)dv, H, Lq, Lk, ldq, ldk, ldo, scale);                     \
  else                                                                                                                               \
    PCM_LAUNCH((attn_bwd_dkdv_kernel<DD, false>), grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,      \
               (const bf16_t*)dO, (const bf16_t*)nullptr, (const bf16_t*)nullptr, lse, delta, (bf16_t*)dk, (bf16_t*)dv, H, Lq, Lk,    \
               ldq, ldk, ldo, scale)
    ATTN_DISPATCH(d, DKV_CALL)
  }
  return pcm_post_launch("pcm_attn_bwd");
}
extern "C" int pcm_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                            float* delta, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int d, int ldq, int ldk,
                            int ldo, float scale, void* stream) {
  return pcm_attn_bwd_ws(q, k, v, o, dO, lse, delta, dq, dk, dv, B, H, Lq, Lk, d, ldq, ldk, ldo, scale, nullptr, 0, stream);
}
