// Device-side pieces shared by the attention kernels (attention.hip: backward + the first forward; attention_fwd.hip: the software-
// pipelined forward): tile geometry, row-major LDS images, transpose-read fragments, register staging.
#pragma once
#include "pcm_common.h"


#define LOG2E 1.4426950408889634f
// cycle stamps of the forward kernel (tools/attn_timeline.py, -DPCM_ABLATE builds only): lane 0 of every wave of ONE mid-grid workgroup
// records s_memtime at 6 points of each of its first 32 key tiles
#if defined(PCM_ABLATE) && defined(ATTN_STAMPS_OWNER)
__device__ unsigned long long g_attn_stamps[4][32][8];
extern "C" int pcm_debug_attn_stamps(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_stamps), sizeof(g_attn_stamps)); }
#define ATTN_STAMP(k)                                                                                                        \
  do {                                                                                                                       \
    if (stamp_on && (kv0 >> 6) < 32 && lane == 0) g_attn_stamps[wave][kv0 >> 6][k] = __builtin_readcyclecounter();          \
  } while (0)
#else
#define ATTN_STAMP(k) do { } while (0)
#endif
#define ATTN_DBG_PARAM , int xcd_remap
#define ATTN_DBG_ARG , g_attn_xcd_remap
#define ATTN_ABL(bit) 0

template <int D>
struct AttnCfg {
  static constexpr int DK16 = (D + 15) / 16;     // QK^T K-steps of 16
  static constexpr int KCH = 2 * DK16;           // 16-B chunks per row-major row (incl. zero pad)
  static constexpr int RKU = KCH | 1;            // row stride in 16-B units (odd -> conflict-free ds_read_b128)
  static constexpr int DV = (D + 31) / 32;       // 32-row tiles of the transposed output
  static constexpr int DG = D / 8;               // 8-wide column groups
};

// row-major tile image: [rows][RKU*16 B]; chunk c of row r at (r*RKU + c)*16.  Both MFMA operand orientations come out of it:
//   * k along the columns (S^T = K Q^T): one ds_read_b128 per fragment;
//   * k along the ROWS (O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS): two ds_read_b64_tr_b16 per fragment
//     (tr_frag below) -- no transposed copy of the tile in LDS, no transposing stage, no packed-operand pre-pass.

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
// pad chunks (columns D .. 16*DK16-1 of every row; the spare chunk of the odd row stride too): zero, or with ``ones`` a 1.0 in column D
// (the tile then carries a ones COLUMN, i.e. a ones row of its transpose: see the forward kernel)
template <int D, int ROWS>
__device__ __forceinline__ void fill_pad_chunks(char* dst, int tid, bool ones) {
  using C = AttnCfg<D>;
  constexpr int NP = C::RKU - C::DG;
  for (int u = tid; u < ROWS * NP; u += 256) {
    int r = u / NP, c = C::DG + (u - r * NP);
    *(uint4*)(dst + (r * C::RKU + c) * 16) = make_uint4((ones && c == C::DG) ? (unsigned)PCM_ONE_BITS : 0u, 0u, 0u, 0u);
  }
}
// fragment with the contraction index along the tile ROWS: A[i = column 32*it + (lane&31)][k], k-slot (hi, e) of step ss = tile row
// 16ss + 8(e>>2) + 4hi + (e&3).  Per 16-lane group g = lane>>4 (columns 32it + 16(g&1) .., hi = g>>1) source lane 4j+q addresses the quad
// (row 16ss + 4hi + j, columns +4q..+3) and receives its own column's 4 rows (pcm_common.h PCM_DS_READ_TR16); rows +8 give e = 4..7.
template <int D>
struct TrFrag {
  using C = AttnCfg<D>;
  int base;
  __device__ __forceinline__ TrFrag(int lane) {
    const int s = lane & 15, j = s >> 2, q = s & 3, g = lane >> 4;
    base = (4 * (g >> 1) + j) * (C::RKU * 16) + (16 * (g & 1) + 4 * q) * 2;
  }
  __device__ __forceinline__ bf16x8 get(const char* tile, int it, int ss) const {
    const char* p = tile + base + 16 * ss * (C::RKU * 16) + 64 * it;
    return pcm_join4(PCM_DS_READ_TR16(p), PCM_DS_READ_TR16(p + 8 * (C::RKU * 16)));
  }
};
// The four k-steps of one 32-column group (one accumulator tile of the k-along-rows MFMAs), issued as 8 untracked transpose reads: they
// can be put in flight long before their MFMAs (forward: before the softmax) and cost the wave no wait until wait() / use.
template <int D>
struct TrQuad {
  using C = AttnCfg<D>;
  bf16x4 lo[4], hi[4];
  template <int IT>
  __device__ __forceinline__ void issue(const char* tile, const TrFrag<D>& f) {
    const char* p = tile + f.base;
#define TRQ_STEP(SS)                                                          \
  PCM_TR16_ISSUE(lo[SS], p, 16 * SS * (C::RKU * 16) + 64 * IT);               \
  PCM_TR16_ISSUE(hi[SS], p, 16 * SS * (C::RKU * 16) + 64 * IT + 8 * (C::RKU * 16));
    TRQ_STEP(0) TRQ_STEP(1) TRQ_STEP(2) TRQ_STEP(3)
#undef TRQ_STEP
  }
  __device__ __forceinline__ void wait() { PCM_TR16_WAIT8(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]); }
  __device__ __forceinline__ void keep() { PCM_TR16_KEEP8(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]); }
  __device__ __forceinline__ bf16x8 frag(int ss) const { return pcm_join4(lo[ss], hi[ss]); }
};
// ---- register staging (issue the NEXT tile's global loads before computing on the current tile;
// the LDS write happens after the next barrier, so L2/HBM latency hides under the MFMA phase) ----
// Loads are UNCONDITIONAL from clamped addresses (a predicated load + zero select makes hipcc wait
// vmcnt(0) right after issue: WAW on the destination); out-of-range rows are zeroed at store time.
// Per-thread geometry of a ROWS x D tile copy, computed ONCE per kernel: byte offset of each of the thread's 16-B pieces from the tile's
// first row in global memory and in the LDS image.  With it a full tile costs no VALU address math at all (round 2 measured ~40 VALU
// instructions per tile here -- two v_mad_i64, clamps, 16 validity selects -- and on gfx950 VALU work does NOT overlap the MFMAs of the
// same SIMD: tools/probes/mfmavalu.hip): the tile base is wave-uniform (scalar registers), the loads are base + 32-bit offset.
template <int D, int ROWS>
struct RowGeom {
  using C = AttnCfg<D>;
  static constexpr int N = (ROWS * C::DG + 255) / 256;
  unsigned goff[N], loff[N];
  __device__ __forceinline__ RowGeom(int ld, int tid) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      int u = tid + 256 * i;
      if (u >= ROWS * C::DG) u = ROWS * C::DG - 1;     // threads beyond the last piece re-load it (their store is predicated off)
      const int rr = u / C::DG, c = u - rr * C::DG;
      goff[i] = (unsigned)(rr * ld + 8 * c) * 2u;
      loff[i] = (unsigned)((rr * C::RKU + c) * 16);
    }
  }
};
template <int D, int ROWS>
struct RowStage {
  using C = AttnCfg<D>;
  static constexpr int N = RowGeom<D, ROWS>::N;
  uint4 r[N];
  int row0_;
  // full tiles (wave-uniform test, a real branch): unconditional loads from uniform base + per-thread offset.  Only the last tile of a
  // ragged sequence takes the clamped path (loads stay unconditional there too: a predicated load + zero select makes hipcc wait
  // vmcnt(0) right after issue; out-of-range rows are zeroed at store time).
  __device__ __forceinline__ void load(const RowGeom<D, ROWS>& gm, const bf16_t* src, int ld, int row0, int nrows_valid, int tid) {
    row0_ = row0;
    // raw buffer loads: resource (scalar) on the operand's base, the tile's first row in the scalar offset, the per-thread piece in a
    // 32-bit VGPR offset -- no 64-bit per-thread pointers to keep or to advance (operand spans stay far below 2 GB per (batch, head))
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x80000000u, 0x00020000);
    if (row0 + ROWS <= nrows_valid) {
      asm volatile("" ::: "memory");
      const int soff = row0 * ld * 2;
#pragma unroll
      for (int i = 0; i < N; i++) r[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, gm.goff[i], soff, 0));
      return;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      int u = tid + 256 * i;
      if (u >= ROWS * C::DG) u = ROWS * C::DG - 1;
      int rr = u / C::DG, c = u - rr * C::DG;
      int row = row0 + rr;
      if (row >= nrows_valid) row = nrows_valid - 1;
      r[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(row * ld + 8 * c) * 2u, 0, 0));
    }
  }
  // branch-free forms for loops whose tiles are known to be full (the software-pipelined forward's steady state: one basic block)
  __device__ __forceinline__ void load_full(const RowGeom<D, ROWS>& gm, const bf16_t* src, int ld, int row0) {
    row0_ = row0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x80000000u, 0x00020000);
    const int soff = row0 * ld * 2;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, gm.goff[i], soff, 0));
  }
  // (threads beyond the last piece hold a second copy of it -- RowGeom clamps their piece index -- and store the same bytes to the same
  // place: no predicate, no exec-mask branch)
  __device__ __forceinline__ void store_full(const RowGeom<D, ROWS>& gm, char* dst, int) const {
#pragma unroll
    for (int i = 0; i < N; i++) *(uint4*)(dst + gm.loff[i]) = r[i];
  }
  __device__ __forceinline__ void store(const RowGeom<D, ROWS>& gm, char* dst, int nrows_valid, int tid) const {
    if (row0_ + ROWS <= nrows_valid) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < N; i++)
        if (tid + 256 * i < ROWS * C::DG) *(uint4*)(dst + gm.loff[i]) = r[i];
      return;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      int u = tid + 256 * i;
      int rr = u / C::DG;
      if (u < ROWS * C::DG) {
        uint4 v = r[i];
        if (row0_ + rr >= nrows_valid) v = make_uint4(0u, 0u, 0u, 0u);
        *(uint4*)(dst + gm.loff[i]) = v;
      }
    }
  }
};
// ---- LDS-DMA staging (csrc/attention_ps.hip, round 5): a ROWS x D row-major tile goes from global memory straight into the padded LDS
// image (buffer_load ... lds, 16 B per lane, no VGPR round trip, no ds_write).  Wave-instruction j of a tile fills LDS pieces 64j .. 64j+63
// (16 B each, contiguous: the DMA writes LDS linearly); piece p = row p / RKU, chunk p % RKU of the image.  Lanes that fall on the pad chunks
// of a row are MASKED OFF, so the pad keeps what fill_pad_chunks put there once (zeros, or the ones columns of the slot / row-sum
// tricks); rows beyond the tensor are fetched out of range, i.e. zero-filled by the buffer hardware.  The four waves split the
// instructions of a tile round-robin.  All per-lane geometry is computed once per kernel.
template <int D, int ROWS>
struct DmaTile {
  using C = AttnCfg<D>;
  static constexpr int NI = (ROWS * C::RKU + 63) / 64;     // wave-instructions per tile (7 at d = 40)
  static constexpr int PW = (NI + 3) / 4;                  // ... per wave
  static_assert((ROWS * C::RKU) % 64 == 0, "tile image must be whole 1-KiB DMA instructions");
  unsigned voff[PW];
  int row[PW];
  bool act[PW];
  __device__ __forceinline__ DmaTile(int ld, int lane, int wave) {
#pragma unroll
    for (int i = 0; i < PW; i++) {
      const int j = wave + 4 * i, p = 64 * j + lane, r = p / C::RKU, c = p - r * C::RKU;
      act[i] = j < NI && c < C::DG;
      row[i] = r;
      voff[i] = (unsigned)(r * ld + 8 * c) * 2u;
    }
  }
  // tile rows row0 .. row0 + ROWS - 1 of ``src`` (row stride ld elements) -> image at ``dst`` (wave-uniform LDS address)
  __device__ __forceinline__ void issue(const bf16_t* src, int ld, int row0, int nrows_valid, char* dst, int wave) const {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x80000000u, 0x00020000);
    const unsigned soff = (unsigned)(row0 * ld) * 2u;
    const int left = nrows_valid - row0;           // rows of this tile inside the tensor
    if (left >= ROWS) {                            // full tile (wave-uniform test, kept a real branch): no per-lane validity select
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < PW; i++) {
        const int j = wave + 4 * i;
        if (j < NI) PCM_DMA16_MASKED(rs, dst + 1024 * j, voff[i], soff, act[i]);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < PW; i++) {
      const int j = wave + 4 * i;
      if (j < NI) PCM_DMA16_MASKED(rs, dst + 1024 * j, row[i] < left ? voff[i] : 0x80000000u, soff, act[i]);
    }
  }
};
template <int D> struct AttnPrefetch { static constexpr bool value = D <= 80; };
// LDS bytes of one row-major tile; the k-along-rows reads of the last 32-column group run up to 32*DV columns wide, i.e. past the end
// of short rows into the next row (finite data feeding accumulator rows >= D that are never stored) -- 64 B of slack behind the last row
template <int D> struct TileBytes { static constexpr int value = 64 * AttnCfg<D>::RKU * 16 + 64; };

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

