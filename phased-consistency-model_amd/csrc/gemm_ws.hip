// pcm_gemm_ws_kernel -- WEIGHTS-STATIONARY member of the pcm_gemm_bf16 family, for the short-K projections of the 64x64 level
// (Attention.to_q / to_out.0, Transformer2DModel.proj_in / proj_out, the fused q|k|v projection: N = 320 or 960, K = 320 (+ 64 / 192 LoRA)
// at M = 131072 / 65536: train_pcm_lora_sd15.py:866-885, discriminator_sd15.py:264-342).
//
// Why (round 6): a 256 x 320 phased tile at K = 320 streams 164 KB of activations AND 205 KB of weights through LDS-DMA
// for 5 K-tiles of MFMA work -- its K loop takes 20 us against 5.3 us of MFMA time, bound by the ~26 GB/s a CU can ingest, and the weight
// operand (identical for every M tile, re-fetched from L2 by each of them) is 55 % of those bytes.  Here the weights do not move: a
// workgroup of 4 waves (one per SIMD, up to 512 VGPRs each) holds a 320-column slice of W -- every wave the MFMA fragments of its 80
// columns over ALL of K, 200-240 registers -- and walks its share of the rows: 64-row activation tiles [64][K] through a two-stage
// LDS-DMA ring (buffer resources: x and, in a second sub-tile, the LoRA t), 4 x 5 accumulators of 16 x 16 per wave, epilogue from a
// wave-private LDS staging area with whole 16-byte row pieces on the global side.  Per 64 rows: 200-320 MFMAs per wave (~1.4-2.3 us), 40-64 KB of
// DMA, 40 KB of stores; the weights cost one 205-330 KB register fill per workgroup (L2 hits), amortised over 4-8 row tiles.
// Same contract and the same fp32 operation order as the phased tile's epilogue (alpha * acc, + bias, + residual, ONE rounding): bit-identical
// results (tests/kernel_cases.py::case_gemm_ws).
// RESULT (MI355X, profiles/r06_l_gemm_ws_ab.txt): SLOWER than the phased tile on every N = 320 shape (x1.06-1.35), x0.88-1.09 at N = 960.  A 64-row
// step takes 5.1 us against 1.4 us of MFMAs: with one wave per SIMD nothing overlaps inside a SIMD -- 20-30 LDS-DMA issues (~150 cycles each), the
// LDS read waits of 10-12 k-steps and the ~700 instructions of the wave-private epilogue all add to the MFMA issue time (the phased tile hides
// them behind its second wave per SIMD).  What would fix it -- the weight slice split over K between two waves of a SIMD, partial tiles meeting
// in LDS -- was sized (section 9 of DESIGN.md) and not built.  TOOLS build only; the product planner never selects this kernel.
// SECOND MEASUREMENT (profiles/r06_q_gemm_ws_forms_ab.txt): the ISA of the first one showed hipcc's s_waitcnt vmcnt(0) in front of every staging
// ds_read / ds_write of the epilogue (it cannot tell which LDS bytes pending LDS-DMA will write) -- the tile prefetched two steps ahead was drained
// every step.  With untracked LDS accesses (PCM_LDS_*, inline asm) the steady state holds only the counted waits, and an EIGHT-wave form
// (pcm_gemm_ws8_kernel below: two waves per SIMD, 40 columns each) removes the one-wave-per-SIMD serialisation as well -- and both forms still only
// MATCH the phased tile at M = 131072 (46.9 us phased, 49.0 four waves, 51.4 eight waves; with bias + residual 55.9 / 59.5 / 50.7) and lose at
// M <= 65536 (the weight fill is amortised over 4 steps).  Because these launches are not issue-bound at all: 168-252 MB of activations in + out in
// 47-55 us is 3.4-5.0 TB/s -- the fabric's rate for a read + write mix.  The phased tile already sits there; no kernel structure moves it.
#include "gemm_dev.h"

#define PCM_RSRC_FLAGS 0x00020000

template <int NK>      // 32-wide k-steps over both segments: K0 + K1 = 32 * NK  (10: K = 320; 12: + rank-64 LoRA)
__global__ __launch_bounds__(256, 1) void pcm_gemm_ws_kernel(GemmDev g, int steps_per_block) {
#if PCM_KERNEL_BODY
  // One ring stage = two sub-tiles, [64 rows][640 B] of segment 0 and (NK == 12) [64 rows][128 B] of segment 1: every LDS-DMA instruction
  // (64 lanes x 16 B) then lies inside ONE segment, i.e. reads through one buffer resource.
  constexpr int NK0 = 10, NK1 = NK - NK0;
  constexpr int T0 = 64 * 640, T1 = 64 * 64 * NK1, TILE = T0 + T1;
  constexpr int NP0 = 10, NP1 = T1 / 4096, NP = NP0 + NP1;      // LDS-DMA pieces per thread and tile: 10 (+ 2)
  constexpr int STG = 64 * 160;                                  // wave-private staging: 64 rows x 80 columns x 2 bytes
  PCM_DYN_SMEM(smem);                                            // [2][TILE] ring | [4][STG] staging
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  const int K0 = g.seg[0].K, K1 = NK1 ? g.seg[1].K : 0;
  const int n0 = blockIdx.y * 320 + 80 * wave;
  const int steps = (g.M + 63) >> 6;
  const int s_begin = blockIdx.x * steps_per_block;
  int s_end = s_begin + steps_per_block; if (s_end > steps) s_end = steps;
  if (s_begin >= s_end) return;
  char* stage = smem + 2 * TILE + wave * STG;

  // ---- loader geometry.  Piece j of this thread is 16-byte slot (tid + 256 j) of its sub-tile in LDS order; slot c of row r holds the LOGICAL
  // chunk (c & ~7) | ((c & 7) ^ ((r >> 1) & 7)) of the row (swizzle on the source side: the LDS side of the DMA is lane-linear), so that the
  // 16-row fragment reads below hit 16 different 16-byte slots.  Per-lane byte offsets of the FIRST tile; later tiles add a scalar offset;
  // rows beyond M lie beyond num_records and are zero-filled by the hardware.
  const unsigned rec0 = (unsigned)g.M * (unsigned)(g.seg[0].lda * 2);
  unsigned voff0[NP0], voff1[NP1 ? NP1 : 1], voffr[10];
#pragma unroll
  for (int j = 0; j < NP0; j++) {
    const int u = tid + 256 * j, r = u / 40, c = u - r * 40;
    const int lc = (c & ~7) | ((c & 7) ^ ((r >> 1) & 7));
    voff0[j] = (unsigned)(64 * s_begin + r) * (unsigned)(g.seg[0].lda * 2) + 16 * lc;
  }
#pragma unroll
  for (int j = 0; j < NP1; j++) {
    const int u = tid + 256 * j, r = u >> 3, c = u & 7;
    voff1[j] = (unsigned)(64 * s_begin + r) * (unsigned)(g.seg[1].lda * 2) + 16 * (c ^ ((r >> 1) & 7));
  }
#pragma unroll
  for (int j = 0; j < 10; j++) {
    const int p = lane + 64 * j, r = p / 10, c = p - r * 10;
    voffr[j] = (unsigned)(64 * s_begin + r) * (unsigned)(g.ldr * 2) + 2 * (n0 + 8 * c);
  }
  auto issue_tile = [&](int s, int buf) {
    __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.seg[0].a, 0, rec0, PCM_RSRC_FLAGS);
    char* dst = smem + buf * TILE + (64 * wave) * 16;
    const unsigned soff0 = (unsigned)(s - s_begin) * (unsigned)(64 * g.seg[0].lda * 2);
#pragma unroll
    for (int j = 0; j < NP0; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, PCM_AS3(dst + 4096 * j), 16, voff0[j], soff0, 0, 0);
    if constexpr (NP1 > 0) {
      __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)g.seg[1].a, 0, (unsigned)g.M * (unsigned)(g.seg[1].lda * 2), PCM_RSRC_FLAGS);
      const unsigned soff1 = (unsigned)(s - s_begin) * (unsigned)(64 * g.seg[1].lda * 2);
#pragma unroll
      for (int j = 0; j < NP1; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, PCM_AS3(dst + T0 + 4096 * j), 16, voff1[j], soff1, 0, 0);
    }
  };
  issue_tile(s_begin, 0);

  // ---- the weights: this wave's 80 columns x all of K as MFMA fragments, in registers for the whole launch
  bf16x8 bw[5][NK];
#pragma unroll
  for (int f = 0; f < 5; f++) {
    const int n = n0 + 16 * f + frow;
#pragma unroll
    for (int ks = 0; ks < NK; ks++) {
      const int k = 32 * ks + 8 * fk;
      bw[f][ks] = ks < NK0 ? *(const bf16x8*)(g.seg[0].w + (size_t)n * K0 + k) : *(const bf16x8*)(g.seg[1].w + (size_t)n * K1 + (k - 320));
    }
  }
  f32x4 bq[5];
#pragma unroll
  for (int f = 0; f < 5; f++) {
    const float4 b4 = g.bias ? *(const float4*)(g.bias + n0 + 16 * f + 4 * fk) : make_float4(0.f, 0.f, 0.f, 0.f);
    bq[f] = f32x4{b4.x, b4.y, b4.z, b4.w};
  }
  PCM_WAIT_VMCNT(0);       // (the register fill is complete before the ring is counted: from here on only LDS-DMA pieces are in flight)
  if (s_begin + 1 < s_end) issue_tile(s_begin + 1, 1);

  // fragment reads: lane (frow, fk) reads row 16 i + frow, logical chunk 4 ks + fk; (row >> 1) & 7 depends on frow only (16 i is a multiple of 16)
  const int swz = (frow >> 1) & 7;
  const bool has_res = g.res != nullptr;
  for (int s = s_begin; s < s_end; s++) {
    const int buf = (s - s_begin) & 1;
    const char* at = smem + buf * TILE;
    // tile s has landed: the first step waits for it here (the tile issued after it stays in flight), every later step confirmed it in the
    // previous step's epilogue wait; behind the barrier every wave's pieces have
    if (s == s_begin) {
      if (s + 1 < s_end) { if constexpr (NP == 10) PCM_WAIT_VMCNT(10); else PCM_WAIT_VMCNT(12); }
      else PCM_WAIT_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();
    // the residual pieces of this step: by LDS-DMA straight into this wave's staging area (piece p at byte 16 p: lane-linear), no registers;
    // they land under the MFMAs (the staging area is free: the previous step's read-out precedes this in the wave's program order)
    if (has_res) {
      __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)g.res, 0, (unsigned)g.M * (unsigned)(g.ldr * 2), PCM_RSRC_FLAGS);
      const unsigned soffr = (unsigned)(s - s_begin) * (unsigned)(64 * g.ldr * 2);
#pragma unroll
      for (int j = 0; j < 10; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, PCM_AS3(stage + 1024 * j), 16, voffr[j], soffr, 0, 0);
    }
    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int f = 0; f < 5; f++) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NK; ks++) {
      bf16x8 af[4];
      if (ks < NK0) {
        const int lc = 4 * ks + fk, pc = (lc & ~7) | ((lc & 7) ^ swz);
#pragma unroll
        for (int i = 0; i < 4; i++) af[i] = *(const bf16x8*)(at + (16 * i + frow) * 640 + pc * 16);
      } else {
        const int pc = (4 * (ks - NK0) + fk) ^ swz;
#pragma unroll
        for (int i = 0; i < 4; i++) af[i] = *(const bf16x8*)(at + T0 + (16 * i + frow) * (64 * NK1) + pc * 16);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int f = 0; f < 5; f++) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[f][ks], af[i], acc[i][f], 0, 0, 0);
    }
    // every wave is done reading ring stage `buf`: the tile two steps ahead goes there NOW, in front of this step's stores -- vmcnt counts stores
    // too, and a wait that has this step's stores behind it would expose their write acknowledgements every step (measured: 6.3 us per
    // 64-row step instead of ~2).  ONE wait per step: everything but the tile just issued -- the residual pieces of this step, the tile of the
    // next step and the previous step's stores (a whole step old) -- is complete; loads return in order, so a count <= NP with the newest NP
    // operations being that tile's loads cannot leave an older load pending.
    __builtin_amdgcn_s_barrier();
    if (s + 2 < s_end) {
      issue_tile(s + 2, buf);
      if constexpr (NP == 10) PCM_WAIT_VMCNT(10); else PCM_WAIT_VMCNT(12);
    } else {
      PCM_WAIT_VMCNT(0);
    }
    // ---- epilogue of the step, inside the wave (staging is wave-private: no workgroup barrier).  The lane that owns an accumulator element
    // finds its residual in the staging area; the sum is rounded once, in place; whole 16-byte row pieces go out.
    PCM_WAVE_LDS_FENCE();
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int f = 0; f < 5; f++) {
        char* slot = stage + (16 * i + frow) * 160 + 32 * f + 8 * fk;
        f32x4 a = acc[i][f] * g.alpha + bq[f];
        // (untracked LDS accesses, PCM_LDS_*: with the compiler's own ds_read / ds_write here hipcc drained the tile just issued -- s_waitcnt
        // vmcnt(0) in front of each -- every step; found after the first measurement of this form)
        if (has_res) {
          u32x2 rr2;
          PCM_LDS_LD64(rr2, slot);
          a[0] += bf2f((bf16_t)(rr2.x & 0xffff)); a[1] += bf2f((bf16_t)(rr2.x >> 16));
          a[2] += bf2f((bf16_t)(rr2.y & 0xffff)); a[3] += bf2f((bf16_t)(rr2.y >> 16));
        }
        u32x2 pk = u32x2{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
        PCM_LDS_ST64(slot, pk);
      }
    PCM_LDS_WAIT_ALL();
    PCM_WAVE_LDS_FENCE();
#pragma unroll
    for (int j = 0; j < 10; j++) {
      const int p = lane + 64 * j, r = p / 10, c = p - r * 10;
      const int m = 64 * s + r;
      u32x4 q;
      PCM_LDS_LD128(q, stage + p * 16);
      if (m < g.M) *(u32x4*)((bf16_t*)g.out + (size_t)m * g.ldo + n0 + 8 * c) = q;
    }
  }
#endif
}

// ---- second form (after the measurement above): EIGHT waves, two per SIMD, the 320-column slice split over COLUMNS -- 40 per wave, as three
// 16-column MFMA tiles of which the third is half used (+20 % MFMA issue on a kernel that is not MFMA-bound) -- so that every wave still runs the
// WHOLE K in one accumulator chain (bit-identical sums, no cross-wave reduction) with 120-144 weight registers, and the loader / epilogue /
// LDS-wait time of one wave hides under the other wave's MFMAs.  Same ring, same counted waits, same epilogue arithmetic; per wave and 64-row
// step: 120-144 MFMAs, 5-6 tile pieces + 5 residual pieces of LDS-DMA, 10 epilogue slots, 5 stores.
template <int NK>
__global__ __launch_bounds__(512, 1) void pcm_gemm_ws8_kernel(GemmDev g, int steps_per_block) {
#if PCM_KERNEL_BODY
  constexpr int NK0 = 10, NK1 = NK - NK0;
  constexpr int T0 = 64 * 640, T1 = 64 * 64 * NK1, TILE = T0 + T1;
  constexpr int NP0 = 5, NP1 = T1 / 8192, NP = NP0 + NP1;        // LDS-DMA pieces per thread and tile: 5 (+ 1)
  constexpr int STG = 64 * 80;                                   // wave-private staging: 64 rows x 40 columns x 2 bytes
  PCM_DYN_SMEM(smem);                                            // [2][TILE] ring | [8][STG] staging | [8][48] fp32 bias (per wave)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  const int K0 = g.seg[0].K, K1 = NK1 ? g.seg[1].K : 0;
  const int n0 = blockIdx.y * 320 + 40 * wave;
  const int steps = (g.M + 63) >> 6;
  const int s_begin = blockIdx.x * steps_per_block;
  int s_end = s_begin + steps_per_block; if (s_end > steps) s_end = steps;
  if (s_begin >= s_end) return;
  char* stage = smem + 2 * TILE + wave * STG;
  float* lbias = (float*)(smem + 2 * TILE + 8 * STG) + 48 * wave;      // this wave's 40 (+ 8 zero) bias values: read per step, no registers held

  // loader geometry: as the four-wave form, 512 threads (piece j of a thread = 16-byte slot tid + 512 j of the sub-tile in LDS order)
  const unsigned rec0 = (unsigned)g.M * (unsigned)(g.seg[0].lda * 2);
  unsigned voff0[NP0], voff1[NP1 ? NP1 : 1], voffr[5];
#pragma unroll
  for (int j = 0; j < NP0; j++) {
    const int u = tid + 512 * j, r = u / 40, c = u - r * 40;
    const int lc = (c & ~7) | ((c & 7) ^ ((r >> 1) & 7));
    voff0[j] = (unsigned)(64 * s_begin + r) * (unsigned)(g.seg[0].lda * 2) + 16 * lc;
  }
#pragma unroll
  for (int j = 0; j < NP1; j++) {
    const int u = tid + 512 * j, r = u >> 3, c = u & 7;
    voff1[j] = (unsigned)(64 * s_begin + r) * (unsigned)(g.seg[1].lda * 2) + 16 * (c ^ ((r >> 1) & 7));
  }
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const int p = lane + 64 * j, r = p / 5, c = p - r * 5;
    voffr[j] = (unsigned)(64 * s_begin + r) * (unsigned)(g.ldr * 2) + 2 * (n0 + 8 * c);
  }
  auto issue_tile = [&](int s, int buf) {
    __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.seg[0].a, 0, rec0, PCM_RSRC_FLAGS);
    char* dst = smem + buf * TILE + (64 * wave) * 16;
    const unsigned soff0 = (unsigned)(s - s_begin) * (unsigned)(64 * g.seg[0].lda * 2);
#pragma unroll
    for (int j = 0; j < NP0; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, PCM_AS3(dst + 8192 * j), 16, voff0[j], soff0, 0, 0);
    if constexpr (NP1 > 0) {
      __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)g.seg[1].a, 0, (unsigned)g.M * (unsigned)(g.seg[1].lda * 2), PCM_RSRC_FLAGS);
      const unsigned soff1 = (unsigned)(s - s_begin) * (unsigned)(64 * g.seg[1].lda * 2);
#pragma unroll
      for (int j = 0; j < NP1; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, PCM_AS3(dst + T0 + 8192 * j), 16, voff1[j], soff1, 0, 0);
    }
  };
  issue_tile(s_begin, 0);

  // the weights: this wave's 40 columns (three 16-column fragments; rows beyond the slice's 40 -- or beyond N -- are computed and dropped)
  bf16x8 bw[3][NK];
#pragma unroll
  for (int f = 0; f < 3; f++) {
    int n = n0 + 16 * f + frow; if (n >= g.N) n = g.N - 1;
#pragma unroll
    for (int ks = 0; ks < NK; ks++) {
      const int k = 32 * ks + 8 * fk;
      bw[f][ks] = ks < NK0 ? *(const bf16x8*)(g.seg[0].w + (size_t)n * K0 + k) : *(const bf16x8*)(g.seg[1].w + (size_t)n * K1 + (k - 320));
    }
  }
  // accumulator element (f, lane) holds output columns n0 + 16 f + 4 fk .. + 3 of row (16 i + frow): inside the slice iff 16 f + 4 fk < 40
  const bool own2 = fk < 2;
  if (lane < 48) lbias[lane] = (g.bias && lane < 40) ? g.bias[n0 + lane] : 0.f;
  PCM_WAIT_VMCNT(0);
  if (s_begin + 1 < s_end) issue_tile(s_begin + 1, 1);

  const int swz = (frow >> 1) & 7;
  const bool has_res = g.res != nullptr;
  for (int s = s_begin; s < s_end; s++) {
    const int buf = (s - s_begin) & 1;
    const char* at = smem + buf * TILE;
    if (s == s_begin) {
      if (s + 1 < s_end) { if constexpr (NP == 5) PCM_WAIT_VMCNT(5); else PCM_WAIT_VMCNT(6); }
      else PCM_WAIT_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();
    if (has_res) {
      __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)g.res, 0, (unsigned)g.M * (unsigned)(g.ldr * 2), PCM_RSRC_FLAGS);
      const unsigned soffr = (unsigned)(s - s_begin) * (unsigned)(64 * g.ldr * 2);
#pragma unroll
      for (int j = 0; j < 5; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, PCM_AS3(stage + 1024 * j), 16, voffr[j], soffr, 0, 0);
    }
    f32x4 acc[4][3];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int f = 0; f < 3; f++) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NK; ks++) {
      // (two row tiles at a time: 8 fragment registers live instead of 16 -- the K + 64 instantiation sits at the 256-register line)
#pragma unroll
      for (int ih = 0; ih < 2; ih++) {
        bf16x8 af[2];
        if (ks < NK0) {
          const int lc = 4 * ks + fk, pc = (lc & ~7) | ((lc & 7) ^ swz);
#pragma unroll
          for (int i = 0; i < 2; i++) af[i] = *(const bf16x8*)(at + (16 * (2 * ih + i) + frow) * 640 + pc * 16);
        } else {
          const int pc = (4 * (ks - NK0) + fk) ^ swz;
#pragma unroll
          for (int i = 0; i < 2; i++) af[i] = *(const bf16x8*)(at + T0 + (16 * (2 * ih + i) + frow) * (64 * NK1) + pc * 16);
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int f = 0; f < 3; f++) acc[2 * ih + i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[f][ks], af[i], acc[2 * ih + i][f], 0, 0, 0);
      }
    }
    // (the order of tile issue, the ONE counted wait and the stores: as in the four-wave form above)
    __builtin_amdgcn_s_barrier();
    if (s + 2 < s_end) {
      issue_tile(s + 2, buf);
      if constexpr (NP == 5) PCM_WAIT_VMCNT(5); else PCM_WAIT_VMCNT(6);
    } else {
      PCM_WAIT_VMCNT(0);
    }
    PCM_WAVE_LDS_FENCE();
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int f = 0; f < 3; f++) {
        if (f == 2 && !own2) continue;
        char* slot = stage + (16 * i + frow) * 80 + 32 * f + 8 * fk;
        u32x4 b4u;
        u32x2 rr2 = u32x2{0u, 0u};
        if (has_res) PCM_LDS_LD128_LD64(b4u, lbias + 16 * f + 4 * fk, rr2, slot);
        else PCM_LDS_LD128(b4u, lbias + 16 * f + 4 * fk);
        f32x4 a = acc[i][f] * g.alpha + f32x4{__uint_as_float(b4u.x), __uint_as_float(b4u.y), __uint_as_float(b4u.z), __uint_as_float(b4u.w)};
        if (has_res) {
          a[0] += bf2f((bf16_t)(rr2.x & 0xffff)); a[1] += bf2f((bf16_t)(rr2.x >> 16));
          a[2] += bf2f((bf16_t)(rr2.y & 0xffff)); a[3] += bf2f((bf16_t)(rr2.y >> 16));
        }
        u32x2 pk = u32x2{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
        PCM_LDS_ST64(slot, pk);
      }
    PCM_LDS_WAIT_ALL();
    PCM_WAVE_LDS_FENCE();
#pragma unroll
    for (int j = 0; j < 5; j++) {       // (one piece at a time: the K + 64 instantiation has no registers for five)
      const int p = lane + 64 * j, r = p / 5, c = p - r * 5;
      const int m = 64 * s + r;
      u32x4 q;
      PCM_LDS_LD128(q, stage + p * 16);
      if (m < g.M) *(u32x4*)((bf16_t*)g.out + (size_t)m * g.ldo + n0 + 8 * c) = q;
    }
  }
#endif
}

size_t pcm_gemm_ws_lds_bytes(int nk) { return 2 * ((size_t)64 * 640 + (size_t)64 * 64 * (nk - 10)) + 4 * (size_t)64 * 160 + 8 * 48 * 4; }      // (+ the eight-wave form's bias rows)

// preconditions: pcm_gemm_ws_ok (gemm.hip).  grid = (row blocks, N / 320); a block walks steps_per_block consecutive 64-row tiles
template <int NK, bool W8>
static int launch_ws(const GemmDev& g, void* stream) {
  const size_t smem = pcm_gemm_ws_lds_bytes(NK);          // (both forms: 4 x 64 x 160 = 8 x 64 x 80 bytes of staging)
  static bool lds_ok = false;
  if (!lds_ok) {
    hipError_t er = W8 ? hipFuncSetAttribute((const void*)pcm_gemm_ws8_kernel<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
                       : hipFuncSetAttribute((const void*)pcm_gemm_ws_kernel<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    PCM_CHECK(er == hipSuccess, PCM_EHIP, "pcm_gemm_bf16: hipFuncSetAttribute(LDS %zu): %s", smem, hipGetErrorString(er));
    lds_ok = true;
  }
  const int groups = g.N / 320, steps = (g.M + 63) / 64;
  int bx = PCM_GRID_CAP(256) / groups; if (bx < 1) bx = 1; if (bx > steps) bx = steps;
  const int spb = (steps + bx - 1) / bx;
  bx = (steps + spb - 1) / spb;
  if constexpr (W8) { PCM_LAUNCH((pcm_gemm_ws8_kernel<NK>), dim3(bx, groups), dim3(512), smem, stream, g, spb); }
  else { PCM_LAUNCH((pcm_gemm_ws_kernel<NK>), dim3(bx, groups), dim3(256), smem, stream, g, spb); }
  return 0;
}
int pcm_gemm_ws_launch(const GemmDev& g, void* stream, int form) {
#if !PCM_HAS_TOOLS
  // measured slower than the phased tile (header of gemm.hip's gemm_ws_ok): the product library does not instantiate the kernel
  PCM_CHECK(false, PCM_EUNSUPPORTED, "pcm_gemm_ws_launch: tools build only");
#else
  const int nk = (g.seg[0].K + (g.nseg > 1 ? g.seg[1].K : 0)) / 32;
  if (form == 2) {
    if (nk == 10) return launch_ws<10, true>(g, stream);      // (K + 64: 144 weight registers + 48 accumulators do not fit 256 without spills -- not instantiated)
  } else {
    if (nk == 10) return launch_ws<10, false>(g, stream);
    if (nk == 12) return launch_ws<12, false>(g, stream);
  }
  PCM_CHECK(false, PCM_EUNSUPPORTED, "pcm_gemm_ws_launch: K total %d", 32 * nk);
#endif
}
