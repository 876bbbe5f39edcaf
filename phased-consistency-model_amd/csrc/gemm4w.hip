// pcm_gemm4w_kernel -- the SHORT-K member of the pcm_gemm_bf16 family: Linear / conv1x1 launches (plain segments, optional LoRA K-segment)
// whose K loop is too short to amortise a tile's first-load round trip and its epilogue.
//
// Why a third kernel.  gemm8p.hip owns a whole CU (8 waves, 147 KB of LDS, one 256 x 320 tile): at K = 320 a tile spends ~5 us in MFMAs and
// ~18 us in {first LDS-DMA round trip, 4-pass LDS-staged epilogue, 160-480 KB of stores / residual reads at the per-CU share of the fabric
// rate}, and nothing can overlap that from outside the workgroup (round 2: a persistent tile loop does not help either, loads and stores
// retire through one in-order vmcnt).  This kernel is built to run TWO workgroups per CU instead: 4 waves (one per SIMD), a 128 x (64*FN)
// tile, 56 KB of LDS, <= 256 VGPRs -- the hardware interleaves one workgroup's epilogue (stores, residual loads, GEGLU) and prologue with
// the other's MFMA phases, wave by wave on every SIMD.
//
//   * per-wave tile 128 x (16*FN) in 8 x FN 16x16x32 accumulators, exactly gemm8p's (weights as the A operand: a lane owns 4 consecutive
//     channels of one pixel row); waves as 1 (M) x 4 (N): all four read the same 128 activation rows from LDS.
//   * BK = 32: a K-step is 128 x 64 B of activations + (64*FN) x 64 B of weights = 28 KB, staged by LDS-DMA (16 rows x 64 B per
//     instruction, 2 + FN instructions per wave), two stages.  16-B chunk c of row r lives at slot c ^ ((-(r >> 2)) & 3): with 64-B rows four
//     rows share a 256-B bank row, and this permutation makes every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS table) hit 16
//     distinct slots; the swizzle is applied to the per-lane SOURCE address (the LDS-DMA destination is lane-linear).
//   * K-step t:  {13 fragment reads of stage t&1; lgkmcnt(0); barrier; issue K-step t+2 into the stage just read; 8*FN MFMAs;
//     vmcnt(2+FN) = "K-step t+1 landed"; barrier}.  The DMA of t+2 flies under the MFMAs of t and t+1; no vmcnt(0) in steady state.
//   * epilogue: the fp32 tile goes through LDS in four 32-row passes for 16-B coalesced bf16 stores (bias, row vector, SiLU, residual,
//     fused GEGLU with the optional pre-activation output) -- same per-piece code as gemm8p.
#include "gemm_dev.h"
#include "gemm_epilogue.h"

#define PCM_RSRC_FLAGS 0x00020000
#define PCM_OOB 0x80000000u

// cycle stamps (tools/gemm4w_ablate.py, -DPCM_ABLATE builds only): lane 0 of every wave of ONE mid-grid tile records s_memtime at
// {entry, first K-step landed, K loop done, after each epilogue pass}; slot 7 = HW_ID of the wave
#ifdef PCM_ABLATE
__device__ unsigned long long g_g4_stamps[4][8];
extern "C" int pcm_debug_gemm4w_stamps(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_g4_stamps), sizeof(g_g4_stamps)); }
#define G4_STAMP(k) do { if (stamp_on && lane == 0) g_g4_stamps[wn][k] = __builtin_readcyclecounter(); } while (0)
#else
#define G4_STAMP(k) do { } while (0)
#endif

template <int FN>
__global__ __launch_bounds__(256, 2) void pcm_gemm4w_kernel(GemmDev g) {
#if PCM_KERNEL_BODY   // the host pass only needs the launch stub (buffer-resource builtins are device-only)
  constexpr int WNC = 16 * FN, BN = 4 * WNC, BM = 128;
  constexpr int OFF_B = BM * 64, STAGE = (BM + BN) * 64;
  constexpr int AI = BM / 64, WI = BN / 64;       // LDS-DMA instructions per wave and K-step: 16 rows x 64 B each
  PCM_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  int nwg = g.tiles_m * g.tiles_n, bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = bid % g.tiles_n, tile_m = bid / g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
#ifdef PCM_ABLATE
  const bool stamp_on = blockIdx.x == gridDim.x / 2;
#endif
  if (g.w4_stagger > 0) {
    // HW_REG_HW_ID (id 4): bits 19:16 = TG_ID, the slot of this workgroup on its CU.  The two co-resident workgroups of a CU hold slots of
    // opposite parity; the odd one starts late, which turns "both in their MFMA phases, then both in their epilogues" into alternation.
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
#ifdef PCM_ABLATE
    if (stamp_on && lane == 0) g_g4_stamps[wn][7] = hwid;
#endif
    if ((hwid >> 16) & 1)
      for (int i = 0; i < g.w4_stagger; i++) __builtin_amdgcn_s_sleep(32);
  }
  G4_STAMP(0);

  // ---- loader: lane -> row lane>>2 of a 16-row group, 16-B slot lane&3; this wave owns groups wn + 4j of both operands
  const int lrow = lane >> 2;
  const unsigned csw16 = (unsigned)(((lane & 3) ^ ((0 - (lrow >> 2)) & 3)) << 4);
  int it_seg = 0, it_k = 0;                         // K-step iterator of the loader (wave-uniform)
  SegDev cs = g.seg[0];
  int ksteps = cs.K >> 5;
  const int T = (g.seg[0].K >> 5) + (g.nseg > 1 ? (g.seg[1].K >> 5) : 0);
  unsigned a_voff[AI], w_voff[WI];
  auto prepare = [&]() {
#pragma unroll
    for (int j = 0; j < AI; j++) {
      const int m = m0 + 16 * (wn + 4 * j) + lrow;
      a_voff[j] = m < g.M ? (unsigned)m * (unsigned)(cs.lda * 2) + csw16 : PCM_OOB;
    }
#pragma unroll
    for (int j = 0; j < WI; j++) {
      const int n = n0 + 16 * (wn + 4 * j) + lrow;
      w_voff[j] = n < g.N ? (unsigned)n * (unsigned)(cs.K * 2) + csw16 : PCM_OOB;
    }
  };
  bool dma_on = true;
  auto issue = [&](int stage) {
    if (PCM_ABL(8) && !dma_on) { it_k++; return; }
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)cs.a, 0, PCM_OOB, PCM_RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)cs.w, 0, PCM_OOB, PCM_RSRC_FLAGS);
    char* dst = smem + stage * STAGE + wn * 1024;
    const unsigned soff = (unsigned)(it_k * 64);
#pragma unroll
    for (int j = 0; j < AI; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, PCM_AS3(dst + j * 4096), 16, a_voff[j], soff, 0, 0);
#pragma unroll
    for (int j = 0; j < WI; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, PCM_AS3(dst + OFF_B + j * 4096), 16, w_voff[j], soff, 0, 0);
    it_k++;
    if (it_k == ksteps) {
      it_k = 0; it_seg++;
      if (it_seg < g.nseg) { cs = g.seg[1]; ksteps = cs.K >> 5; prepare(); }
    }
  };

  f32x4 acc[8][FN];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int f = 0; f < FN; f++) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  prepare();
  issue(0);
  if (1 < T) {
    issue(1);
    if constexpr (FN == 5) PCM_WAIT_VMCNT(7); else PCM_WAIT_VMCNT(6);
  } else {
    PCM_WAIT_VMCNT(0);
  }
  __builtin_amdgcn_s_barrier();
  G4_STAMP(1);
  dma_on = false;

  // fragment reads: lane (frow = lane&15, fk = lane>>4) reads row base+frow, logical 16-B chunk fk (k = 8*fk .. 8*fk+7); every base is a
  // multiple of 16, so the swizzle key depends on frow only
  const int frow = lane & 15, fk = lane >> 4;
  const int foff = frow * 64 + ((fk ^ ((0 - (frow >> 2)) & 3)) << 4);
  bf16x8 af[8], bfr[FN];
  for (int t = 0; t < T; t++) {
    const char* cur = smem + (t & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < 8; i++) af[i] = *(const bf16x8*)(cur + i * 1024 + foff);
#pragma unroll
    for (int f = 0; f < FN; f++) bfr[f] = *(const bf16x8*)(cur + OFF_B + (WNC * wn + 16 * f) * 64 + foff);
    PCM_WAIT_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                 // every wave holds its fragments of stage t&1: the stage may be overwritten
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < T) issue(t & 1);
    __builtin_amdgcn_s_setprio(1);
    if (!PCM_ABL(4))
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int f = 0; f < FN; f++) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[f], af[i], acc[i][f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (t + 2 < T) {                              // K-step t+1 landed; the pieces of t+2 issued above stay in flight
      if constexpr (FN == 5) PCM_WAIT_VMCNT(7); else PCM_WAIT_VMCNT(6);
    } else {
      PCM_WAIT_VMCNT(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                 // ... for every wave's pieces: stage (t+1)&1 is readable
    __builtin_amdgcn_sched_barrier(0);
  }

  G4_STAMP(2);
  if (PCM_ABL(2)) { if (g.alpha != 123456.f) return; }
  // ---- epilogue.  lane owns pixel row (lane&15) of a fragment and 4 consecutive channels 4*(lane>>4)+r.
  // four passes of 32 rows x BN fp32 through LDS (the K-loop stages are dead; the loop's last barrier already separates them from the
  // fragment reads), shared with gemm8p.hip: gemm_epilogue.h
  PcmEpi<FN, 1>::run(g, smem, acc, tid, 0, wn, m0, n0, false);
  G4_STAMP(6);
#endif
}

// two K stages, or the bf16 staging of the fused-GEGLU epilogue (64 rows of outputs + pre-activations), whichever is larger
size_t pcm_gemm4w_lds_bytes(int fn) {
  const size_t kloop = 2 * (size_t)(128 + 64 * fn) * 64, geglu = fn == 5 ? PcmEpi<5, 1>::geglu_lds_bytes() : PcmEpi<4, 1>::geglu_lds_bytes();
  return kloop > geglu ? kloop : geglu;
}

template <int FN>
static int launch4w(const GemmDev& g, void* stream) {
  PCM_LAUNCH((pcm_gemm4w_kernel<FN>), dim3(g.tiles_m * g.tiles_n), dim3(256), pcm_gemm4w_lds_bytes(FN), stream, g);
  return 0;
}
int pcm_gemm4w_launch(const GemmDev& g, int fn, void* stream) { return fn == 5 ? launch4w<5>(g, stream) : launch4w<4>(g, stream); }
