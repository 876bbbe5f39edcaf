// pcm_lora_wgrad_bf16 — weight gradients of the rank-64 LoRA factors (the only trainable
// parameters of PCM-LoRA distillation):
//     G[g][r] += alpha * sum_m Big[m][g] * Small[m][r]
//   dB[n][r] : Big = dY [M][N],            Small = t = x A^T   [M][64]
//   dA[r][k] : Big = x  [M][K] or im2col,  Small = u = dY B    [M][64]
// The contraction index m is the SLOW dimension of both operands, so tiles are transposed on the
// way into LDS: each thread loads an 8(m) x 8(col) bf16 block (8 x 16 B), transposes it in
// registers and writes 8 x ds_write_b128 into a [col][m] image whose 16-B chunks are XOR-swizzled
// with ((row ^ (row>>3)) & 15) — conflict-free for both the transposed writes and the
// ds_read_b128 MFMA fragment reads.  Block = 64 cols of Big x 64 ranks, 128 rows of m per step,
// split over M across workgroups, fp32 atomics into the flat gradient buffer.
#include "wgrad_dev.h"

__device__ __forceinline__ int wg_off(int row, int chunk) { return row * 256 + ((chunk ^ ((row ^ (row >> 3)) & 15)) << 4); }

__global__ __launch_bounds__(256) void pcm_wgrad_kernel(WgDev a) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 64 * 256];
  char* Bt = lds;             // [64 g][128 m]
  char* St = lds + 64 * 256;  // [64 r][128 m]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int which = tid >> 7, u = tid & 127, mg = u >> 3, cg = u & 7;  // unit: rows 8mg.., cols 8cg..
  const int g0 = blockIdx.x * 64;
  const int m_begin = blockIdx.y * a.m_per_block;
  int m_end = m_begin + a.m_per_block; if (m_end > a.M) m_end = a.M;
  const int wg = wave & 1, wr = wave >> 1;
  const int frow = lane & 31, hi = lane >> 5;
  int tap_y = 0, tap_x = 0, ci0 = 0;
  if (a.mode == PCM_SEG_CONV3X3) {
    int tap = g0 / a.C; ci0 = g0 - tap * a.C; tap_y = tap / 3; tap_x = tap - tap_y * 3;
  }
  const int HoWo = a.Ho * a.Wo;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;

  // Register-staged pipeline: the NEXT 128-row chunk is fetched (unconditional loads from clamped
  // addresses; validity kept as a bit mask and applied at transpose time) while the MFMAs run on the
  // current one.
  uint4 rr[8];
  unsigned okmask = 0;
  const int gcol = g0 + 8 * cg;
  auto fetch = [&](int mc) {
    okmask = 0;
    int m = mc + 8 * mg;
    int b = 0, oy = 0, ox = 0;
    if (which == 0 && a.mode == PCM_SEG_CONV3X3) {
      int mm = m < a.M ? m : a.M - 1;
      b = mm / HoWo; int rem = mm - b * HoWo; oy = rem / a.Wo; ox = rem - oy * a.Wo;
    }
#pragma unroll
    for (int j = 0; j < 8; j++, m++) {
      bool ok = m < m_end;
      const bf16_t* p;
      if (which == 1) {
        p = a.small_ + (size_t)(ok ? m : m_begin) * a.lds_ + 8 * cg;
      } else if (a.mode == PCM_SEG_PLAIN) {
        ok = ok && gcol < a.G;
        p = a.big + (size_t)(ok ? m : m_begin) * a.ldb + (gcol < a.G ? gcol : 0);
      } else {
        int vy = oy * a.stride + tap_y - 1, vx = ox * a.stride + tap_x - 1;
        int sh = a.src_mode != PCM_SRC_DIRECT;
        ok = ok && vy >= 0 && vy < (a.Hs << sh) && vx >= 0 && vx < (a.Ws << sh);
        if (a.src_mode == PCM_SRC_ZEROINS2) ok = ok && !((vy | vx) & 1);
        int sy = ok ? (vy >> sh) : 0, sx = ok ? (vx >> sh) : 0;
        p = a.big + ((size_t)((ok ? b : 0) * a.Hs + sy) * a.Ws + sx) * a.C + ci0 + 8 * cg;
        // next output pixel (row-major over (b, oy, ox))
        ox++;
        if (ox == a.Wo) { ox = 0; oy++; if (oy == a.Ho) { oy = 0; b++; } }
      }
      rr[j] = *(const uint4*)p;
      okmask |= (ok ? 1u : 0u) << j;
    }
  };
  fetch(m_begin);
  for (int mc = m_begin; mc < m_end; mc += 128) {
    uint4 oo[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (!((okmask >> j) & 1)) rr[j] = make_uint4(0u, 0u, 0u, 0u);
    transpose8x8_bf16(rr, oo);
    __syncthreads();  // previous chunk's fragment reads are done
    char* T = which ? St : Bt;
#pragma unroll
    for (int e = 0; e < 8; e++) *(uint4*)(T + wg_off(8 * cg + e, mg)) = oo[e];
    __syncthreads();
    if (mc + 128 < m_end) fetch(mc + 128);
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      bf16x8 af = *(const bf16x8*)(Bt + wg_off(32 * wg + frow, 2 * ks + hi));
      bf16x8 bf = *(const bf16x8*)(St + wg_off(32 * wr + frow, 2 * ks + hi));
      // lanes run along the MFMA B operand's rows: put the operand whose output index is contiguous
      // in memory there, so the fp32 atomics of a wave touch 128-B runs
      acc = a.swap ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, af, acc, 0, 0, 0)
                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
    }
  }
  if (a.swap) {  // D[i = r][j = g]: lane -> g, regs -> r
    const int g = g0 + 32 * wg + frow;
    if (g < a.G) {
#pragma unroll
      for (int q = 0; q < 16; q++) {
        int r = 32 * wr + (q & 3) + 8 * (q >> 2) + 4 * hi;
        wg_emit(a, (size_t)g * a.g_stride + (size_t)r * a.r_stride, blockIdx.y, g, r, acc[q] * a.alpha);
      }
    }
    return;
  }
  // D[i = g][j = r]: lane -> r = 32wr + (lane&31); regs -> g = 32wg + (q&3) + 8(q>>2) + 4hi
  const int r = 32 * wr + frow;
#pragma unroll
  for (int q = 0; q < 16; q++) {
    int g = g0 + 32 * wg + (q & 3) + 8 * (q >> 2) + 4 * hi;
    if (g >= a.G) continue;
    size_t off;
    if (a.out_conv) {
      int tap = g / a.C, ci = g - tap * a.C;
      off = (size_t)r * 9 * a.C + (size_t)ci * 9 + tap;
    } else {
      off = (size_t)g * a.g_stride + (size_t)r * a.r_stride;
    }
    wg_emit(a, off, blockIdx.y, g, r, acc[q] * a.alpha);
  }
}

// reproducible form: out[off(g, r)] += sum over the M split's slabs, in split order (4 independent chains, combined in a fixed tree)
__global__ __launch_bounds__(256) void pcm_wgrad_finalize_kernel(const float* part, int msplit, int G, float* out, long g_stride, long r_stride,
                                                                 int out_conv, int C) {
  const long n = (long)G * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int by = 0;
    for (; by + 3 < msplit; by += 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) s[u] += part[(size_t)(by + u) * n + i];
    }
    for (int u = 0; by < msplit; by++, u++) s[u] += part[(size_t)by * n + i];
    const int g = (int)(i >> 6), r = (int)(i & 63);
    size_t off;
    if (out_conv) { const int tap = g / C, ci = g - tap * C; off = (size_t)r * 9 * C + (size_t)ci * 9 + tap; }
    else off = (size_t)g * g_stride + (size_t)r * r_stride;
    out[off] += (s[0] + s[1]) + (s[2] + s[3]);
  }
}

PCM_KNOB int g_wg_target = 512, g_wg_minchunks = 4, g_wg_auto = 1;
PCM_TOOLS_ONLY(extern "C" void pcm_debug_wgrad_grid(int target_blocks, int min_chunks) {   // tuning hook (tools/wgrad_probe.py); 0, 0 = shipped rule
  g_wg_auto = (target_blocks <= 0 && min_chunks <= 0) ? 1 : 0;
  g_wg_target = target_blocks > 0 ? target_blocks : 512;
  g_wg_minchunks = min_chunks > 0 ? min_chunks : 4;
})

static int wg_convert(const pcm_wgrad_args* p, WgDev& a) {
  PCM_CHECK(p && p->big && p->small_ && p->out && p->M > 0 && p->G > 0, PCM_EINVAL, "pcm_lora_wgrad_bf16: null/empty");
  PCM_CHECK(PCM_ALIGNED16(p->big) && PCM_ALIGNED16(p->small_) && (p->lds_ % 8) == 0 && p->lds_ >= 64, PCM_EALIGN,
            "pcm_lora_wgrad_bf16: operand alignment / small ld");
  memset(&a, 0, sizeof(a));
  a.big = (const bf16_t*)p->big; a.ldb = p->ldb; a.G = p->G; a.mode = p->mode; a.Hs = p->Hs; a.Ws = p->Ws; a.C = p->C;
  a.stride = p->stride; a.src_mode = p->src_mode; a.Ho = p->Ho > 0 ? p->Ho : 1; a.Wo = p->Wo > 0 ? p->Wo : 1;
  a.small_ = (const bf16_t*)p->small_; a.lds_ = p->lds_; a.M = p->M; a.out = p->out; a.g_stride = p->g_stride;
  a.r_stride = p->r_stride; a.out_conv = p->out_conv; a.alpha = p->alpha;
  a.swap = (!p->out_conv && p->g_stride < p->r_stride) ? 1 : 0;
  a.part = nullptr;
  if (p->mode == PCM_SEG_CONV3X3) {
    PCM_CHECK(p->C > 0 && (p->C % 64) == 0 && p->G == 9 * p->C && (p->M % (a.Ho * a.Wo)) == 0 && (p->stride == 1 || p->stride == 2),
              PCM_EINVAL, "pcm_lora_wgrad_bf16: conv view needs C%%64==0, G==9*C, M==B*Ho*Wo");
  } else {
    PCM_CHECK(p->mode == PCM_SEG_PLAIN && (p->G % 8) == 0 && (p->ldb % 8) == 0 && p->ldb >= p->G && !p->out_conv, PCM_EINVAL,
              "pcm_lora_wgrad_bf16: plain view needs G%%8==0, ldb%%8==0");
  }
  return PCM_OK;
}

// n independent weight gradients (pcm_hip.h): the plain-view jobs share launches (wgrad_tr.hip), the others run as single calls
extern "C" int pcm_lora_wgrad_multi_bf16(const pcm_wgrad_args* list, int n, void* stream) {
  PCM_CHECK(list && n > 0 && n <= 64, PCM_EINVAL, "pcm_lora_wgrad_multi_bf16: 1..64 jobs");
  WgDev jobs[64];
  unsigned char taken[64];
  for (int i = 0; i < n; i++) {
    if (int rc = wg_convert(list + i, jobs[i])) return rc;
    jobs[i].part = (float*)list[i].workspace;     // reproducible-form jobs are refused by the shared-launch planner and run as single calls
  }
  if (int rc = pcm_wgrad_tr_launch_multi(jobs, n, taken, stream)) return rc;
  for (int i = 0; i < n; i++)
    if (!taken[i])
      if (int rc = pcm_lora_wgrad_bf16(list + i, stream)) return rc;
  return pcm_post_launch("pcm_lora_wgrad_multi_bf16");
}

// the M split of the register-transposing kernel (fills a.m_per_block)
static int wg_plan_split(const pcm_wgrad_args* p, WgDev& a, int* tiles_g_out) {
  int tiles_g = (p->G + 63) / 64;
  int chunks = (p->M + 127) / 128;
  // M split.  Every block ends with 4096 fp32 atomics, i.e. msplit*G*256 B of atomic traffic against M*G*2 B of operand reads:
  // (1) split as far as needed for ~2 blocks per CU (>= 4 chunks per block), (2) go on to ~8 blocks per CU only while the atomics
  // stay below ~8 % of the reads (msplit <= M/1536).  Measured (tools/wgrad_probe.py, bs 16): 64x64x960 conv dA 385 -> 290 us,
  // 64x64x320 131 -> 113 us, [65536 x 2560] dB 115 -> 98 us; shapes with M <= 4096 keep the first rule.
  auto cdiv = [](int x, int y) { return (x + y - 1) / y; };
  int msplit = cdiv(PCM_GRID_CAP(g_wg_target), tiles_g);
  if (msplit > cdiv(chunks, g_wg_minchunks)) msplit = cdiv(chunks, g_wg_minchunks);
  if (g_wg_auto) {
    int extra = cdiv(PCM_GRID_CAP(2048), tiles_g);
    if (extra > p->M / 1536) extra = p->M / 1536;
    if (extra > cdiv(chunks, 2)) extra = cdiv(chunks, 2);
    if (extra > msplit) msplit = extra;
  }
  if (msplit > chunks) msplit = chunks;
  if (msplit < 1) msplit = 1;
  a.m_per_block = ((chunks + msplit - 1) / msplit) * 128;
  *tiles_g_out = tiles_g;
  return (p->M + a.m_per_block - 1) / a.m_per_block;
}
// slabs of the reproducible form = the M split the call would take (same planner as the launch; > 0, or < 0 on error)
static int wg_msplit(const pcm_wgrad_args* p, WgDev& a) {
  int ms = 0;
  const int rc = pcm_wgrad_tr_launch(a, nullptr, &ms, true);
  if (rc < 0) return rc;
  if (rc == 0) return ms;
  int tg;
  return wg_plan_split(p, a, &tg);
}
extern "C" size_t pcm_lora_wgrad_workspace_bytes(const pcm_wgrad_args* p) {
  WgDev a;
  if (wg_convert(p, a)) return 0;
  const int ms = wg_msplit(p, a);
  return ms > 0 ? (size_t)ms * p->G * 64 * sizeof(float) : 0;
}

extern "C" int pcm_lora_wgrad_bf16(const pcm_wgrad_args* p, void* stream) {
  WgDev a;
  if (int rc = wg_convert(p, a)) return rc;
  int msplit = 0;
  if (p->workspace) {
    WgDev t = a;
    msplit = wg_msplit(p, t);
    if (msplit < 0) return msplit;
    PCM_CHECK(((uintptr_t)p->workspace % 16) == 0 && p->workspace_bytes >= (size_t)msplit * p->G * 64 * sizeof(float), PCM_EINVAL,
              "pcm_lora_wgrad_bf16: workspace too small (%zu < %zu)", p->workspace_bytes, (size_t)msplit * p->G * 64 * sizeof(float));
    a.part = (float*)p->workspace;
  }
  auto finalize = [&]() {
    if (!a.part) return;
    const long n = (long)p->G * 64;
    long blocks = (n + 255) / 256; if (blocks > PCM_GRID_CAP(1024)) blocks = PCM_GRID_CAP(1024);
    PCM_LAUNCH(pcm_wgrad_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)a.part, msplit, p->G, a.out, a.g_stride, a.r_stride,
               a.out_conv, a.C);
  };
  {   // LDS-DMA + transpose-read kernels (wgrad_tr.hip) for the plain and the stride-1 3x3 views; everything else stays here
    const int rc = pcm_wgrad_tr_launch(a, stream);
    if (rc == 0) { finalize(); return pcm_post_launch("pcm_lora_wgrad_bf16"); }
    if (rc < 0) return rc;
  }
  int tiles_g;
  const int ms = wg_plan_split(p, a, &tiles_g);
  PCM_LAUNCH(pcm_wgrad_kernel, dim3(tiles_g, ms), dim3(256), 0, stream, a);
  finalize();
  return pcm_post_launch("pcm_lora_wgrad_bf16");
}
