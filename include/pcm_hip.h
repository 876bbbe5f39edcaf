/* pcm_hip.h — C ABI of libpcm_hip.so: the MI355X (gfx950) kernels behind the phased-consistency
 * distillation step of G-U-N/Phased-Consistency-Model (SD1.5 PCM-LoRA).
 *
 * Drop-in boundary.  The reference has no FFI of its own: its operator surface is
 * the set of torch/diffusers/peft calls reached from
 *   code/text_to_image_sd15/train_pcm_lora_sd15.py:1115-1301  (the step)
 *   code/text_to_image_sd15/discriminator_sd15.py:84-345       (UNet wiring, copied from diffusers)
 * Each entry point below names the reference call site / library op it replaces.
 *
 * Contract
 *  - plain pointers + sizes; the CALLER owns every buffer; the library never allocates or frees
 *    device memory.  Global state of the product libraries (libpcm_hip.so, libpcm_hip_f16.so): one immutable 16-byte
 *    zero page in the code object (out-of-bounds LDS-DMA lanes) and once-only hipFuncSetAttribute flags -- nothing
 *    mutable, no environment variable is read, no pcm_debug_* symbol is exported (`nm -D` shows none).  The A/B
 *    switches, tile-forcing hooks, launch counters and environment switches of the development rounds exist only in
 *    the TOOLS build of the same sources (-DPCM_TOOLS -> libpcm_hip_tools.so; csrc/pcm_common.h PCM_KNOB / PCM_LAZY_KNOB:
 *    process-wide, not thread-safe), which tools/ and the hook-using tests load; its default behaviour is the product's.
 *  - every call enqueues on `stream` (a hipStream_t passed as void*; torch's current stream)
 *    and returns without synchronising; safe under hipGraph stream capture.
 *  - return 0 on success, a negative PCM_E* code otherwise; never throws.  pcm_last_error()
 *    returns a thread-local message.
 *  - layouts: activations channels-last  [B, H*W, C]  (== token layout [B, L, C]); bf16 as raw
 *    uint16; weights bf16 [N][K] with K contiguous (conv: K = (kh, kw, ci), ci fastest);
 *    LoRA master weights stay in peft layout (A [r,in(,k,k)], B [out,r(,1,1)]) in fp32 — the
 *    bf16 operand copies are produced by pcm_pack_* below.
 */
#ifndef PCM_HIP_H
#define PCM_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PCM_OK 0
#define PCM_EINVAL (-1)      /* bad shape / argument */
#define PCM_EALIGN (-2)      /* pointer or stride not 16-byte aligned */
#define PCM_EHIP (-3)        /* HIP runtime error at launch */
#define PCM_EUNSUPPORTED (-4)

#define PCM_BF16 0
#define PCM_F32 1

#define PCM_ACT_NONE 0
#define PCM_ACT_SILU 1
#define PCM_ACT_LEAKY 2 /* LeakyReLU(0.01): DiscriminatorHead, discriminator_sd15.py:354 */
/* GEGLU fused into the projection (diffusers GEGLU.forward: hidden, gate = proj(x).chunk(2); hidden * gelu(gate)).
 * The weight rows (and bias) must be packed INTERLEAVED in groups of 2: [v0, v1, g0, g1, v2, v3, g2, g3, ...] (N = 2*inner rows;
 * abi >= 3 -- abi 2 interleaved in groups of 8): the four consecutive output channels one MFMA lane accumulates are then the two
 * values and the two gates of one output pair, and value * gelu(gate) is formed in registers.  The output has N/2 columns
 * (ldo >= N/2).  No residual / row vector; bf16 output only.  Optional second output: pre_out. */
#define PCM_ACT_GEGLU 3

const char* pcm_last_error(void);
int pcm_abi_version(void);
/* identity of the kernel sources this library was built from (abi >= 5): "<16 hex digits of sha256 over csrc/*.hip, csrc/*.h,
 * include/pcm_hip.h>-<variant>"; the host side refuses a default library whose id is not its tree's, bench.py / smoke() print it */
const char* pcm_build_id(void);
/* The 16-bit storage / MFMA operand format of THIS build of the library: PCM_FMT_BF16 (libpcm_hip.so, the default and what bench.py
 * measures) or PCM_FMT_F16 (libpcm_hip_f16.so: the same sources compiled with -DPCM_ACT_F16 -- IEEE half, for the reference's
 * --mixed_precision=fp16 recipes, train_pcm_lora_sd15.sh:9, and for the 1e-3 loss validation against the fp32 oracle).  Wherever this
 * header says "bf16" for a buffer it means "the library's 16-bit format"; entry-point names do not change.  fp32 / fp64 arguments
 * (master weights, accumulators, statistics, losses) are the same in both. */
#define PCM_FMT_BF16 0
#define PCM_FMT_F16 1
int pcm_act_dtype(void);

/* ---- contraction: Linear / conv1x1 / conv3x3 (implicit GEMM) + LoRA injection --------------
 * replaces F.linear / F.conv2d inside diffusers ResnetBlock2D.conv1/conv2, Attention.to_q/k/v/
 * to_out.0, FeedForward, proj_in/out, Down/Upsample2D.conv and peft lora.Linear/lora.Conv2d
 * ( y = base(x) + (alpha/r) * B(A(x)), train_pcm_lora_sd15.py:866-885 ), forward and dgrad.
 *
 * out[m][n] = act( sum_seg sum_k A_seg[m][k] * W_seg[n][k] + bias[n] + rowvec[m / rows_per_batch][n] )
 *             + residual[m][n]
 * A segment is either a plain row-major matrix or an implicit im2col view of an NHWC tensor. */
#define PCM_SEG_PLAIN 0
#define PCM_SEG_CONV3X3 1
#define PCM_SRC_DIRECT 0    /* conv reads the source tensor as is */
#define PCM_SRC_UPSAMPLE2 1 /* conv runs on the nearest-2x upsampling of the source (Upsample2D) */
#define PCM_SRC_ZEROINS2 2  /* conv runs on the zero-insertion-2x of the source (stride-2 dgrad) */
typedef struct {
  const void* a; /* bf16: plain [M][lda]  |  conv NHWC [B][Hs][Ws][C] */
  const void* w; /* bf16 [N][K] */
  int K;         /* plain: columns, %8==0 ; conv: 9*C with C%64==0 */
  int lda;       /* plain only (elements) */
  int mode;      /* PCM_SEG_* */
  int Hs, Ws, C; /* conv: SOURCE tensor dims */
  int stride;    /* conv: 1 or 2 (output pixel -> input pixel step, on the virtual input) */
  int src_mode;  /* PCM_SRC_* */
} pcm_gemm_seg;

typedef struct {
  int M, N;
  int Ho, Wo;             /* conv: output spatial dims (M = B*Ho*Wo); plain: ignored */
  const float* bias;      /* [N] or NULL */
  const void* rowvec;     /* bf16 [M/rows_per_batch][N] or NULL (time-embedding add, :ResnetBlock2D) */
  int rows_per_batch;
  const void* residual;   /* bf16 [M][ldr] or NULL; added AFTER act */
  int ldr;
  void* out;              /* [M][ldo] */
  int ldo;
  int out_dtype;          /* PCM_BF16 | PCM_F32 */
  int act;                /* PCM_ACT_* */
  float alpha;            /* scales the accumulated sum before bias (1.0 normally) */
  void* workspace;        /* optional fp32 scratch for split-K (under-filled grids with long K); NULL = never split */
  size_t workspace_bytes; /* >= pcm_gemm_workspace_bytes(...) */
  /* PCM_ACT_GEGLU only (abi >= 2): besides the activated output the call can keep the PRE-activation (sum + bias, bf16, interleaved
   * column order of the packed weight, N columns) of rows m < pre_rows in pre_out[m][ldp] -- what autograd saves for GEGLU.backward
   * (the grad-requiring half of a batch); pcm_geglu_bwd_interleaved reads it.  NULL / 0: nothing kept. */
  void* pre_out;
  int pre_rows, ldp;
  /* abi >= 5 -- two fusions of the glue around a contraction (diffusers' UNet wiring, discriminator_sd15.py:264-342):
   *  out2 / ldo2: a SECOND copy of the bf16 output rows, out2[m][ldo2] (same values as out).  The producer of a down-path skip tensor
   *    writes it straight into the channel range of the buffer the up path will read as torch.cat([h, skip], dim=1): no concat pass
   *    (bf16 output only, not with PCM_ACT_GEGLU; ldo2 % 4 == 0, 8-byte aligned).  NULL: none.
   *  chstats / stats_rows: per-(sample, channel) {sum, sum of squares} of the STORED (16-bit-rounded) output values, accumulated with fp64
   *    atomics into chstats[m / stats_rows][n][2] (pre-zeroed by the caller; stats_rows = rows per sample, a multiple of 64): what the
   *    GroupNorm that reads this tensor next needs (pcm_groupnorm_apply_chstats), so no statistics pass over it.  Only where
   *    pcm_gemm_emits_chstats() says so for the same arguments -- other plans ignore the field and the caller runs pcm_groupnorm_stats.
   *    NULL: none. */
  void* out2;
  int ldo2;
  double* chstats;
  int stats_rows;
} pcm_gemm_epi;

/* bytes of `workspace` the call would use (0: no split-K for this shape) */
size_t pcm_gemm_workspace_bytes(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* epi);
int pcm_gemm_bf16(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* epi, void* stream);
/* Which kernel family and K split the call above takes for these arguments (a pure function of them, nothing is launched; abi >= 4):
 *   1000 * F + splitk   the phased 8-wave tile gemm8p (F = 4: 256x256, F = 5: 256x320) or, F = 0, a 4-wave tile of gemm.hip;
 *   10000 + 1000 * F + 1  gemm4w (two workgroups per CU);  64 the rank-64 streaming kernels;  65 the 3x3 halo-window rank-64 kernel;
 *   32 the batch-row kernel (M <= 16);  30000 + K/32 the weights-stationary kernel (abi 5: short-K projections, N % 320 == 0).  Negative: the PCM_E* code the call would return.  bench.py's roofline leg classes its timed
 * launches with it. */
int pcm_gemm_plan_code(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* epi);
/* 1 when pcm_gemm_bf16 with these arguments (epi->chstats set or not: it is not read) accumulates the per-channel statistics described at
 * pcm_gemm_epi.chstats, 0 when the plan it takes does not (the caller then runs the statistics pass), < 0: the PCM_E* code (abi >= 5) */
int pcm_gemm_emits_chstats(const pcm_gemm_seg* segs, int nseg, const pcm_gemm_epi* epi);

/* LoRA weight gradients (autograd of peft lora.Linear / lora.Conv2d under loss.backward(),
 * train_pcm_lora_sd15.py:1296).  G[g][r] += alpha * sum_m Big[m][g] * Small[m][r], r in [0,64).
 * Big is plain [M][ldb] (G columns) or the im2col view of an NHWC tensor; Small is [M][64].
 * Accumulates with fp32 atomics into `out` at  out[g*g_stride + r*r_stride]  (the kernel puts the
 * contiguous one of the two output indices along the wave's lanes) or, for out_conv=1 (peft Conv2d A
 * layout [r][C][3][3], uncoalesced):  g=(tap,ci) -> out[r*9*C + ci*9 + tap]. */
typedef struct {
  const void* big; int ldb; int G;      /* plain: G columns (%8==0) */
  int mode; int Hs, Ws, C, stride, src_mode, Ho, Wo; /* conv view (G = 9*C) */
  const void* small_; int lds_;         /* bf16 [M][lds_], 64 columns used */
  int M;
  float* out; long g_stride, r_stride; int out_conv;
  float alpha;
  /* abi >= 4 -- reproducible form.  NULL / 0: the M split's partial sums meet in `out` as fp32 atomics (fastest; the summation ORDER, and
   * with it the last bits of the gradient, changes from run to run).  A workspace of >= pcm_lora_wgrad_workspace_bytes(a): every
   * block of the M split stores its partial tile to its own slab and an ordered finalize launch adds the slabs to `out` in split order:
   * bitwise identical results run to run (what cuDNN/cuBLAS "deterministic algorithms" buy the reference under
   * torch.use_deterministic_algorithms; costs the slab round trip and one more launch per job). */
  void* workspace; size_t workspace_bytes;
} pcm_wgrad_args;
size_t pcm_lora_wgrad_workspace_bytes(const pcm_wgrad_args* a);   /* bytes the reproducible form of this job needs (0: bad arguments) */
int pcm_lora_wgrad_bf16(const pcm_wgrad_args* a, void* stream);
/* n (1..64) independent jobs of the call above in as few launches as possible: the weight gradients of one autograd node (lora_A and
 * lora_B of a module; the six of a fused q/k/v projection) are 3-20 us kernels each, so they share launches.  Same result as n calls. */
int pcm_lora_wgrad_multi_bf16(const pcm_wgrad_args* list, int n, void* stream);   /* jobs that carry a workspace run one by one (the same workspace may serve them all) */

/* ---- Dense conv3x3 weight gradient, channels-last  (autograd of the trainable nn.Conv2d(C, C, 3, 1, 1) layers of DiscriminatorHead,
 * discriminator_sd15.py:349-362, in the discriminator step train_pcm_lora_sd15_adv.py:1383-1391) --------------------------------
 *   dW[co][kh][kw][ci] += alpha * sum_{b,y,x} dy[b][y][x][co] * x[b][y+kh-1][x+kw-1][ci]        (stride 1, zero padding 1)
 * x [B,H,W,Cin] bf16, dy [B,H,W,Cout] bf16 (both dense), dW fp32 in the library's [co][kh][kw][ci] layout (accumulated into: the caller
 * zeroes it once per optimizer step).  Needs H%8 == 0, W%8 == 0, Cin%8 == 0, Cout%64 == 0 and each operand below 2 GB; PCM_EINVAL
 * otherwise (the Cout/64 rank-64 calls of pcm_lora_wgrad_bf16 compute the same sum for any geometry). */
int pcm_conv3x3_wgrad_bf16(const void* x, const void* dy, float* dW, int B, int H, int W, int Cin, int Cout, float alpha, void* stream);

/* ---- GroupNorm(32)(+SiLU), channels-last  (diffusers ResnetBlock2D.norm1/2, conv_norm_out,
 * Transformer2DModel.norm) --------------------------------------------------------------- */
/* stats[b][g] = {sum, sumsq} in fp64, zeroed by the call itself */
int pcm_groupnorm_stats(const void* x, double* stats, int B, int HW, int C, int G, void* stream);
int pcm_groupnorm_apply(const void* x, const double* stats, const float* gamma, const float* beta,
                        void* y, int B, int HW, int C, int G, float eps, int act, void* stream);
/* abi >= 5: the same apply when the producing contraction(s) accumulated per-(sample, channel) sums (pcm_gemm_epi.chstats, fp64 {sum, sumsq}):
 * channel c < c_split of sample b reads chstats[(b * stats_ld + c) * 2 ..], channel c >= c_split reads chstats2[(b * stats_ld2 + c - c_split) * 2 ..]
 * (x = torch.cat([h, skip], dim=1) of the up blocks: two producers; chstats2 NULL: one source for all C channels).  Group statistics are
 * formed from them in the call (no statistics pass over x) and also written to stats_out[B][G][2] (may be NULL) in the format of
 * pcm_groupnorm_stats, which the backward entry points read */
int pcm_groupnorm_apply_chstats(const void* x, const double* chstats, int stats_ld, const double* chstats2, int stats_ld2, int c_split,
                                double* stats_out, const float* gamma, const float* beta, void* y, int B, int HW, int C, int G,
                                float eps, int act, void* stream);
/* backward wrt x only (norm affine params are frozen): two launches */
int pcm_groupnorm_bwd_stats(const void* x, const void* dy, const double* stats, const float* gamma,
                            const float* beta, double* bstats, int B, int HW, int C, int G,
                            float eps, int act, void* stream);

/* Same two reductions, but ACCUMULATING into a buffer the caller has already zeroed (e.g. one slice of an arena cleared by a single
 * memset per network pass: the SD1.5 step runs 180 GroupNorm statistics passes). */
int pcm_groupnorm_stats_acc(const void* x, double* stats, int B, int HW, int C, int G, void* stream);
int pcm_groupnorm_bwd_stats_acc(const void* x, const void* dy, const double* stats, const float* gamma, const float* beta, double* bstats,
                                int B, int HW, int C, int G, float eps, int act, void* stream);
/* Contention-free form of the two reductions (the one the UNet runner uses): every workgroup writes its partial sums into the
 * caller's `workspace` and a second, tiny launch adds them up into stats[b][g] (written, not accumulated).  Same-address fp64
 * atomics cost ~0.5 us each on MI355X and serialize per (b, g), which bounds the atomic form to ~2 workgroups per CU. */
size_t pcm_groupnorm_workspace_bytes(int B, int HW, int C, int G);
int pcm_groupnorm_stats_ws(const void* x, double* stats, int B, int HW, int C, int G, void* workspace, size_t workspace_bytes, void* stream);
int pcm_groupnorm_bwd_stats_ws(const void* x, const void* dy, const double* stats, const float* gamma, const float* beta, double* bstats,
                               int B, int HW, int C, int G, float eps, int act, void* workspace, size_t workspace_bytes, void* stream);
int pcm_groupnorm_bwd_apply(const void* x, const void* dy, const double* stats, const double* bstats,
                            const float* gamma, const float* beta, void* dx, int B, int HW, int C,
                            int G, float eps, int act, void* stream);
/* abi >= 5: the same with the gradient arriving over the block's skip path added in the same pass, dx = gn_dx + dres (dres bf16 [B][HW][C] or
 * NULL): the separate add kernel of ResnetBlock2D / Transformer2DModel's residual backward is not needed */
int pcm_groupnorm_bwd_apply_res(const void* x, const void* dy, const double* stats, const double* bstats, const float* gamma,
                                const float* beta, const void* dres, void* dx, int B, int HW, int C, int G, float eps, int act, void* stream);

/* affine-parameter gradients (the discriminator heads' norms are trainable): dgamma/dbeta fp32 [C], ACCUMULATED */
int pcm_groupnorm_param_grad(const void* x, const void* dy, const double* stats, const float* gamma,
                             const float* beta, float* dgamma, float* dbeta, int B, int HW, int C, int G,
                             float eps, int act, void* stream);
/* abi >= 5, reproducible form of the call above (workspace >= pcm_groupnorm_param_grad_workspace_bytes) */
size_t pcm_groupnorm_param_grad_workspace_bytes(int B, int HW, int C, int G);
int pcm_groupnorm_param_grad_ws(const void* x, const void* dy, const double* stats, const float* gamma, const float* beta, float* dgamma,
                                float* dbeta, int B, int HW, int C, int G, float eps, int act, void* workspace, size_t workspace_bytes, void* stream);

/* ---- LayerNorm over the last dim (BasicTransformerBlock.norm1/2/3) ----------------------- */
int pcm_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, int M, int C, float eps, void* stream);
/* dx = LN backward (+ optional accumulate of `dres` into dx: the residual-branch gradient) */
int pcm_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean,
                      const float* rstd, const void* dres, void* dx, int M, int C, void* stream);

/* ---- GEGLU (diffusers GEGLU: h, g = proj(x).chunk(2,-1); h * gelu_erf(g)) ---------------- */
int pcm_geglu_fwd(const void* hg, void* out, int M, int C4, void* stream);           /* hg [M][2*C4] */
int pcm_geglu_bwd(const void* hg, const void* dout, void* dhg, int M, int C4, void* stream);
/* the same gradient from the INTERLEAVED pre-activation kept by a fused PCM_ACT_GEGLU projection (pcm_gemm_epi.pre_out, row stride ldp);
 * dhg comes out in the standard [values | gates] column order */
int pcm_geglu_bwd_interleaved(const void* pre, int ldp, const void* dout, void* dhg, int M, int C4, void* stream);

/* ---- scaled-dot-product attention (Attention.processor: torch SDPA / xformers
 * memory_efficient_attention, train_pcm_lora_sd15.py:947-957) ------------------------------
 * q [B][Lq][ldq], k/v [B][Lk][ldk], head h occupies columns [h*d, (h+1)*d); o like q.
 * lse [B][H][Lq] fp32 in the log2 domain of the scaled scores. d in {40, 80, 160} (or any
 * multiple of 8 up to 160). */
int pcm_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H,
                 int Lq, int Lk, int d, int ldq, int ldk, int ldo, float scale, void* stream);
int pcm_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO,
                 const float* lse, float* delta /*[B][H][Lq] scratch*/, void* dq, void* dk, void* dv,
                 int B, int H, int Lq, int Lk, int d, int ldq, int ldk, int ldo, float scale,
                 void* stream);

/* Workspace variants (ABI kept from round 1, when the k-along-rows operands V^T / K^T / Q^T / dO^T were pre-transposed into a caller
 * workspace).  The kernels now read those operands with LDS transpose reads (ds_read_b64_tr_b16) straight out of the row-major tiles:
 * pcm_attn_workspace_bytes returns 0 for every shape and the *_ws calls ignore `workspace`; they are exactly pcm_attn_fwd / pcm_attn_bwd. */
size_t pcm_attn_workspace_bytes(int B, int H, int Lq, int Lk, int d, int backward);
int pcm_attn_fwd_ws(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk, int d, int ldq,
                    int ldk, int ldo, float scale, void* workspace, size_t workspace_bytes, void* stream);
int pcm_attn_bwd_ws(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse, float* delta, void* dq,
                    void* dk, void* dv, int B, int H, int Lq, int Lk, int d, int ldq, int ldk, int ldo, float scale, void* workspace,
                    size_t workspace_bytes, void* stream);

/* The same attention with a PRE-SCALED query (abi >= 4): q' = q * (d^-1/2 * log2 e), i.e. the caller folds the softmax scale AND the
 * base change into the to_q projection (its packed weights and its LoRA factor s*B; pcm_amd/model.py), so the scores leave the QK^T MFMA
 * in the log2 domain and the kernels spend no VALU work on scaling: for head dims with spare contraction slots (40) the reference
 * subtraction rides the MFMA too, and the forward tracks no running maximum after the first key tile (csrc/attention_ps.hip).
 * Same layouts and strides as pcm_attn_fwd / pcm_attn_bwd; no `scale` argument.  lse is in the log2 domain as before.
 * pcm_attn_bwd_prescaled returns dq' = dL/dq' (the gradient with respect to the PRE-SCALED query: back-propagating it through the
 * scaled to_q weights gives dL/dx directly); dk / dv as before (both NULL: only dq').  `o` must be given (delta is computed in the call). */
int pcm_attn_fwd_prescaled(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk, int d, int ldq,
                           int ldk, int ldo, void* stream);
int pcm_attn_bwd_prescaled(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse, float* delta,
                           void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int d, int ldq, int ldk, int ldo, void* stream);

/* ---- small data-movement ops of the UNet wiring (discriminator_sd15.py:312-342) ---------- */
int pcm_upsample2x_nhwc(const void* x, void* y, int B, int H, int W, int C, void* stream);
int pcm_pool2x_sum_nhwc(const void* dy, void* dx, int B, int H, int W, int C, void* stream); /* bwd of upsample: dx[H][W] from dy[2H][2W] */
int pcm_concat_channels(const void* a, int Ca, const void* b, int Cb, void* out, long rows, void* stream);
int pcm_split_channels(const void* in, void* a, int Ca, void* b, int Cb, long rows, int accumulate_a, void* stream);
int pcm_add_bf16(const void* a, const void* b, void* out, long n, void* stream);
int pcm_colsum_bf16(const void* x, void* out /*fp32 [B][C], zeroed by the call*/, int B, int HW, int C, void* stream); /* d(time_emb_proj out) */
/* reproducible forms (abi >= 4): the cross-block fp32 / fp64 atomics of the call above (summation order changes run to run) become
 * per-block partials in `workspace` + one ordered finalize launch -> bitwise identical results run to run. */
size_t pcm_colsum_workspace_bytes(int B, int HW, int C);
int pcm_colsum_bf16_ws(const void* x, void* out, int B, int HW, int C, void* workspace, size_t workspace_bytes, void* stream);
int pcm_silu_bf16(const void* x, void* y, long n, void* stream);
/* dx = dy * silu'(x): backward of the SiLU on the conditioning vector (silu(temb), and between the two embedder linears) when the
 * SD3 adversarial trainers' LoRA list adapts the conditioning path (train_pcm_lora_sd3_adv.py:992-1015) */
int pcm_silu_bwd_bf16(const void* x, const void* dy, void* dx, long n, void* stream);

/* conv_in (4->C0, NCHW fp32 latent in, NHWC bf16 out) and conv_out (C0->4, NHWC bf16 in, NCHW fp32
 * out) + its input gradient — UNet2DConditionModel.conv_in / conv_out, not LoRA targets. */
int pcm_conv_in_fwd(const float* x_nchw, const float* w /*[C0][4][3][3]*/, const float* bias, void* y,
                    int B, int H, int W, int C0, void* stream);
/* abi >= 5: the same with a second copy of the output rows at y2[(b*H + y)*W + x][ld2] (NULL: none) -- conv_in's output is the first skip tensor */
int pcm_conv_in_fwd2(const float* x, const float* w, const float* bias, void* y, void* y2, int ld2, int B, int H, int W, int C0, void* stream);
int pcm_conv_out_fwd(const void* x, const float* w /*[4][C0][3][3]*/, const float* bias, float* y_nchw,
                     int B, int H, int W, int C0, void* stream);
int pcm_conv_out_bwd(const float* dy_nchw, const float* w, void* dx, int B, int H, int W, int C0, void* stream);

/* conv1x1 to ONE channel (DiscriminatorHead.conv_out, discriminator_sd15.py:362): out[m] = x[m,:].w + b (fp32);
 * backward: dx = dy w (bf16, optional), dw += sum_m dy x, db += sum dy (fp32, accumulated) */
int pcm_rowdot_fwd(const void* x, const float* w, const float* bias, float* out, long M, int C, void* stream);
int pcm_rowdot_bwd(const void* x, const float* w, const float* dy, void* dx, float* dw, float* db, long M, int C, void* stream);
/* abi >= 5, reproducible form: per-workgroup partials in the caller's workspace (>= pcm_rowdot_bwd_workspace_bytes) + an ordered finalize that
 * adds into dw / db like the atomic form: bitwise identical run to run */
size_t pcm_rowdot_bwd_workspace_bytes(long M, int C);
int pcm_rowdot_bwd_ws(const void* x, const float* w, const float* dy, void* dx, float* dw, float* db, long M, int C, void* workspace,
                      size_t workspace_bytes, void* stream);

/* sinusoidal timestep projection (diffusers Timesteps(320, flip_sin_to_cos=True, shift=0)) */
int pcm_timestep_embedding(const int64_t* t, void* out /*bf16 [B][dim]*/, int B, int dim, void* stream);

/* ---- phased-consistency math on latents, NCHW [B][4][H][W] (reference-owned) -------------
 * The reference's DDIM tables ddim_alpha_cumprods_prev are float64 (np.asarray over python floats,
 * train_pcm_lora_sd15.py:297-299), so its jump / x_prev / target tensors are float64 and are cast
 * with .float() only at the UNet input and in the loss (:1264, :1285-1290).  These kernels keep
 * that: fp32 where the reference is fp32 (no FMA contraction), fp64 where it is fp64. */
/* add_noise: scheduling_ddpm_modified.py:500-524 (fp32) */
int pcm_add_noise(const float* x, const float* noise, const float* alphas_cumprod, const int64_t* t,
                  float* out, int B, int per_sample, void* stream);
/* predicted_origin(epsilon) + ddim_style_multiphase_pred + boundary blend
 * (train_pcm_lora_sd15.py:268-280, :321-341, :1212/:1280):
 *   x0 = (sample - sigma_t*eps)/alpha_t ; e = largest edge <= index ;
 *   jump = sqrt(acp_prev[e])*x0 + sqrt(1-acp_prev[e])*eps ; out = c_skip ? sample : jump
 * sample is fp32 (online: noisy_model_input) or fp64 (target: x_prev) per sample_f64.
 * target_mode=0: online (c_skip=0) ; 1: target (c_skip = index in edges).  out is fp32 (= .float()).
 * Also writes coef[b] = d out / d eps (backward of the online branch) and
 * end_t[b] = ddim_timesteps_prev[e].  acp_prev is the fp64 table. */
int pcm_phase_jump(const float* eps, const void* sample, int sample_f64, const int64_t* t,
                   const int64_t* index, const float* alphas_cumprod, const double* acp_prev,
                   const int64_t* t_prev, const int64_t* edges, int n_edges, int target_mode,
                   float* out, float* coef, int64_t* end_t, int B, int per_sample, void* stream);
/* CFG-augmented DDIM step (train_pcm_lora_sd15.py:1224-1258): x_prev in fp64 (+ fp32 copy) */
int pcm_cfg_ddim_step(const float* eps_c, const float* eps_u, const float* sample, const int64_t* t,
                      const int64_t* index, const float* w, const float* alphas_cumprod,
                      const double* acp_prev, double* x_prev, float* x_prev_f32, int B,
                      int per_sample, void* stream);

/* Inference sampler step (log_validation, train_pcm_lora_sd15.py:120-145: StableDiffusionPipeline denoising loop with
 * DDIMScheduler(timestep_spacing="trailing", clip_sample=False, set_alpha_to_one=False), eta = 0): classifier-free guidance
 * combine (eps_u may be NULL: guidance_scale <= 1) + x_{t-1} from x_t.  alpha_* are alphas_cumprod values; fp32 throughout. */
int pcm_sampler_ddim_step(const float* eps_c, const float* eps_u, const float* x, float alpha_t, float alpha_prev, float guidance,
                          float* out, long n, void* stream);
/* ---- MMDiT block pieces of the SD3 variant.  The transformer is diffusers' SD3Transformer2DModel (un-vendored
 * dependency, pinned by code/text_to_image_sd3/environment.sd3.yaml); its call order is witnessed in-repo by the copied forward at
 * code/text_to_image_sd3/discriminator_sd3.py:73-137 (pos_embed -> time_text_embed -> context_embedder -> JointTransformerBlocks ->
 * norm_out -> proj_out -> unpatchify). ---------------------------------------------------------------------------------------- */
/* AdaLayerNormZero / AdaLayerNormContinuous: LayerNorm without affine, then y = xhat * gamma[b] + beta[b] (gamma = 1 + scale, beta = shift,
 * fp32 [B][C]; rows_per_batch consecutive rows share one pair).  bwd: input gradient only (+ optional accumulate of dres). */
int pcm_layernorm_mod_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int C,
                          float eps, int rows_per_batch, void* stream);
int pcm_layernorm_mod_bwd(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, const void* dres,
                          void* dx, int M, int C, int rows_per_batch, void* stream);
/* gated residual of the adaLN-Zero block: out = (res ? res : 0) + gate[b] * y   (bf16 [M][C], gate fp32 [B][C]) */
int pcm_rowgate_fma(const void* y, const float* gate, const void* res, void* out, int M, int C, int rows_per_batch, void* stream);
/* FeedForward(activation_fn="gelu-approximate") */
int pcm_gelu_tanh_fwd(const void* x, void* y, long n, void* stream);
int pcm_gelu_tanh_bwd(const void* x, const void* dy, void* dx, long n, void* stream);
/* PatchEmbed's Conv2d(k=2, s=2) as a GEMM: fp32 NCHW image -> bf16 token rows [B*(H/2)*(W/2)][4C], column order 0 = (c,p,q) (conv weight),
 * 1 = (p,q,c) (the unpatchify einsum "nhwpqc->nchpwq", discriminator_sd3.py:112-131); and the inverse (fp32 tokens -> fp32 image). */
int pcm_patchify2x2(const float* img, void* tokens, int B, int C, int H, int W, int order, void* stream);
int pcm_unpatchify2x2(const float* tokens, float* img, int B, int C, int H, int W, int order, void* stream);
/* gradients of the per-sample modulation vectors (adaLN projections with LoRA, train_pcm_lora_sd3_adv.py:992-1015): per sample b,
 * out_a[b][c] = sum_l dy * u with u = (x - mean[row]) * rstd[row] (mean != NULL: d(1+scale)) or u = x (mean == NULL: d gate);
 * out_b[b][c] = sum_l dy (d shift; optional).  x, dy bf16 [B*L][C]; outputs fp32 [B][C], zeroed by the call. */
int pcm_mod_grad(const void* x, const void* dy, const float* mean, const float* rstd, float* out_a, float* out_b, int B, int L, int C,
                 void* stream);
/* abi >= 5, reproducible form of the call above (workspace >= pcm_mod_grad_workspace_bytes, 16-byte aligned) */
size_t pcm_mod_grad_workspace_bytes(int B, int L, int C);
int pcm_mod_grad_ws(const void* x, const void* dy, const float* mean, const float* rstd, float* out_a, float* out_b, int B, int L, int C,
                    void* workspace, size_t workspace_bytes, void* stream);
/* sinusoidal projection of FLOAT timesteps (sigma * 1000, train_pcm_lora_sd3.py:1295-1300), flip_sin_to_cos, shift 0 -> bf16 [B][dim] */
int pcm_timestep_embedding_f32(const float* t, void* out, int B, int dim, void* stream);

/* ---- flow-matching PCM math of the SD3 variant (paths under code/text_to_image_sd3/) ----------------------
 * Tables as the reference's EulerSolver builds them (train_pcm_lora_sd3.py:158-175): sigmas float32 [E], sigmas_prev FLOAT64 [E]
 * (np.asarray over python floats), so every result that touches sigma_prev is float64, like the reference's. */
/* train_pcm_lora_sd3.py:1291,:1301   noisy = sigmas[index] * noise + (1 - sigmas[index]) * x   (float32) */
int pcm_fm_add_noise(const float* x, const float* noise, const float* sigmas, const int64_t* index, float* out, int B,
                     int per_sample, void* stream);
/* EulerSolver.euler_style_multiphase_pred (:192-230): edges = floor(linspace(0, E, multiphase, endpoint=False)) on the device;
 * end = last edge <= index[b]; out = sample + (sigmas_prev[end] - (target_mode ? sigmas_prev[index] : sigmas[index])) * model_pred
 * (float64; out_f32 optional copy; end_index optional [B]).  sample is float32 (noisy input, :1313) or float64 (x_prev, :1368). */
int pcm_fm_phase_jump(const void* sample, int sample_f64, const float* model_pred, const int64_t* index, const float* sigmas,
                      const double* sigmas_prev, const int64_t* edges, int n_edges, int target_mode, double* out,
                      float* out_f32, int64_t* end_index, int B, int per_sample, void* stream);
/* teacher CFG with the fixed w (:1334, :1352-1354; uncond may be NULL: --not_apply_cfg_solver) + EulerSolver.euler_step (:184-190):
 * x_prev = sample + (sigmas_prev[index] - sigmas[index]) * (cond + w * (cond - uncond))   (float64 + optional float32 copy) */
int pcm_fm_cfg_euler_step(const float* cond, const float* uncond, const float* sample, const int64_t* index, float w,
                          const float* sigmas, const double* sigmas_prev, double* x_prev, float* x_prev_f32, int B,
                          int per_sample, void* stream);
/* adversarial trainers, train_pcm_lora_sd3_adv.py:1413-1445: x_adv = ((1 - s_adv) * x + (s_adv - s_end) * noise) / (1 - s_end) with
 * s_* = sigmas_prev[end_index | adv_index]; float64 in (model_pred / target and the reference's randn_like are float64), float64 and / or
 * float32 out; ratio[b] = (1 - s_adv) / (1 - s_end) (optional, the generator step's chain-rule factor). */
int pcm_fm_noise_travel(const double* x, const double* noise, const double* sigmas_prev, const int64_t* end_index,
                        const int64_t* adv_index, double* out, float* out_f32, float* ratio, int B, int per_sample, void* stream);
/* Inference: one step of PCMFMDeterministicScheduler.step (pcm_fm_deterministic_scheduler.py:225-233; noise == NULL) or
 * PCMFMStochasticScheduler.step (pcm_fm_stochastic_scheduler.py:225-233; noise = the step's randn_like draw), float32, optionally
 * fused with the pipeline's guidance combine v = v_u + guidance * (v_c - v_u) (model_output_uncond may be NULL). */
int pcm_fm_sampler_step(const float* model_output, const float* model_output_uncond, float guidance, const float* sample, float sigma,
                        float sigma_next, const float* noise, float* out, long n, void* stream);

/* loss (l2 | huber, :1283-1293) forward + gradient wrt the student's eps prediction:
 * loss[0] = mean(...) (accumulated in fp64, zeroed by the call) ; d_eps = dloss/dmodel_pred * coef[b] * grad_scale */
int pcm_consistency_loss(const float* model_pred, const float* target, const float* coef, int huber,
                         float huber_c, double* loss, float* d_eps, float grad_scale, int B,
                         int per_sample, void* stream);
/* the same with per-block fp64 partials + an ordered finalize instead of one fp64 atomic per block (workspace >= PCM_REDUCE_WS_BYTES) */
#define PCM_REDUCE_WS_BYTES 32768
int pcm_consistency_loss_ws(const float* model_pred, const float* target, const float* coef, int huber, float huber_c, double* loss,
                            float* d_eps, float grad_scale, int B, int per_sample, void* workspace, size_t workspace_bytes, void* stream);

/* noise_travel: scheduling_ddpm_modified.py:526-554 (fp32); sqrt_r[b] = d out / d x (optional) */
int pcm_noise_travel(const float* x, const float* noise, const float* alphas_cumprod, const int64_t* t_cur,
                     const int64_t* t_tgt, float* out, float* sqrt_r, int B, int per_sample, void* stream);
/* hinge losses of the latent discriminator on one head's logit map (discriminator_sd15.py:412-434):
 * mode 0 (d_loss): loss += scale*(mean relu(f+1) + mean relu(1-r)); mode 1 (g_loss): loss += scale*mean relu(1-f);
 * d_fake / d_real (optional) = d loss / d logit * grad_scale.  `loss` accumulates over heads. */
int pcm_hinge_loss(const float* fake, const float* real, int mode, float scale, double* loss, float* d_fake,
                   float* d_real, float grad_scale, long n, void* stream);
/* abi >= 5, reproducible form: one workgroup per call, so the sum a call adds to `loss` has a fixed order (the per-head logit maps are small) */
int pcm_hinge_loss_ordered(const float* fake, const float* real, int mode, float scale, double* loss, float* d_fake, float* d_real,
                           float grad_scale, long n, void* stream);

/* out[b][:] += x[b][:] * s1[b] * s2[b]  (generator step: d fake_adv -> d eps through noise_travel and the phase jump) */
int pcm_scale_add_rows(float* out, const float* x, const float* s1, const float* s2, int B, int per_sample, void* stream);

/* ---- optimizer (torch.optim.AdamW + clip_grad_norm_, train_pcm_lora_sd15.py:1297-1301) ---- */
int pcm_sumsq_f32(const float* g, double* out /*1, zeroed by the call*/, long n, void* stream);
int pcm_sumsq_f32_ws(const float* g, double* out, long n, void* workspace /* >= PCM_REDUCE_WS_BYTES */, size_t workspace_bytes, void* stream);
int pcm_adamw_clip_step(float* p, const float* g, float* m, float* v, const double* gradsq,
                        float max_norm, float lr, float beta1, float beta2, float eps, float wd,
                        int step, float grad_scale, long n,
                        const int64_t* step_dev /*NULL: use `step`*/, const float* lr_dev /*NULL: use `lr`*/,
                        void* stream);
/* Loss scaling for the half build (accelerate's GradScaler around train_pcm_lora_sd15.py:1296-1299 when --mixed_precision=fp16):
 * pcm_scale_f32_dev multiplies the loss gradient by the scale S held in DEVICE memory before the backward; the *_scaled optimizer step
 * divides it out again (g' = g * grad_scale / S, clip on the unscaled norm) and does NOTHING when *gradsq is not finite;
 * pcm_loss_scale_update then applies GradScaler.update(): S *= growth after `interval` consecutive finite steps, S *= backoff (and
 * *step_dev -= 1: a skipped update is not an optimizer step) after a non-finite one.  Everything stays capturable in a hipGraph. */
int pcm_scale_f32_dev(float* x, const float* scale_dev, long n, void* stream);
int pcm_adamw_clip_step_scaled(float* p, const float* g, float* m, float* v, const double* gradsq,
                               float max_norm, float lr, float beta1, float beta2, float eps, float wd,
                               float grad_scale, long n, const int64_t* step_dev, const float* lr_dev /*NULL: use `lr`*/,
                               const float* loss_scale_dev, void* stream);
int pcm_loss_scale_update(float* scale, int* good_steps, int64_t* step_dev /*may be NULL*/, const double* gradsq,
                          float growth, float backoff, int interval, void* stream);
/* update_ema (train_pcm_lora_sd15.py:344-355; defined by the reference, never called) */
int pcm_ema_update(float* target, const float* source, float rate, long n, void* stream);
/* the same, skipped when *gradsq (the squared global gradient norm of the optimizer step it follows) is not finite: with loss-scaled half
 * gradients pcm_adamw_clip_step_scaled skips such a step (torch.cuda.amp.GradScaler.step), and the EMA must not move either (abi >= 4) */
int pcm_ema_update_gated(float* target, const float* source, float rate, long n, const double* gradsq, void* stream);

/* ---- operand packing (fp32 master -> bf16 MFMA operand layouts) -------------------------- */
/* linear weight [N][K] fp32 -> bf16 [N][K] (scaled) and/or transposed bf16 [K][N] */
int pcm_pack_linear(const float* w, void* w_nk, void* w_kn, int N, int K, float scale, void* stream);
/* conv weight [N][C][3][3] fp32 -> fwd operand bf16 [N][(kh,kw,c)] and/or dgrad operand
 * bf16 [C][(kh',kw',n)] with the taps flipped */
/* src_khwc = 0: w is [N][C][3][3] (torch / peft);  1: w is [N][3][3][C] (this library's internal layout of
 * the LoRA conv-A factors, chosen so their weight-gradient atomics are contiguous) */
int pcm_pack_conv3x3(const float* w, void* w_fwd, void* w_dgrad, int N, int C, float scale, int src_khwc, void* stream);
/* all LoRA operand copies in one launch: desc = strided fp32 matrix src[R][Cc] in the flat parameter buffer
 * -> bf16 copy and/or transpose at element offsets of the flat operand buffer (offset < 0: skip).
 * descs / blk_start (prefix of 32x32-tile counts, ndesc+1 entries) live in DEVICE memory. */
typedef struct {
  long src_off, dst_copy_off, dst_t_off;
  int R, Cc, lds, ldc, ldt;
  float scale;
} pcm_pack_desc;
int pcm_pack_segmented(const float* src_base, void* dst_base, const pcm_pack_desc* descs, const int* blk_start,
                       int ndesc, int total_blocks, void* stream);
int pcm_cast_f32_bf16(const float* x, void* y, long n, void* stream);
int pcm_cast_bf16_f32(const void* x, float* y, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
