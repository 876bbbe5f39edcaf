// Error reporting for the C ABI (thread-local message, no exceptions across the boundary).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "pcm_common.h"

static thread_local char g_err[512] = "";

void pcm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int pcm_post_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pcm_set_error("%s: HIP launch error: %s", what, hipGetErrorString(e));
    return PCM_EHIP;
  }
  return PCM_OK;
}

// Zero fill as a KERNEL.  hipMemsetAsync inside a captured hipGraph becomes a memset node; on this stack (ROCm 7.2) replays of a graph with
// such nodes in front of atomic accumulations gave wrong sums from the second replay on (tests/test_gpu_bench_config.py: graph vs eager),
// so every "zero, then accumulate" sequence of the library clears with this launch instead (bytes must be a multiple of 4).
__global__ __launch_bounds__(256) void pcm_zero_kernel(unsigned* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0u;
}
void pcm_zero_async(void* p, size_t bytes, void* stream) {
  const long n = (long)(bytes / 4);
  long b = (n + 255) / 256; if (b > PCM_GRID_CAP(1024)) b = PCM_GRID_CAP(1024); if (b < 1) b = 1;
  PCM_LAUNCH(pcm_zero_kernel, dim3((unsigned)b), dim3(256), 0, stream, (unsigned*)p, n);
}

// Ordered reduction of per-workgroup partial sums (the reproducible forms of the cross-workgroup reductions, include/pcm_hip.h abi 4 / 5):
//   out[i] (+)= sum_{p = 0 .. nparts-1, in this order} part[p * stride + i],   i in [0, n)
// the summation order is fixed by the partial index, not by which workgroup finished first.
__global__ __launch_bounds__(256) void pcm_partials_finalize_kernel(const float* part, long stride, float* out, int nparts, long n, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int p = 0; p < nparts; p++) t += part[(size_t)p * stride + i];
    out[i] = accumulate ? out[i] + t : t;
  }
}
void pcm_partials_finalize(const float* part, long stride, float* out, int nparts, long n, int accumulate, void* stream) {
  long b = (n + 255) / 256; if (b > PCM_GRID_CAP(1024)) b = PCM_GRID_CAP(1024); if (b < 1) b = 1;
  PCM_LAUNCH(pcm_partials_finalize_kernel, dim3((unsigned)b), dim3(256), 0, stream, part, stride, out, nparts, n, accumulate);
}

extern "C" const char* pcm_last_error(void) { return g_err; }
// identity of the sources this library was built from: the first 16 hex digits of sha256 over csrc/*.hip, csrc/*.h and include/pcm_hip.h
// + "-" + the build variant (pcm_amd/build.py compiles it into this object); "unknown" for a build that did not go through build.py
#ifndef PCM_BUILD_ID
#define PCM_BUILD_ID "unknown"
#endif
extern "C" const char* pcm_build_id(void) { return PCM_BUILD_ID; }
extern "C" int pcm_abi_version(void) { return 5; }   // 2: pcm_gemm_epi gained pre_out / pre_rows / ldp (PCM_ACT_GEGLU second output); 3: PCM_ACT_GEGLU rows interleaved in groups of 2 (was 8); 4: pcm_wgrad_args gained workspace / workspace_bytes, the *_ws reproducible reductions; 5: pcm_gemm_epi gained out2 / ldo2 (second output copy: concat-free skips) and chstats / stats_rows (GroupNorm statistics from the producing epilogue), pcm_gemm_emits_chstats, pcm_groupnorm_apply_chstats, pcm_build_id, the reproducible forms pcm_rowdot_bwd_ws / pcm_groupnorm_param_grad_ws / pcm_mod_grad_ws / pcm_hinge_loss_ordered
// the 16-bit activation / weight format this library was compiled for (pcm_common.h): 0 = bfloat16 (libpcm_hip.so), 1 = IEEE half
// (libpcm_hip_f16.so, -DPCM_ACT_F16).  The host side checks it against the tensors it is about to pass.
#ifdef PCM_ACT_F16
extern "C" int pcm_act_dtype(void) { return 1; }
#else
extern "C" int pcm_act_dtype(void) { return 0; }
#endif
