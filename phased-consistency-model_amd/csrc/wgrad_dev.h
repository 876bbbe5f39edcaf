// Device-side argument block of the LoRA weight-gradient kernels (wgrad.hip: register-transposing kernel for every geometry;
// wgrad_tr.hip: LDS-DMA + transpose-read kernels for the plain and the stride-1 3x3 views).
#pragma once
#include "pcm_common.h"

struct WgDev {
  const bf16_t* big; int ldb, G, mode, Hs, Ws, C, stride, src_mode, Ho, Wo;
  const bf16_t* small_; int lds_, M;
  float* out; long g_stride, r_stride; int out_conv; float alpha; int m_per_block; int swap;
  float* part;     // reproducible form (pcm_hip.h): slab base, block `by` of the M split stores its tile to part[by][g][r] (r fastest, 64 wide)
};
// one output element of a block's tile: fp32 atomic into `out`, or a plain store into the block's slab (an ordered finalize adds the slabs)
template <typename WG>
__device__ __forceinline__ void wg_emit(const WG& a, size_t off, int by, int g, int r, float v) {
  if (a.part) a.part[((size_t)by * a.G + g) * 64 + r] = v;
  else atomicAdd(a.out + off, v);
}


#define PCM_WGRAD_MULTI_MAX 8
// several plain-view jobs in one launch (wgrad_tr.hip); taken[i] = 1 for the jobs it launched, the caller runs the others one by one
int pcm_wgrad_tr_launch_multi(const WgDev* jobs, int n, unsigned char* taken, void* stream);
// returns 0 when a transpose-read kernel took the call, 1 when the geometry is not one of theirs (caller falls back), < 0 on error
int pcm_wgrad_tr_launch(const WgDev& a, void* stream, int* msplit_out = nullptr, bool plan_only = false);
