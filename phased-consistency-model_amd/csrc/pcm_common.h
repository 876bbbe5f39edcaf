// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the PCM distillation hot path.
// wave = 64 lanes; bf16 is carried as raw uint16 bits; fp32 accumulation everywhere.
#pragma once
#ifdef PCM_HOST_EMU  // set ONLY by tests/emu/build_emu.py (host index-math checks); never in the product build
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define PCM_LAUNCH(kern, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kern, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__)
#define PCM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define PCM_AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define PCM_AS3(p) ((__attribute__((address_space(3))) void*)(p))
#define PCM_EXPF(x) __expf(x)
#define PCM_EXP2F(x) __builtin_amdgcn_exp2f(x)
// counted waits for hand-pipelined LDS-DMA loops (hipcc never emits these for LDS-DMA -> ds_read dependences)
#define PCM_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define PCM_WAIT_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// 16-byte-per-lane LDS-DMA with lanes masked off: an inactive lane transfers nothing and its LDS slot keeps its contents (EXEC mask)
#define PCM_DMA16_MASKED(rs, lds, voff, soff, active)                                                   \
  do {                                                                                                  \
    if (active) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, PCM_AS3(lds), 16, voff, soff, 0, 0);       \
  } while (0)
// code-generation controls with no meaning off the hardware (the emulator header defines the same names as no-ops, so that kernel files
// never test PCM_HOST_EMU themselves): PCM_HW_ONLY(statements) = scheduler hints / register-class requests / hardware-register reads;
// PCM_PIN_V / PCM_PIN_S = opaque copy of a vector / scalar value (keeps hipcc from hoisting what depends on it);
// PCM_KERNARG_REF(T, arr, i) = element i of an array that is the FIRST member of the kernel's argument struct, read from the kernarg
// segment by scalar loads (no per-job copy of the argument block in registers)
// a wave's LDS operations execute in issue order: lanes of ONE wave may hand data to each other through LDS without a workgroup barrier,
// provided the compiler keeps the order (this is a scheduling fence, no instruction; the emulator's fibers rendezvous here)
#define PCM_WAVE_LDS_FENCE() __builtin_amdgcn_wave_barrier()
// LDS accesses hipcc does not track (inline asm; the emulator's are plain accesses).  Why: while LDS-DMA pieces are in flight hipcc puts
// s_waitcnt vmcnt(0) in front of EVERY ds_read / ds_write it emits itself (it cannot tell which LDS bytes the DMA will write), which drains a
// prefetched tile the moment an epilogue touches its staging area.  These are for regions the code itself has ordered against the DMA (a counted
// PCM_WAIT_VMCNT).  Each load is ONE asm statement with its own s_waitcnt: with the wait in a second statement hipcc may copy the destination
// registers between the two (a v_mov of a register the LDS has not written yet -- seen on hardware as a wrong upper dword, never on the emulator).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define PCM_LDS_LD64(dst, p) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dst) : "v"((unsigned)(size_t)(p)) : "memory")
#define PCM_LDS_LD128(dst, p) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dst) : "v"((unsigned)(size_t)(p)) : "memory")
#define PCM_LDS_LD128_LD64(d128, p128, d64, p64)                                                                             \
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d128), "=&v"(d64)               \
               : "v"((unsigned)(size_t)(p128)), "v"((unsigned)(size_t)(p64)) : "memory")
#define PCM_LDS_ST64(p, val) asm volatile("ds_write_b64 %0, %1" : : "v"((unsigned)(size_t)(p)), "v"(val) : "memory")
#define PCM_LDS_WAIT_ALL() asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory")
#define PCM_HW_ONLY(...) __VA_ARGS__
#define PCM_PIN_V(x) asm volatile("" : "+v"(x))
#define PCM_PIN_S(x) asm volatile("" : "+s"(x))
#define PCM_KERNARG_REF(T, arr, i) (((const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr())[i])
#endif
// HIP compiles a kernel file twice; the host pass only needs the launch stubs (device builtins do not exist there): kernel bodies are
// wrapped in #if PCM_KERNEL_BODY
#if defined(__HIP_DEVICE_COMPILE__) || defined(PCM_HOST_EMU)
#define PCM_KERNEL_BODY 1
#else
#define PCM_KERNEL_BODY 0
#endif
#include <stdint.h>
#include <string.h>

#include <type_traits>
// grid-stride kernels: cap of the launch grid (the host emulator runs blocks serially, keep it small there)
#ifdef PCM_HOST_EMU
#define PCM_GRID_CAP(n) 4
#else
#define PCM_GRID_CAP(n) (n)
#endif

#include "../../include/pcm_hip.h"

// ---- process-wide tuning / test knobs exist in the TOOLS build only -------------------------------------------------------------------
// The product library (libpcm_hip.so, libpcm_hip_f16.so) carries NO mutable global state besides once-only hipFuncSetAttribute flags: every
// A/B switch, tile-forcing hook, launch counter and environment variable of the development rounds is compiled to its shipped constant, the
// pcm_debug_* entry points are not exported and the kernel variants only they can reach are not instantiated.  -DPCM_TOOLS
// (pcm_amd/build.py variant "tools" -> lib/libpcm_hip_tools.so, what tools/ and the hook-using tests load) and the host emulator build keep them.
#if defined(PCM_TOOLS) || defined(PCM_HOST_EMU)
#define PCM_HAS_TOOLS 1
#include <stdlib.h>
#define PCM_KNOB static                     /* PCM_KNOB int g_x = 0;  -- a variable with a pcm_debug_* setter */
#define PCM_TOOLS_ONLY(...) __VA_ARGS__
static inline int pcm_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// a knob whose -1 means "the environment switch `env` if set, else `dflt`", read once:  PCM_LAZY_KNOB(big_mode, g_big_mode, "PCM_GEMM_BIG", 1)
#define PCM_LAZY_KNOB(fn, var, env, dflt)  \
  static int var = -1;                     \
  static int fn() {                        \
    if (var < 0) var = pcm_env_int(env, dflt); \
    return var;                            \
  }
#else
#define PCM_HAS_TOOLS 0
#define PCM_KNOB static constexpr
#define PCM_TOOLS_ONLY(...)
#define PCM_LAZY_KNOB(fn, var, env, dflt) static constexpr int fn() { return dflt; }
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) -- indices usable as asm immediates / register-array subscripts
template <int I, int N, typename F>
__device__ __forceinline__ void pcm_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); pcm_static_for<I + 1, N>(f); }
}
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16 (gfx950 LDS transpose read).  Lane mapping measured on MI355X (tools/probes/trread.hip): inside each 16-lane
// group, result lane i receives element (i & 3) of the 8-byte row addressed by SOURCE lane 4*j + (i >> 2), for j = 0..3.  I.e. source
// lanes 4j..4j+3 address the four 4-element quads of row j of a 4 x 16 block and lane i gets column i: the k-strided operand of an
// MFMA (8 consecutive k for one row / column) comes out of a row-major LDS tile with two of these reads.  Addresses must be 8-B aligned.
#ifdef PCM_HOST_EMU
#define PCM_DS_READ_TR16(p) pcm_emu::ds_read_tr16_b64((const void*)(p))
#else
#define PCM_DS_READ_TR16(p) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(p))
#endif
// The builtin form above is tracked by hipcc like any LDS read -- including against pending LDS-DMA writes: in a loop that keeps
// buffer_load...lds pieces in flight across its barriers hipcc puts s_waitcnt vmcnt(0) in front of the first builtin read and drains the
// prefetch (measured on wgrad_tr.hip).  PCM_TR16_ISSUE is the same instruction as inline asm (guide section 5.7): hipcc neither counts
// nor waits for it, so (1) it can be issued EARLY (e.g. the V^T fragments of an attention tile before the softmax) and (2) it does not
// drain the DMA queue; the destination is valid only after PCM_TR16_WAITn(...) naming every destination (data dependence for the
// consumers).  ``imm`` must be a compile-time constant < 65536 (pcm_static_for gives loop indices as constants).
#ifdef PCM_HOST_EMU
#define PCM_TR16_ISSUE(dst, base, imm) (dst) = pcm_emu::ds_read_tr16_b64((const char*)(base) + (imm))
#define PCM_TR16_WAIT8(a, b, c, d, e, f, g, h) ((void)0)
#define PCM_TR16_KEEP8(a, b, c, d, e, f, g, h) ((void)0)
#define PCM_TR16_WAIT4(a, b, c, d) ((void)0)
#define PCM_TR16_WAIT2(a, b) ((void)0)
#else
#define PCM_TR16_ISSUE(dst, base, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"((unsigned)(size_t)(base)), "n"(imm))
#define PCM_TR16_WAIT8(a, b, c, d, e, f, g, h) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h))
#define PCM_TR16_KEEP8(a, b, c, d, e, f, g, h) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h))
#define PCM_TR16_WAIT4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define PCM_TR16_WAIT2(a, b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b))
#endif
__device__ __forceinline__ bf16x8 pcm_join4(bf16x4 a, bf16x4 b) { return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// two fp32 values per VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: 4.5 SIMD cycles for two results against 3.2 for one
// v_fma_f32, tools/probes/valubench.hip) -- element-wise operators on this type select the packed instructions
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pcm_pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
typedef unsigned short bf16_t;

// thread-local last-error string (C-ABI: never throw across the boundary)
void pcm_set_error(const char* fmt, ...);
#define PCM_CHECK(cond, code, ...)      \
  do {                                  \
    if (!(cond)) {                      \
      pcm_set_error(__VA_ARGS__);       \
      return (code);                    \
    }                                   \
  } while (0)
#define PCM_ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)
int pcm_post_launch(const char* what);
void pcm_partials_finalize(const float* part, long stride, float* out, int nparts, long n, int accumulate, void* stream);   // runtime.hip: ordered sum of per-workgroup partials
void pcm_zero_async(void* p, size_t bytes, void* stream);   // zero fill as a kernel launch (runtime.hip: why not hipMemsetAsync)

// ---- the 16-bit activation / weight format.  Default build: bfloat16 (libpcm_hip.so).  -DPCM_ACT_F16 (pcm_amd/build.py variant "f16",
// libpcm_hip_f16.so) compiles the SAME sources with IEEE half as the storage and MFMA operand type: the reference's
// --mixed_precision=fp16 recipes (train_pcm_lora_sd15.sh:9) and the validation mode behind DESIGN section 5's 1e-3 loss comparison (half keeps 11
// significand bits against 8).  Everything that touches the format is in this block: bf2f / f2bf / pack_bf2 and the two MFMA builtins (the kernels
// carry operands as raw 16-bit lanes, ``bf16_t`` / ``bf16x8`` are format-agnostic containers; entry points keep their *_bf16 names).
// pcm_act_dtype() (runtime.hip) reports which one a library was built with.
#ifdef PCM_ACT_F16
#define PCM_ONE_BITS 0x3c00u      // 1.0 in the 16-bit format (attention: the "ones" column that makes the PV MFMA produce the softmax denominator)
#ifdef PCM_HOST_EMU
__device__ __forceinline__ float bf2f(bf16_t h) { return pcm_emu::half_to_float(h); }
__device__ __forceinline__ bf16_t f2bf(float f) { return pcm_emu::float_to_half(f); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }
#else
typedef _Float16 pcm_h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 pcm_h8_t __attribute__((ext_vector_type(8)));
typedef float pcm_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {        // round-to-nearest-even (v_cvt_pk_f16_f32), overflow -> inf
  pcm_f2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pcm_h2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
__device__ __forceinline__ f32x16 pcm_mfma_32x32x16_h(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pcm_h8_t, a), __builtin_bit_cast(pcm_h8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 pcm_mfma_16x16x32_h(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pcm_h8_t, a), __builtin_bit_cast(pcm_h8_t, b), c, 0, 0, 0);
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) pcm_mfma_32x32x16_h((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) pcm_mfma_16x16x32_h((a), (b), (c))
#endif
#else   // bfloat16
#define PCM_ONE_BITS 0x3f80u
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
#ifdef PCM_HOST_EMU
__device__ __forceinline__ bf16_t f2bf(float f) {  // round-to-nearest-even, NaN kept quiet
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
#else
// gfx950 has a hardware RNE convert (v_cvt_pk_bf16_f32): one instruction per two values
typedef __bf16 pcm_bf2_t __attribute__((ext_vector_type(2)));
typedef float pcm_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  pcm_f2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pcm_bf2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
#endif
#endif
// Activations are evaluated per element inside HBM-bound kernels (GroupNorm+SiLU touches every activation of the UNet), so their
// VALU cost matters: v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions).  Results are rounded to bf16.
#ifdef PCM_HOST_EMU
#define PCM_RCPF(x) (1.0f / (x))
#else
#define PCM_RCPF(x) __builtin_amdgcn_rcpf(x)
#endif
__device__ __forceinline__ float silu_f(float x) { return x * PCM_RCPF(1.0f + PCM_EXPF(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
  float s = PCM_RCPF(1.0f + PCM_EXPF(-x));
  return s * (1.0f + x * (1.0f - s));
}
// GELU (erf form, F.gelu default) = x * Phi(x).  Phi - 1/2 is evaluated as an odd degree-17 polynomial on |x| <= 4 (0 / 1 beyond:
// Phi(-4) = 3.2e-5): max |Phi error| 3.2e-5, i.e. |gelu error| <= 1.3e-4 absolute everywhere -- 40x below one bf16 rounding of a result of
// that size -- in 15 full-rate VALU operations and NO transcendental.  The Abramowitz-Stegun erf used before (exp + rcp, both quarter rate) made
// value * gelu(gate) of the fused feed-forward epilogue cost as many SIMD cycles as the K = 320 projection's MFMAs (round 4).
__device__ __forceinline__ float pcm_phi_f(float x) {
  const float t = fminf(fmaxf(x, -4.0f), 4.0f), z = t * t;
  float p = 7.804711256e-11f;
  p = p * z - 6.827683748e-09f; p = p * z + 2.666981721e-07f; p = p * z - 6.222018266e-06f; p = p * z + 9.829133151e-05f;
  p = p * z - 1.130966313e-03f; p = p * z + 9.869967510e-03f; p = p * z - 6.640203406e-02f; p = p * z + 3.989198652e-01f;
  const float phi = p * t + 0.5f;
  return x < -4.0f ? 0.0f : (x > 4.0f ? 1.0f : phi);      // tails exact to 3.2e-5: gelu(x) -> 0 / x, no |x| * 3.2e-5 drift for large gates
}
__device__ __forceinline__ float gelu_erf_f(float x) { return x * pcm_phi_f(x); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) {      // Phi(x) + x * phi(x)
  return pcm_phi_f(x) + x * 0.3989422804014327f * PCM_EXPF(-0.5f * x * x);
}
// combine a value with the one held by the lane 32 apart (the two halves of a 32x32 MFMA accumulator column).  v_permlane32_swap is a VALU
// instruction: no LDS round trip and no lgkmcnt wait in the middle of a softmax (ds_bpermute is both).
#ifdef PCM_HOST_EMU
__device__ __forceinline__ float pcm_xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float pcm_xhalf_sum(float v) { return v + __shfl_xor(v, 32); }
#else
__device__ __forceinline__ float pcm_xhalf_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pcm_xhalf_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
#endif
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// 8x8 bf16 transpose in registers: r[j] = row j (8 bf16 as 4 dwords) -> o[c] = column c (8 rows)
__device__ __forceinline__ void transpose8x8_bf16(const uint4 (&r)[8], uint4 (&o)[8]) {
  unsigned p[4][8];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const unsigned a[4] = {r[2 * i].x, r[2 * i].y, r[2 * i].z, r[2 * i].w};
    const unsigned b[4] = {r[2 * i + 1].x, r[2 * i + 1].y, r[2 * i + 1].z, r[2 * i + 1].w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
      p[i][2 * d] = (a[d] & 0xffffu) | (b[d] << 16);
      p[i][2 * d + 1] = (a[d] >> 16) | (b[d] & 0xffff0000u);
    }
  }
#pragma unroll
  for (int c = 0; c < 8; c++) o[c] = make_uint4(p[0][c], p[1][c], p[2][c], p[3][c]);
}
