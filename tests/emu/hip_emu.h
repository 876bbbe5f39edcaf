// TEST INFRASTRUCTURE ONLY — host emulation of the small HIP/gfx950 subset the kernels in
// phased-consistency-model_amd/csrc use, so that their index math (MFMA fragment layouts, LDS
// tiling, im2col masks, reductions) can be checked on the GPU-less build container.
// The shipped library is built by hipcc for gfx950 and never sees this header
// (csrc/pcm_common.h includes it only under -DPCM_HOST_EMU, which only tests/emu/build_emu.py sets).
//
// Model: one block at a time; every thread of the block is a fiber (own stack, hand-rolled x86-64 context switch); __syncthreads and
// wave-collectives (shuffles, MFMA, global_load_lds) are cooperative rendezvous points.
// MFMA fragment layouts follow /opt/skills/guides/cdna_hip_programming.md §3.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
// HIP's vector types carry their natural alignment (float4 / uint4: 16 B, float2 / uint2: 8 B): a misaligned 16-byte lane access is
// a fault (or a split access) on the GPU.  With the same alignment here, -fsanitize=alignment (PCM_EMU_UBSAN=1 in build_emu.py)
// reports every under-aligned vector load / store of a kernel with its source line.
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return {a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return {a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return {a, b}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
// launch-limit model (gfx950): a launch that the hardware would reject is NOT run and leaves hipErrorInvalidValue for hipGetLastError,
// exactly like a real launch -- the C ABI then reports rc = PCM_EHIP "HIP launch error: invalid argument"
#define hipErrorInvalidValue 1
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
namespace pcm_emu { extern hipError_t g_last_error; hipError_t set_max_dyn_lds(const void* fn, int bytes); }
static inline hipError_t hipGetLastError() { hipError_t e = pcm_emu::g_last_error; pcm_emu::g_last_error = 0; return e; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipErrorInvalidValue ? "invalid argument" : "emu"; }
static inline hipError_t hipFuncSetAttribute(const void* fn, int attr, int value) {
  return attr == hipFuncAttributeMaxDynamicSharedMemorySize ? pcm_emu::set_max_dyn_lds(fn, value) : hipErrorInvalidValue;
}
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }

namespace pcm_emu {

enum State { RUNNABLE, WAIT_WAVE, WAIT_BLOCK, DONE };
struct DmaOp { void* dst; const void* src; int size; };   // src == nullptr: zero fill (out-of-range buffer load)
struct Fiber {
  void* sp;        // saved stack pointer while the fiber is switched out
  char* stack;
  State st;
  dim3 tid;
  int lin, wave, lane;
  std::vector<DmaOp> pend;   // LDS-DMA issued but not yet landed (PCM_EMU_LAZY_DMA=1: lands only at s_waitcnt vmcnt / __syncthreads)
};
struct WaveScratch {
  // double-buffered exchange area (see hip_emu.h header comment in launch())
  uint64_t u64[2][64];
  short ab[2][2][64][8];
  const void* gp[2][64];
  void* lp[2][64];
  int arrived, alive, gen;
};

extern std::vector<Fiber> g_fibers;
extern std::vector<WaveScratch> g_waves;
extern void* g_sched_sp;
extern "C" void pcm_ctx_switch(void** save_sp, void* to_sp);
extern Fiber* g_cur;
extern int g_block_arrived, g_block_alive;
extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern std::function<void()> g_body;
extern char* g_dyn_smem;
extern bool g_lazy_dma;

void yield_to_sched();
void wave_sync();
void block_sync();
void launch(const void* fn, dim3 grid, dim3 block, size_t smem, std::function<void()> body);

inline int lane_id() { return g_cur->lane; }
inline WaveScratch& wave() { return g_waves[g_cur->wave]; }

template <typename T>
inline T shfl_idx(T v, int src) {
  static_assert(sizeof(T) <= 8, "");
  WaveScratch& w = wave();
  int b = w.gen & 1;
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  w.u64[b][lane_id()] = bits;
  wave_sync();
  T out;
  memcpy(&out, &w.u64[b][src & 63], sizeof(T));
  return out;
}

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// IEEE half <-> float (the -DPCM_ACT_F16 build of the kernels: csrc/pcm_common.h); round-to-nearest-even, overflow -> inf, subnormals kept
inline float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u, u;
  if (e == 0) {
    if (m == 0) u = sign;
    else { int sh = 0; while (!(m & 1024u)) { m <<= 1; sh++; } u = sign | ((uint32_t)(113 - sh) << 23) | ((m & 1023u) << 13); }
  } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
  else u = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline uint16_t float_to_half(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                  // NaN
  if (a >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);                 // >= 65536 -> inf (65520 .. 65536 rounds to inf below)
  if (a < 0x33000001u) return (uint16_t)sign;                              // <= 2^-25 -> 0 (ties to even)
  int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;
  int shift = e < -14 ? 13 + (-14 - e) : 13;                               // subnormal results shift further
  uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  uint32_t h = e < -14 ? q : (((uint32_t)(e + 15) << 10) + (q - 1024u));   // q carries the implicit bit; a mantissa carry bumps the exponent
  return (uint16_t)(sign | h);
}
#ifdef PCM_ACT_F16
inline float bf2f(short s) { return half_to_float((uint16_t)s); }
#else
inline float bf2f(short s) {
  uint32_t u = ((uint32_t)(uint16_t)s) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
#endif

// D[i][j] += sum_k A[i][k] B[k][j];  A: lane l holds A[l&31][8*(l>>5)+e]; B: B[8*(l>>5)+e][l&31];
// D: lane l reg r -> j = l&31, i = (r&3) + 8*(r>>2) + 4*(l>>5)
inline f32x16_t mfma_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c, int, int, int) {
  WaveScratch& w = wave();
  int bsel = w.gen & 1, l = lane_id();
  for (int e = 0; e < 8; e++) { w.ab[bsel][0][l][e] = a[e]; w.ab[bsel][1][l][e] = b[e]; }
  wave_sync();
  int j = l & 31;
  for (int r = 0; r < 16; r++) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 16; k++)
      acc = fmaf(bf2f(w.ab[bsel][0][i + 32 * (k >> 3)][k & 7]), bf2f(w.ab[bsel][1][j + 32 * (k >> 3)][k & 7]), acc);
    c[r] = acc;
  }
  return c;
}
// A: lane l holds A[l&15][8*(l>>4)+e]; B[8*(l>>4)+e][l&15]; D: j = l&15, i = 4*(l>>4)+r
inline f32x4_t mfma_16x16x32_bf16(bf16x8_t a, bf16x8_t b, f32x4_t c, int, int, int) {
  WaveScratch& w = wave();
  int bsel = w.gen & 1, l = lane_id();
  for (int e = 0; e < 8; e++) { w.ab[bsel][0][l][e] = a[e]; w.ab[bsel][1][l][e] = b[e]; }
  wave_sync();
  int j = l & 15;
  for (int r = 0; r < 4; r++) {
    int i = 4 * (l >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 32; k++)
      acc = fmaf(bf2f(w.ab[bsel][0][i + 16 * (k >> 3)][k & 7]), bf2f(w.ab[bsel][1][j + 16 * (k >> 3)][k & 7]), acc);
    c[r] = acc;
  }
  return c;
}

// ds_read_b64_tr_b16: see csrc/pcm_common.h (mapping measured on MI355X with tools/probes/trread.hip)
typedef short bf16x4_t __attribute__((ext_vector_type(4)));
inline bf16x4_t ds_read_tr16_b64(const void* p) {
  if (((uintptr_t)p) & 7) { fprintf(stderr, "emu: ds_read_b64_tr_b16 address not 8-byte aligned (the hardware returns the aligned address's data)\n"); abort(); }
  WaveScratch& w = wave();
  int bsel = w.gen & 1, l = lane_id();
  w.gp[bsel][l] = p;
  wave_sync();
  const int g0 = l & ~15, i = l & 15;
  bf16x4_t out;
  for (int j = 0; j < 4; j++) out[j] = ((const short*)w.gp[bsel][g0 + 4 * j + (i >> 2)])[i & 3];
  return out;
}

// LDS-DMA is asynchronous on the hardware: the data lands some time between the issue and the s_waitcnt vmcnt that
// retires it.  The emulator runs the two extremes: eager (default: lands at issue) and lazy (PCM_EMU_LAZY_DMA=1: lands
// only when a counted wait / __syncthreads retires it) -- a kernel whose results agree under both has no RAW / WAR
// dependence on the landing time across its barrier-separated phases.
inline void dma_do(const DmaOp& o) { if (o.src) memcpy(o.dst, o.src, o.size); else memset(o.dst, 0, o.size); }
inline void wait_vmcnt(int n) {
  std::vector<DmaOp>& q = g_cur->pend;
  size_t k = 0;
  while (q.size() - k > (size_t)n) dma_do(q[k++]);
  if (k) q.erase(q.begin(), q.begin() + k);
}
inline void dma_issue(void* dst, const void* src, int size) {
  DmaOp o{dst, src, size};
  if (g_lazy_dma) g_cur->pend.push_back(o); else dma_do(o);
}
// LDS dst = (first lane's ldsptr) + lane*size ; src = own gptr (offset must be 0)
inline void global_load_lds(const void* g, void* lds, int size, int offset, int) {
  if (offset != 0) { fprintf(stderr, "emu: global_load_lds offset!=0 unsupported\n"); abort(); }
  WaveScratch& w = wave();
  int bsel = w.gen & 1, l = lane_id();
  w.gp[bsel][l] = g;
  w.lp[bsel][l] = lds;
  wave_sync();
  dma_issue((char*)w.lp[bsel][0] + (size_t)l * size, g, size);
}
// raw buffer resource: byte address = base + voffset + soffset; out of range (voffset >= num_records - soffset) reads zero
struct BufferRsrc { const char* base; uint32_t num_records; };
inline BufferRsrc make_buffer_rsrc(const void* p, int stride, uint32_t num_records, int) {
  if (stride != 0) { fprintf(stderr, "emu: only raw (stride 0) buffers\n"); abort(); }
  return BufferRsrc{(const char*)p, num_records};
}
inline void buffer_load_lds(BufferRsrc rs, void* lds, int size, uint32_t voff, uint32_t soff, int imm, int) {
  if (imm != 0) { fprintf(stderr, "emu: buffer_load_lds imm offset unsupported\n"); abort(); }
  WaveScratch& w = wave();
  int bsel = w.gen & 1, l = lane_id();
  w.lp[bsel][l] = lds;
  w.u64[bsel][l] = soff;
  wave_sync();
  if (w.u64[bsel][0] != soff) { fprintf(stderr, "emu: buffer_load_lds soffset must be wave-uniform\n"); abort(); }
  bool oob = soff > rs.num_records || voff >= rs.num_records - soff;
  dma_issue((char*)w.lp[bsel][0] + (size_t)l * size, oob ? nullptr : rs.base + voff + soff, size);
}

// the same with an EXEC mask: every lane of the wave makes the call (the emulator's wave rendezvous needs all of them), inactive lanes
// transfer nothing and leave their LDS slot untouched -- what `if (active) buffer_load ... lds` does on the hardware
inline void buffer_load_lds_masked(BufferRsrc rs, void* lds, int size, uint32_t voff, uint32_t soff, bool active) {
  WaveScratch& w = wave();
  int bsel = w.gen & 1, l = lane_id();
  w.lp[bsel][l] = lds;
  w.u64[bsel][l] = soff;
  wave_sync();
  if (w.u64[bsel][0] != soff) { fprintf(stderr, "emu: buffer_load_lds soffset must be wave-uniform\n"); abort(); }
  if (!active) return;
  bool oob = soff > rs.num_records || voff >= rs.num_records - soff;
  dma_issue((char*)w.lp[bsel][0] + (size_t)l * size, oob ? nullptr : rs.base + voff + soff, size);
}

// plain buffer load of 16 bytes per lane (register destination): same address rule as buffer_load_lds, synchronous
typedef int emu_v4i __attribute__((ext_vector_type(4)));
inline emu_v4i buffer_load_b128(BufferRsrc rs, uint32_t voff, uint32_t soff, int) {
  emu_v4i r = {0, 0, 0, 0};
  bool oob = soff > rs.num_records || voff >= rs.num_records - soff;
  if (!oob) memcpy(&r, rs.base + voff + soff, 16);
  return r;
}

}  // namespace pcm_emu

#define threadIdx (pcm_emu::g_threadIdx)
#define blockIdx (pcm_emu::g_blockIdx)
#define blockDim (pcm_emu::g_blockDim)
#define gridDim (pcm_emu::g_gridDim)
#define __syncthreads() (pcm_emu::wait_vmcnt(0), pcm_emu::block_sync())
#define __builtin_amdgcn_s_barrier() pcm_emu::block_sync()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(...) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_getreg(x) 0u
// hardware-only code-generation controls of csrc/pcm_common.h
#define PCM_HW_ONLY(...)
#define PCM_WAVE_LDS_FENCE() pcm_emu::wave_sync()
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define PCM_LDS_LD64(dst, p) (dst) = *(const decltype(dst)*)(p)
#define PCM_LDS_LD128(dst, p) (dst) = *(const decltype(dst)*)(p)
#define PCM_LDS_LD128_LD64(d128, p128, d64, p64) do { (d128) = *(const decltype(d128)*)(p128); (d64) = *(const decltype(d64)*)(p64); } while (0)
#define PCM_LDS_ST64(p, val) *(decltype(val)*)(p) = (val)
#define PCM_LDS_WAIT_ALL() ((void)0)
#define PCM_PIN_V(x) ((void)0)
#define PCM_PIN_S(x) ((void)0)
#define PCM_KERNARG_REF(T, arr, i) ((arr)[i])
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 pcm_emu::mfma_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 pcm_emu::mfma_16x16x32_bf16
#define __builtin_amdgcn_global_load_lds(g, l, s, o, a) pcm_emu::global_load_lds((const void*)(g), (void*)(l), s, o, a)
#define __builtin_amdgcn_readfirstlane(x) pcm_emu::shfl_idx((x), 0)
#define PCM_AS1(p) (p)
#define PCM_AS3(p) (p)
typedef pcm_emu::BufferRsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, n, flags) pcm_emu::make_buffer_rsrc((const void*)(p), stride, n, flags)
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, l, size, voff, soff, imm, aux) pcm_emu::buffer_load_lds(rs, (void*)(l), size, voff, soff, imm, aux)
#define __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, aux) pcm_emu::buffer_load_b128(rs, voff, soff, aux)
#define PCM_DMA16_MASKED(rs, lds, voff, soff, active) pcm_emu::buffer_load_lds_masked(rs, (void*)(lds), 16, voff, soff, active)
#define PCM_WAIT_VMCNT(n) pcm_emu::wait_vmcnt(n)
#define PCM_WAIT_LGKMCNT0() ((void)0)

template <typename T> static inline T __shfl_xor(T v, int m, int = 64) { return pcm_emu::shfl_idx(v, pcm_emu::lane_id() ^ m); }
template <typename T> static inline T __shfl_down(T v, int d, int = 64) {
  int s = pcm_emu::lane_id() + d;
  return pcm_emu::shfl_idx(v, s > 63 ? pcm_emu::lane_id() : s);
}
template <typename T> static inline T __shfl(T v, int s, int = 64) { return pcm_emu::shfl_idx(v, s); }

static inline int __all(int pred) {
  pcm_emu::WaveScratch& w = pcm_emu::wave();
  int b = w.gen & 1;
  w.u64[b][pcm_emu::lane_id()] = pred ? 1 : 0;
  int alive = w.alive;
  pcm_emu::wave_sync();
  int ok = 1;
  for (int i = 0; i < 64 && i < alive; i++) ok &= (int)w.u64[b][i];
  return ok;
}
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
#define PCM_EXPF(x) expf(x)
#define PCM_EXP2F(x) exp2f(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

#define PCM_LAUNCH(kern, grid, block, smem, stream, ...) \
  pcm_emu::launch((const void*)(kern), (grid), (block), (smem), [=]() { kern(__VA_ARGS__); })
#define PCM_DYN_SMEM(name) char* name = pcm_emu::g_dyn_smem
