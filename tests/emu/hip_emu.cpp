// TEST INFRASTRUCTURE ONLY — fiber scheduler for tests/emu/hip_emu.h
#include "hip_emu.h"
#include <algorithm>
#include <map>

namespace pcm_emu {
std::vector<Fiber> g_fibers;
std::vector<WaveScratch> g_waves;
void* g_sched_sp = nullptr;
}  // namespace pcm_emu
// Minimal SysV x86-64 context switch (callee-saved registers + stack pointer).  swapcontext() also saves / restores the signal mask
// with a system call per switch; a wave-collective does one switch per lane, so that syscall dominated the emulator's run time.
asm(R"(
.text
.globl pcm_ctx_switch
.type pcm_ctx_switch,@function
pcm_ctx_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size pcm_ctx_switch,.-pcm_ctx_switch
)");
namespace pcm_emu {
Fiber* g_cur = nullptr;
int g_block_arrived = 0, g_block_alive = 0;
dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
std::function<void()> g_body;
char* g_dyn_smem = nullptr;
bool g_lazy_dma = false;
int g_wave_order = 0;
static const size_t STACK = 256 * 1024;

void yield_to_sched() { pcm_ctx_switch(&g_cur->sp, g_sched_sp); }

void wave_sync() {
  Fiber* f = g_cur;
  WaveScratch& w = g_waves[f->wave];
  w.arrived++;
  if (w.arrived == w.alive) {
    w.arrived = 0;
    w.gen++;
    for (auto& o : g_fibers)
      if (o.wave == f->wave && o.st == WAIT_WAVE) o.st = RUNNABLE;
    return;  // last arriver continues immediately
  }
  f->st = WAIT_WAVE;
  yield_to_sched();
}

void block_sync() {
  Fiber* f = g_cur;
  g_block_arrived++;
  if (g_block_arrived == g_block_alive) {
    g_block_arrived = 0;
    for (auto& o : g_fibers)
      if (o.st == WAIT_BLOCK) o.st = RUNNABLE;
    return;
  }
  f->st = WAIT_BLOCK;
  yield_to_sched();
}

static void trampoline() {
  g_body();
  wait_vmcnt(0);
  Fiber* f = g_cur;
  f->st = DONE;
  // a finished lane no longer takes part in rendezvous
  WaveScratch& w = g_waves[f->wave];
  w.alive--;
  g_block_alive--;
  if (w.alive > 0 && w.arrived == w.alive) {
    w.arrived = 0; w.gen++;
    for (auto& o : g_fibers) if (o.wave == f->wave && o.st == WAIT_WAVE) o.st = RUNNABLE;
  }
  if (g_block_alive > 0 && g_block_arrived == g_block_alive) {
    g_block_arrived = 0;
    for (auto& o : g_fibers) if (o.st == WAIT_BLOCK) o.st = RUNNABLE;
  }
  pcm_ctx_switch(&f->sp, g_sched_sp);
  __builtin_trap();   // a finished fiber is never resumed
}

// gfx950 launch limits (MI355X_MICROARCH.md): 160 KiB of LDS per CU / workgroup, of which a kernel may use more than 64 KiB of DYNAMIC
// LDS only after hipFuncSetAttribute(MaxDynamicSharedMemorySize) raised its limit; 1024 threads per block; grid.y/z <= 65535.
// (static __shared__ arrays are host statics here and not counted: tests/test_codeobj_limits.py checks them on the gfx950 code object.)
hipError_t g_last_error = 0;
static std::map<const void*, size_t> g_max_dyn;
static const size_t LDS_PER_CU = 160 * 1024, LDS_DEFAULT_DYN = 64 * 1024;
hipError_t set_max_dyn_lds(const void* fn, int bytes) {
  if (bytes < 0 || (size_t)bytes > LDS_PER_CU) return hipErrorInvalidValue;
  g_max_dyn[fn] = (size_t)bytes;
  return 0;
}

void launch(const void* fn, dim3 grid, dim3 block, size_t smem, std::function<void()> body) {
  {
    auto it = g_max_dyn.find(fn);
    const size_t lim = it == g_max_dyn.end() ? LDS_DEFAULT_DYN : std::max(it->second, LDS_DEFAULT_DYN);
    const unsigned long nthr = (unsigned long)block.x * block.y * block.z;
    if (smem > lim || smem > LDS_PER_CU || nthr == 0 || nthr > 1024 || grid.x == 0 || grid.y == 0 || grid.z == 0 || grid.y > 65535 ||
        grid.z > 65535 || grid.x > 2147483647u) {
      fprintf(stderr, "pcm_emu: launch REJECTED (gfx950 limits): dynamic LDS %zu B (limit %zu), block %lu, grid (%u,%u,%u)\n", smem, lim, nthr,
              grid.x, grid.y, grid.z);
      g_last_error = hipErrorInvalidValue;
      return;
    }
  }
  int nthreads = block.x * block.y * block.z;
  int nwaves = (nthreads + 63) / 64;
  g_body = body;
  { const char* e = getenv("PCM_EMU_LAZY_DMA"); g_lazy_dma = e && e[0] == '1'; }
  { const char* e = getenv("PCM_EMU_ORDER"); g_wave_order = e ? atoi(e) : 0; }
  g_blockDim = block;
  g_gridDim = grid;
  std::vector<char> dyn(smem + 64);
  g_dyn_smem = (char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
  if ((int)g_fibers.size() < nthreads) {
    size_t old = g_fibers.size();
    g_fibers.resize(nthreads);
    for (size_t i = old; i < g_fibers.size(); i++) g_fibers[i].stack = (char*)malloc(STACK);
  }
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        g_blockIdx = dim3(bx, by, bz);
        g_waves.assign(nwaves, WaveScratch());
        for (int w = 0; w < nwaves; w++) {
          g_waves[w].arrived = 0; g_waves[w].gen = 0;
          g_waves[w].alive = std::min(64, nthreads - 64 * w);
        }
        g_block_arrived = 0;
        g_block_alive = nthreads;
        for (int t = 0; t < nthreads; t++) {
          Fiber& f = g_fibers[t];
          f.st = RUNNABLE;
          f.pend.clear();
          f.lin = t; f.wave = t / 64; f.lane = t % 64;
          f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          // fresh stack: [top-8] dummy return address, [top-16] entry point popped by the switch's `ret` (so the entry sees
          // rsp = 16n + 8 like after a call), six zeroed callee-saved slots below
          uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
          void** sp = (void**)top;
          *--sp = nullptr;
          *--sp = (void*)&trampoline;
          for (int r = 0; r < 6; r++) *--sp = nullptr;
          f.sp = (void*)sp;
        }
        int done = 0;
        while (done < nthreads) {
          bool progressed = false;
          for (int ti = 0; ti < 64 * nwaves; ti++) {
            // wave visiting order (PCM_EMU_ORDER): the hardware runs the waves of a block concurrently, so a kernel may not depend on
            // which wave reaches a point first; 0 = ascending (default), 1 = descending, 2 = odd waves first.  A missing barrier between
            // one wave's LDS writes and another wave's reads shows up as a result that changes with the order.
            const int w_ = ti / 64, l_ = ti % 64;
            int wv = w_;
            if (g_wave_order == 1) wv = nwaves - 1 - w_;
            else if (g_wave_order == 2) wv = (2 * w_ + 1 < nwaves) ? 2 * w_ + 1 : 2 * (w_ - nwaves / 2);
            const int t = wv * 64 + l_;
            if (t >= nthreads) continue;
            Fiber& f = g_fibers[t];
            if (f.st != RUNNABLE) continue;
            progressed = true;
            g_cur = &f;
            g_threadIdx = f.tid;
            pcm_ctx_switch(&g_sched_sp, f.sp);
            if (f.st == DONE) done++;
          }
          if (!progressed) {
            fprintf(stderr, "pcm_emu: DEADLOCK in block (%u,%u,%u): %d/%d done (divergent barrier?)\n",
                    bx, by, bz, done, nthreads);
            abort();
          }
        }
      }
  g_cur = nullptr;
}
}  // namespace pcm_emu

// self-test hook for tests/test_emu_kernels.py::test_emulator_enforces_launch_limits: launch an empty kernel with the given dynamic LDS
// request / block size (optionally after raising the kernel's dynamic-LDS cap) and return what hipGetLastError would report
static void emu_empty_kernel() {}
extern "C" int pcm_emu_try_launch(long dyn_lds, int block, int gy, long raise_cap_to) {
  if (raise_cap_to >= 0) { if (hipFuncSetAttribute((const void*)emu_empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)raise_cap_to)) return -1; }
  PCM_LAUNCH(emu_empty_kernel, dim3(1, gy), dim3(block), (size_t)dyn_lds, nullptr);
  return hipGetLastError();
}
