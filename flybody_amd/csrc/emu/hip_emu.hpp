// TEST INFRASTRUCTURE ONLY -- a minimal single-threaded emulation of the HIP constructs used by
// the flybody kernels, so that the *kernel source itself* can be exercised on a GPU-less build
// box by `pytest -m "not gpu"` (tests/test_kernel_emulation.py).  It is never loaded by the
// product path: flybody_amd/engine.py only ever opens libflybody_hip.so and raises if no GPU
// is present.
//
// One workgroup = 64 user-space fibers (hand-rolled x86-64 context switch) executed round-robin;
// __syncthreads() and the wave shuffles are "yield to the next fiber" points.  Setting
// FB_EMU_REVERSE=1 runs the lanes in reverse order between barriers, which turns most missing-
// barrier races into result differences.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <algorithm>

#if !defined(__x86_64__)
#error "hip_emu.hpp supports x86-64 only"
#endif

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

using std::min;
using std::max;

struct emu_dim3 { unsigned x, y, z; emu_dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef emu_dim3 dim3;
static emu_dim3 threadIdx, blockIdx;     // single OS thread: plain globals, swapped per fiber

// ---- fibers -------------------------------------------------------------------------
#define EMU_MAXLANES 256
static int EMU_LANES = 64;          // threads per block of the current launch
#define EMU_STACK (512*1024)
struct EmuFiber { void* sp; char* stack; bool done; };
static EmuFiber emu_fib[EMU_MAXLANES];
static void* emu_main_sp;
static int emu_cur = -1, emu_ndone = 0, emu_dir = 1;
static void (*emu_entry_fn)(void*);
static void* emu_entry_arg;
static uint64_t emu_exch[EMU_MAXLANES];

extern "C" void emu_ctx_switch(void** from_sp, void* to_sp);
__asm__(
    ".text\n.globl emu_ctx_switch\n.type emu_ctx_switch,@function\nemu_ctx_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size emu_ctx_switch,.-emu_ctx_switch\n");

static inline int emu_next(int i) { return (i + emu_dir + EMU_LANES) % EMU_LANES; }

static inline void emu_yield() {
  int from = emu_cur;
  int to = emu_next(from);
  // skip finished fibers
  for (int k = 0; k < EMU_LANES && emu_fib[to].done; k++) to = emu_next(to);
  if (emu_fib[to].done || to == from) return;
  emu_cur = to; threadIdx.x = to;
  emu_ctx_switch(&emu_fib[from].sp, emu_fib[to].sp);
  threadIdx.x = emu_cur;
}

static void emu_trampoline() {
  emu_entry_fn(emu_entry_arg);
  int me = emu_cur;
  emu_fib[me].done = true; emu_ndone++;
  if (emu_ndone == EMU_LANES) { void* dummy; emu_ctx_switch(&dummy, emu_main_sp); }
  int to = emu_next(me);
  while (emu_fib[to].done) to = emu_next(to);
  emu_cur = to; threadIdx.x = to;
  void* dummy;
  emu_ctx_switch(&dummy, emu_fib[to].sp);
  abort();
}

static void emu_run_block(void (*fn)(void*), void* arg, unsigned block) {
  const char* rev = getenv("FB_EMU_REVERSE");
  emu_dir = (rev && rev[0] == '1') ? -1 : 1;
  emu_entry_fn = fn; emu_entry_arg = arg; emu_ndone = 0;
  blockIdx.x = block;
  for (int i = 0; i < EMU_LANES; i++) {
    if (!emu_fib[i].stack) emu_fib[i].stack = (char*)aligned_alloc(64, EMU_STACK);
    uintptr_t top = ((uintptr_t)(emu_fib[i].stack + EMU_STACK)) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8*sizeof(void*));
    for (int k = 0; k < 6; k++) sp[k] = nullptr;
    sp[6] = (void*)&emu_trampoline; sp[7] = nullptr;
    emu_fib[i].sp = sp; emu_fib[i].done = false;
  }
  int first = emu_dir > 0 ? 0 : EMU_LANES - 1;
  emu_cur = first; threadIdx.x = first;
  emu_ctx_switch(&emu_main_sp, emu_fib[first].sp);
}

// ---- device intrinsics ----------------------------------------------------------------
static inline void __syncthreads() { emu_yield(); }
template <typename T> static inline T emu_exchange(T v, int src) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  int me = threadIdx.x;
  uint64_t bits = 0; memcpy(&bits, &v, sizeof(T));
  emu_exch[me] = bits;
  emu_yield();                       // everyone has published
  T r; uint64_t rb = emu_exch[(me & ~63) + (src & 63)]; memcpy(&r, &rb, sizeof(T));
  emu_yield();                       // everyone has read
  return r;
}
template <typename T> static inline T __shfl_xor(T v, int m, int = 64) { return emu_exchange(v, ((int)threadIdx.x & 63) ^ m); }
template <typename T> static inline T __shfl_up(T v, int d, int = 64) { int me = threadIdx.x & 63; return emu_exchange(v, me >= d ? me - d : me); }
template <typename T> static inline T __shfl(T v, int src, int = 64) { return emu_exchange(v, src); }
static inline unsigned long long __ballot(int pred) {
  int me = threadIdx.x;
  emu_exch[me] = pred ? 1 : 0;
  emu_yield();
  unsigned long long r = 0;
  for (int i = 0; i < 64; i++) if (emu_exch[(me & ~63) + i]) r |= (1ull << i);
  emu_yield();
  return r;
}
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }   // fibers run one at a time
static inline int atomicOr(int* p, int v) { int o = *p; *p = o | v; return o; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

// ---- runtime API ------------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
typedef emu_event* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
static inline const char* hipGetErrorString(hipError_t) { return "emu error"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : 1; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind) {
  for (size_t r = 0; r < h; r++) memcpy((char*)d + r*dp, (const char*)s + r*sp, w);
  return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess;
}

// kernel launch: run the blocks one after another, 64 fibers each
#include <tuple>
#include <utility>
template <typename F, typename Tup, size_t... I>
static inline void emu_apply(F f, Tup& t, std::index_sequence<I...>) { f(std::get<I>(t)...); }
template <typename F, typename... Args>
static inline void emu_launch(F kernel, dim3 grid, dim3 block, Args... args) {
  if (block.x % 64 != 0 || block.x > EMU_MAXLANES) { fprintf(stderr, "hip_emu: block size must be a multiple of 64 and <= 256\n"); abort(); }
  EMU_LANES = (int)block.x;
  auto tup = std::make_tuple(args...);
  struct Ctx { F k; decltype(tup)* t; } ctx = {kernel, &tup};
  auto thunk = [](void* p) { Ctx* c = (Ctx*)p; emu_apply(c->k, *c->t, std::index_sequence_for<Args...>{}); };
  for (unsigned b = 0; b < grid.x; b++) emu_run_block(thunk, &ctx, b);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)
