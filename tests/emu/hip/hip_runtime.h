// CPU stand-in for <hip/hip_runtime.h>: TEST INFRASTRUCTURE ONLY.
//
// The kernels of raytracer_amd/csrc/rtmi.hip (and the kernels_*.h parts it includes) are compiled unchanged as host C++ against this header and run one
// workgroup at a time, every work-item as a fiber on its own stack, so that AddressSanitizer and a
// debugger see each "device" access.  What it checks that a GPU run cannot report:
//   * every access to a hipMalloc'ed buffer is bounds-checked (the buffers are malloc blocks; ASan's
//     red zones follow them) and new buffers hold garbage, not zeros;
//   * a wave-level operation (__ballot, __any, __all, __shfl*, readfirstlane) only sees the lanes that
//     are at that very call: lanes that are elsewhere wait, as the hardware's exec mask makes them;
//     calls that do not see the whole live wave are counted per kernel (emu::report());
//   * a work-item that is still at __syncthreads() when the rest of its workgroup has finished, or
//     waves that wait at different barriers, end the run with a message instead of a hang.
// What it does not model: memory ordering (a block runs alone; all stores are visible at once), LDS
// bank conflicts, timing.  Nothing outside tests/ includes this file; the product is built by hipcc
// against the real header.
//
// Scheduling.  All lanes of a wave run, one after the other, until each is blocked in a wave-level
// operation, in __syncthreads(), or has returned.  Of the blocked lanes, those whose PROGRAM POSITION
// (the chain of return addresses from the kernel's frame inwards, compared outermost first, lower
// address first) is smallest go first: they are the lanes a structured-control-flow machine would
// be executing while the others wait at the reconvergence point further down.  Code built with
// frame pointers and without inlining (see tests/emu/Makefile) makes that chain exact.
#pragma once

#include <sys/mman.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#endif
#endif
#ifdef EMU_ASAN
#include <sanitizer/common_interface_defs.h>
#endif
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define EMU_TSAN 1
#endif
#endif
// Race detection (make -C tests/emu tsan).  Every work-item is a ThreadSanitizer fiber.  Switches between the
// scheduler and a waiting work-item carry no ordering; ordering comes only from what orders work-items on a GPU:
// a wave-level operation (among the lanes that take part), __syncthreads() (the workgroup), atomics, and the end
// of a workgroup (workgroups run one after the other here, so accesses of DIFFERENT workgroups are never
// reported).  What is reported: two work-items of one workgroup that touch the same LDS or global location, one
// of them writing, with none of the above in between -- also lanes of one wave that rely on executing in lock
// step without saying so (a wave_barrier between the store and the load is what says so).
#ifdef EMU_TSAN
#include <sanitizer/tsan_interface.h>
#define EMU_NO_TSAN __attribute__((no_sanitize("thread")))
#else
#define EMU_NO_TSAN
#endif

// ---- language ---------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
// One workgroup runs at a time, so function-local statics serve as its LDS.  They keep what the previous
// workgroup left in them (as LDS does) and, being ordinary globals, they get AddressSanitizer's red zones:
// an index past the end of a __shared__ array is reported.
#define __shared__ static
#define __constant__
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) (&(x))  // a file-scope __device__ variable is an ordinary global here

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return {x, y}; }

// HIP's integer/float min and max in the global namespace
#define EMU_MINMAX(T)                                      \
  static inline T min(T a, T b) { return b < a ? b : a; } \
  static inline T max(T a, T b) { return a < b ? b : a; }
EMU_MINMAX(int) EMU_MINMAX(unsigned) EMU_MINMAX(long) EMU_MINMAX(unsigned long) EMU_MINMAX(long long)
EMU_MINMAX(unsigned long long)
#undef EMU_MINMAX
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }

static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }

// ---- runtime API ------------------------------------------------------------------------
typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidConfiguration = 9, hipErrorNoDevice = 100, hipErrorLaunchFailure = 719 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum : unsigned { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
struct emu_stream_s { int id; };
typedef emu_stream_s* hipStream_t;
struct emu_event_s { std::chrono::steady_clock::time_point t; };
typedef emu_event_s* hipEvent_t;
struct hipDeviceProp_t {
  char name[256];
  char gcnArchName[256];
  size_t totalGlobalMem;
  int multiProcessorCount;
  int wavefront;
  int maxThreadsPerBlock;
  size_t sharedMemPerBlock;
  int clockRate;
};

// Switch stacks: saves the callee-saved registers of the caller on its stack, stores that stack pointer in
// *save_sp and continues on load_sp (x86-64 System V).  A hand-written switch instead of swapcontext(): no
// signal-mask system calls, and the sanitizer does not re-scan a whole fiber stack at every switch.
extern "C" void emu_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n"
    ".weak emu_switch\n"
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size emu_switch, .-emu_switch\n");

namespace emu {

inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// ---- fibers, one per work-item of the running workgroup ---------------------------------------
constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 1u << 19;  // per work-item; the lowest page is a guard page
constexpr int MAX_LANES = 1024;
constexpr int MAX_CHAIN = 40;

enum State { RUNNABLE, AT_WAVE_OP, AT_BARRIER, FINISHED };
enum Op { OP_BALLOT, OP_SHFL, OP_FIRST, OP_RENDEZVOUS };

struct Tid { unsigned x, y, z; };

struct Lane {
  void* sp;  // saved stack pointer while the lane is not running
  State st;
  Tid tid;
  int index;  // linear work-item id in the workgroup
  // the wave-level operation or barrier the lane is blocked in
  Op op;
  unsigned long long in, out;
  int src;  // OP_SHFL: source lane
  uintptr_t chain[MAX_CHAIN];  // program position, outermost frame first
  int nchain;
  void* fake_stack;
  void* top_frame;  // frame of the fiber's entry function: where the position chain ends
  void* tsan_fiber;
};

struct KernelStats {
  long launches = 0, blocks = 0, wave_ops = 0, partial_wave_ops = 0, shfl_from_absent = 0, barriers = 0;
};

struct Machine {
  std::mutex mu;  // one launch at a time, whichever host thread it comes from
  void* main_sp = nullptr;
  Lane* lanes = nullptr;
  char* stacks = nullptr;
  Lane* cur = nullptr;
  int nlanes = 0;
  dim3 bdim, gdim, bidx;
  const std::function<void()>* body = nullptr;
  const void* main_bottom = nullptr;
  size_t main_size = 0;
  const char* kernel = "";
  KernelStats* ks = nullptr;
  unsigned long long seed = 0, rng_state = 0;
  EMU_NO_TSAN unsigned rnd() {  // xorshift64*
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (unsigned)((rng_state * 2685821657736338717ull) >> 33);
  }
  std::map<std::string, KernelStats> stats;
  int last_error = 0;  // what hipGetLastError() returns next
  void* tsan_main = nullptr;
  char tsan_block_sync = 0;            // happens-before carriers (addresses only)
  char tsan_wave_sync[MAX_LANES / WAVE] = {};
  long allocations = 0, fail_allocation = -1;  // fault injection: the allocation with this ordinal fails
  long bad_launches = 0;
};

EMU_NO_TSAN inline Machine& M() {
  static Machine m;
  return m;
}

[[noreturn]] EMU_NO_TSAN inline void die(const char* what) {
  Machine& m = M();
  fprintf(stderr, "emu: %s  [kernel %s, block %u, work-item %d]\n", what, m.kernel, m.bidx.x, m.cur ? m.cur->index : -1);
  fflush(stderr);
  abort();
}

EMU_NO_TSAN inline void switch_to_main(Lane* l, bool dying) {
  Machine& m = M();
#ifdef EMU_ASAN
  __sanitizer_start_switch_fiber(dying ? nullptr : &l->fake_stack, m.main_bottom, m.main_size);
#endif
  (void)dying;
#ifdef EMU_TSAN
  // a work-item that has returned hands its history to the scheduler (the next workgroup starts after it);
  // one that merely waits hands over nothing
  __tsan_switch_to_fiber(m.tsan_main, dying ? 0 : __tsan_switch_to_fiber_no_sync);
#endif
  emu_switch(&l->sp, m.main_sp);
#ifdef EMU_ASAN
  __sanitizer_finish_switch_fiber(l->fake_stack, &m.main_bottom, &m.main_size);
#endif
}

EMU_NO_TSAN inline void trampoline() {
  Machine& m = M();
  Lane* l = m.cur;
#ifdef EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &m.main_bottom, &m.main_size);
#endif
  l->top_frame = __builtin_frame_address(0);
  (*m.body)();
  l->st = FINISHED;
  switch_to_main(l, true);
  abort();  // a finished fiber is never resumed
}

EMU_NO_TSAN inline void resume(Lane* l) {
  Machine& m = M();
  m.cur = l;
#ifdef EMU_ASAN
  void* fake = nullptr;
  __sanitizer_start_switch_fiber(&fake, m.stacks + (size_t)l->index * STACK_BYTES, STACK_BYTES);
#endif
#ifdef EMU_TSAN
  __tsan_switch_to_fiber(l->tsan_fiber, __tsan_switch_to_fiber_no_sync);
#endif
  emu_switch(&m.main_sp, l->sp);
#ifdef EMU_ASAN
  __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
  m.cur = nullptr;
}

// Record where the calling lane is: return addresses from the frame that called the emu entry point
// out to (not including) the trampoline, then reversed.
EMU_NO_TSAN __attribute__((noinline)) inline void capture_position(Lane* l, void* frame) {
  uintptr_t tmp[MAX_CHAIN];
  int n = 0;
  void** fp = (void**)frame;
  while (fp && (void*)fp < l->top_frame && n < MAX_CHAIN) {
    tmp[n++] = (uintptr_t)fp[1];
    void** up = (void**)fp[0];
    if (up <= fp) break;
    fp = up;
  }
  l->nchain = n;
  for (int i = 0; i < n; ++i) l->chain[i] = tmp[n - 1 - i];
}

EMU_NO_TSAN inline int compare_position(const Lane& a, const Lane& b) {
  const int n = std::min(a.nchain, b.nchain);
  for (int i = 0; i < n; ++i)
    if (a.chain[i] != b.chain[i]) return a.chain[i] < b.chain[i] ? -1 : 1;
  return a.nchain == b.nchain ? 0 : (a.nchain < b.nchain ? -1 : 1);
}

// Block the calling lane in a wave-level operation; returns when the scheduler has filled `out`.
EMU_NO_TSAN __attribute__((noinline)) inline unsigned long long wave_op(Op op, unsigned long long in, int src, void* frame) {
  Machine& m = M();
  Lane* l = m.cur;
  if (!l) die("wave-level operation outside a kernel");
  l->op = op; l->in = in; l->src = src;
  capture_position(l, frame);
  l->st = AT_WAVE_OP;
#ifdef EMU_TSAN
  __tsan_release(&m.tsan_wave_sync[l->index / WAVE]);
#endif
  switch_to_main(l, false);
#ifdef EMU_TSAN
  __tsan_acquire(&m.tsan_wave_sync[l->index / WAVE]);
#endif
  return l->out;
}

EMU_NO_TSAN __attribute__((noinline)) inline void barrier(void* frame) {
  Machine& m = M();
  Lane* l = m.cur;
  if (!l) die("__syncthreads outside a kernel");
  capture_position(l, frame);
  l->st = AT_BARRIER;
#ifdef EMU_TSAN
  __tsan_release(&m.tsan_block_sync);
#endif
  switch_to_main(l, false);
#ifdef EMU_TSAN
  __tsan_acquire(&m.tsan_block_sync);
#endif
}

// The lanes of wave [lo, hi) that go next: the blocked lanes at the smallest program position.
// Returns their state (AT_WAVE_OP or AT_BARRIER), or FINISHED if no lane of the wave is blocked.
EMU_NO_TSAN inline State wave_front(int lo, int hi, std::vector<int>& front) {
  Machine& m = M();
  front.clear();
  int lead = -1;
  for (int i = lo; i < hi; ++i) {
    Lane& l = m.lanes[i];
    if (l.st == FINISHED) continue;
    if (lead < 0) { lead = i; front.push_back(i); continue; }
    const int c = compare_position(l, m.lanes[lead]);
    if (c < 0) { lead = i; front.clear(); front.push_back(i); }
    else if (c == 0) front.push_back(i);
  }
  if (lead < 0) return FINISHED;
  const State st = m.lanes[lead].st;
  for (int i : front)
    if (m.lanes[i].st != st || (st == AT_WAVE_OP && m.lanes[i].op != m.lanes[lead].op)) die("lanes at one program position disagree on the operation");
  return st;
}

EMU_NO_TSAN inline void complete_wave_op(int lo, int hi, const std::vector<int>& front) {
  Machine& m = M();
  int live = 0;
  for (int i = lo; i < hi; ++i) live += m.lanes[i].st != FINISHED;
  ++m.ks->wave_ops;
  if ((int)front.size() != live) ++m.ks->partial_wave_ops;
  const Op op = m.lanes[front[0]].op;
  unsigned long long ballot = 0;
  for (int i : front)
    if (m.lanes[i].in) ballot |= 1ull << (i - lo);
  for (int i : front) {
    Lane& l = m.lanes[i];
    switch (op) {
      case OP_BALLOT: l.out = ballot; break;
      case OP_FIRST: l.out = m.lanes[front[0]].in; break;
      case OP_RENDEZVOUS: l.out = 0; break;
      case OP_SHFL: {
        const int s = lo + l.src;
        bool present = l.src >= 0 && l.src < WAVE && s < hi;
        if (present) present = std::find(front.begin(), front.end(), s) != front.end();
        if (l.src < 0 || l.src >= WAVE) l.out = l.in;  // out of the wave: own value (HIP's rule)
        else if (present) l.out = m.lanes[s].in;
        else { l.out = 0xdeadbeefdeadbeefull; ++m.ks->shfl_from_absent; }
        break;
      }
    }
  }
  for (int i : front) m.lanes[i].st = RUNNABLE;
}

EMU_NO_TSAN inline void run_block() {
  Machine& m = M();
  const int n = m.nlanes;
  const int nw = (n + WAVE - 1) / WAVE;
  for (int i = 0; i < n; ++i) {
    Lane& l = m.lanes[i];
    l.st = RUNNABLE;
    l.index = i;
    l.tid.x = i % m.bdim.x;
    l.tid.y = (i / m.bdim.x) % m.bdim.y;
    l.tid.z = i / (m.bdim.x * m.bdim.y);
    l.nchain = 0;
    l.fake_stack = nullptr;
#ifdef EMU_TSAN
    l.tsan_fiber = __tsan_create_fiber(0);  // inherits what the scheduler has seen so far, like a new thread
#endif
    // first switch to the lane: six zeroed registers are popped, then `ret` enters trampoline()
    void** top = (void**)(m.stacks + (size_t)(i + 1) * STACK_BYTES);
    top[-1] = nullptr;
    top[-2] = (void*)&trampoline;
    for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
    l.sp = (void*)(top - 8);
  }
  // Waves advance one wave-level operation at a time.  With EMU_SEED unset (or 0) a wave keeps going until
  // it reaches a barrier or finishes and lanes run in index order; with a seed, the wave that steps next and
  // the order of the lanes within it are drawn at random, which varies how the waves of a workgroup interleave.
  std::vector<int> front, steppable, lane_order(WAVE);
  std::vector<char> wstate(nw, 0);  // 0: can step, 1: waits at a barrier, 2: finished
  for (;;) {
    steppable.clear();
    int finished_waves = 0;
    for (int w = 0; w < nw; ++w) {
      if (wstate[w] == 0) steppable.push_back(w);
      finished_waves += wstate[w] == 2;
    }
    if (finished_waves == nw) {
#ifdef EMU_TSAN
      for (int i = 0; i < n; ++i) __tsan_destroy_fiber(m.lanes[i].tsan_fiber);
#endif
      return;
    }
    if (steppable.empty()) {
      // every wave that still runs is at a barrier: they must all be at the same one
      int lead = -1;
      for (int w = 0; w < nw; ++w) {
        if (wstate[w] != 1) continue;
        const int lo = w * WAVE, hi = std::min(n, lo + WAVE);
        if (wave_front(lo, hi, front) != AT_BARRIER) die("emulator: a waiting wave is not at a barrier");
        if (lead < 0) lead = front[0];
        else if (compare_position(m.lanes[front[0]], m.lanes[lead]) != 0) die("waves of a workgroup wait at different __syncthreads()");
        int live = 0;
        for (int i = lo; i < hi; ++i) live += m.lanes[i].st != FINISHED;
        if ((int)front.size() != live) ++m.ks->partial_wave_ops;
        for (int i : front) m.lanes[i].st = RUNNABLE;
        wstate[w] = 0;
      }
      ++m.ks->barriers;
      continue;
    }
    const int w = m.seed ? steppable[m.rnd() % steppable.size()] : steppable[0];
    const int lo = w * WAVE, hi = std::min(n, lo + WAVE);
    do {
      for (int i = lo; i < hi; ++i) lane_order[i - lo] = i;
      if (m.seed)
        for (int i = hi - lo - 1; i > 0; --i) std::swap(lane_order[i], lane_order[m.rnd() % (unsigned)(i + 1)]);
      for (int k = 0; k < hi - lo; ++k)
        if (m.lanes[lane_order[k]].st == RUNNABLE) resume(&m.lanes[lane_order[k]]);
      const State st = wave_front(lo, hi, front);
      if (st == AT_WAVE_OP) complete_wave_op(lo, hi, front);
      else wstate[w] = st == AT_BARRIER ? 1 : 2;
    } while (wstate[w] == 0 && (!m.seed || (m.rnd() & 3) != 0));
  }
}

template <class... KArgs, class... Args>
EMU_NO_TSAN inline void launch(const char* name, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t, hipStream_t, Args&&... args) {
  Machine& m = M();
  std::lock_guard<std::mutex> lock(m.mu);
  std::tuple<std::decay_t<KArgs>...> held(static_cast<std::decay_t<KArgs>>(args)...);
  const std::function<void()> body = [&] { std::apply(kernel, held); };
  const int n = (int)(block.x * block.y * block.z);
  if (n <= 0 || n > MAX_LANES || grid.x == 0 || grid.y == 0 || grid.z == 0) {
    // the runtime rejects an empty grid or workgroup: nothing runs and the launch reports an error
    fprintf(stderr, "emu: launch of %s with grid (%u,%u,%u) block (%u,%u,%u) rejected\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z);
    __atomic_store_n(&m.last_error, 9, __ATOMIC_RELAXED);
    ++m.bad_launches;
    return;
  }
  if (!m.lanes) {
    m.lanes = new Lane[MAX_LANES];
    m.stacks = (char*)mmap(nullptr, (size_t)MAX_LANES * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m.stacks == (char*)MAP_FAILED) die("no memory for the fiber stacks");
    for (int i = 0; i < MAX_LANES; ++i) mprotect(m.stacks + (size_t)i * STACK_BYTES, 4096, PROT_NONE);
    m.seed = (unsigned long long)env_int("EMU_SEED", 0);
    m.rng_state = m.seed * 0x9E3779B97F4A7C15ull + 1;
  }
#ifdef EMU_TSAN
  m.tsan_main = __tsan_get_current_fiber();
#endif
  m.kernel = name;
  m.ks = &m.stats[name];
  ++m.ks->launches;
  m.body = &body;
  m.nlanes = n;
  m.bdim = block;
  m.gdim = grid;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        m.bidx = dim3(bx, by, bz);
        ++m.ks->blocks;
        run_block();
      }
  m.body = nullptr;
  m.kernel = "";
}

// Per-kernel totals since the process started, one line per kernel.
EMU_NO_TSAN inline void report(FILE* f) {
  Machine& m = M();
  std::lock_guard<std::mutex> lock(m.mu);
  fprintf(f, "emu-bad-launches %ld\n", m.bad_launches);
  for (const auto& kv : m.stats)
    fprintf(f, "emu-stats launches=%ld blocks=%ld wave_ops=%ld partial_wave_ops=%ld shfl_from_absent=%ld barriers=%ld kernel=%s\n",
            kv.second.launches, kv.second.blocks, kv.second.wave_ops, kv.second.partial_wave_ops, kv.second.shfl_from_absent,
            kv.second.barriers, kv.first.c_str());
}

template <class T>
inline unsigned long long to_bits(T v) {
  static_assert(sizeof(T) <= 8, "wave-level operations move at most 8 bytes");
  unsigned long long b = 0;
  memcpy(&b, &v, sizeof(T));
  return b;
}
template <class T>
inline T from_bits(unsigned long long b) {
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}

}  // namespace emu

#define threadIdx (emu::M().cur->tid)
#define blockIdx (emu::M().bidx)
#define blockDim (emu::M().bdim)
#define gridDim (emu::M().gdim)
static constexpr int warpSize = 64;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu::launch(#kernel, kernel, grid, block, shmem, stream, ##__VA_ARGS__)

// ---- wave-level operations and barriers (never inlined: their frame is the innermost position) --
__attribute__((noinline)) static unsigned long long __ballot(int pred) {
  return emu::wave_op(emu::OP_BALLOT, pred != 0, 0, __builtin_frame_address(0));
}
__attribute__((noinline)) static int __any(int pred) {
  return emu::wave_op(emu::OP_BALLOT, pred != 0, 0, __builtin_frame_address(0)) != 0;
}
__attribute__((noinline)) static int __all(int pred) {
  return emu::wave_op(emu::OP_BALLOT, pred == 0, 0, __builtin_frame_address(0)) == 0;
}
template <class T>
__attribute__((noinline)) static T __shfl(T v, int src_lane, int width = 64) {
  (void)width;
  return emu::from_bits<T>(emu::wave_op(emu::OP_SHFL, emu::to_bits(v), src_lane & 63, __builtin_frame_address(0)));
}
template <class T>
__attribute__((noinline)) static T __shfl_down(T v, unsigned delta, int width = 64) {
  (void)width;
  const int lane = emu::M().cur->index & 63;
  return emu::from_bits<T>(emu::wave_op(emu::OP_SHFL, emu::to_bits(v), lane + (int)delta, __builtin_frame_address(0)));
}
template <class T>
__attribute__((noinline)) static T __shfl_up(T v, unsigned delta, int width = 64) {
  (void)width;
  const int lane = emu::M().cur->index & 63;
  return emu::from_bits<T>(emu::wave_op(emu::OP_SHFL, emu::to_bits(v), lane - (int)delta, __builtin_frame_address(0)));
}
template <class T>
__attribute__((noinline)) static T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  const int lane = emu::M().cur->index & 63;
  return emu::from_bits<T>(emu::wave_op(emu::OP_SHFL, emu::to_bits(v), lane ^ mask, __builtin_frame_address(0)));
}
__attribute__((noinline)) static int emu_readfirstlane(int v) {
  return (int)emu::wave_op(emu::OP_FIRST, (unsigned)v, 0, __builtin_frame_address(0));
}
// lanes exchange data through LDS around this one: all lanes at the call must have arrived
__attribute__((noinline)) static void emu_wave_barrier() {
  (void)emu::wave_op(emu::OP_RENDEZVOUS, 0, 0, __builtin_frame_address(0));
}
__attribute__((noinline)) static void __syncthreads() { emu::barrier(__builtin_frame_address(0)); }
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- atomics ------------------------------------------------------------------------------------
// One fiber runs at a time, so plain read-modify-write is atomic here; the race-detection build uses real atomic
// operations so that ThreadSanitizer knows these accesses are meant to be concurrent.
#ifdef EMU_TSAN
template <class T, class U>
static inline T atomicAdd(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicSub(T* p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicOr(T* p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicAnd(T* p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicExch(T* p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicMin(T* p, U v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while ((T)v < old && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
template <class T, class U>
static inline T atomicMax(T* p, U v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (old < (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
template <class T, class U, class V>
static inline T atomicCAS(T* p, U cmp, V v) {
  T expected = (T)cmp;
  __atomic_compare_exchange_n(p, &expected, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return expected;
}
#else
template <class T, class U>
static inline T atomicAdd(T* p, U v) { T old = *p; *p = (T)(old + (T)v); return old; }
template <class T, class U>
static inline T atomicSub(T* p, U v) { T old = *p; *p = (T)(old - (T)v); return old; }
template <class T, class U>
static inline T atomicMin(T* p, U v) { T old = *p; if ((T)v < old) *p = (T)v; return old; }
template <class T, class U>
static inline T atomicMax(T* p, U v) { T old = *p; if (old < (T)v) *p = (T)v; return old; }
template <class T, class U>
static inline T atomicOr(T* p, U v) { T old = *p; *p = (T)(old | (T)v); return old; }
template <class T, class U>
static inline T atomicAnd(T* p, U v) { T old = *p; *p = (T)(old & (T)v); return old; }
template <class T, class U>
static inline T atomicExch(T* p, U v) { T old = *p; *p = (T)v; return old; }
template <class T, class U, class V>
static inline T atomicCAS(T* p, U cmp, V v) { T old = *p; if (old == (T)cmp) *p = (T)v; return old; }
#endif

// ---- memory, streams, events ------------------------------------------------------------------
static inline const char* hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorNoDevice: return "no device";
    case hipErrorInvalidConfiguration: return "invalid configuration argument";
    default: return "emulated failure";
  }
}
static inline hipError_t emu_alloc(void** p, size_t n) {
  if (!p) return hipErrorInvalidValue;
  if (__atomic_fetch_add(&emu::M().allocations, 1L, __ATOMIC_RELAXED) == __atomic_load_n(&emu::M().fail_allocation, __ATOMIC_RELAXED))
    return hipErrorOutOfMemory;  // (host threads may allocate at the same time)
  void* q = nullptr;
  if (posix_memalign(&q, 256, n ? n : 1) != 0) return hipErrorOutOfMemory;
  memset(q, emu::env_int("EMU_FILL", 0xA5), n ? n : 1);  // fresh device memory is not zero
  *p = q;
  return hipSuccess;
}
template <class T>
static inline hipError_t hipMalloc(T** p, size_t n) { return emu_alloc((void**)p, n); }
template <class T>
static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { return emu_alloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  if (n) memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy(d, s, n, k); }
static inline hipError_t hipMemset(void* d, int v, size_t n) {
  if (n) memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { return hipMemset(d, v, n); }
static inline hipError_t hipGetSymbolAddress(void** p, const void* symbol) { *p = const_cast<void*>(symbol); return hipSuccess; }
static inline hipError_t hipMemcpyToSymbol(const void* symbol, const void* src, size_t n, size_t offset = 0, hipMemcpyKind = hipMemcpyHostToDevice) {
  memmove((char*)const_cast<void*>(symbol) + offset, src, n);
  return hipSuccess;
}
static inline hipError_t hipGetLastError() { return __atomic_exchange_n(&emu::M().last_error, 0, __ATOMIC_RELAXED); }
static inline hipError_t hipPeekAtLastError() { return __atomic_load_n(&emu::M().last_error, __ATOMIC_RELAXED); }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = emu::env_int("EMU_DEVICES", 1); return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d >= 0 && d < emu::env_int("EMU_DEVICES", 1) ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "emulated gfx950");
  snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
  p->totalGlobalMem = (size_t)8 << 30;
  p->multiProcessorCount = emu::env_int("EMU_CUS", 2);
  p->wavefront = 64;
  p->maxThreadsPerBlock = 1024;
  p->sharedMemPerBlock = 160 * 1024;
  p->clockRate = 2400000;
  return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new emu_stream_s{0}; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = new emu_stream_s{0}; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event_s{}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
