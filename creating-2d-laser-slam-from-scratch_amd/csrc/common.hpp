// Shared host-side plumbing of liblslam_gpu.so: context, error reporting, device buffers,
// HIP-event kernel timing.  gfx950 only, no compatibility layers.
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/lslam_gpu.h"

namespace lslam {

extern thread_local std::string g_last_error;

struct KernelTimer {
  struct Rec {
    const char* name;
    hipEvent_t e0, e1;
  };
  bool enabled = false;
  std::string only;  // empty = time every kernel, else only launches under this name
  std::vector<Rec> pending;
  std::vector<hipEvent_t> pool;
  std::map<std::string, std::pair<int64_t, double>> totals;  // name -> (launches, ms)

  hipEvent_t get() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
  // fold finished records into totals (synchronises on the last event)
  void drain() {
    for (auto& r : pending) {
      (void)hipEventSynchronize(r.e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, r.e0, r.e1);
      auto& t = totals[r.name];
      t.first += 1;
      t.second += ms;
      pool.push_back(r.e0);
      pool.push_back(r.e1);
    }
    pending.clear();
  }
  ~KernelTimer() {
    for (auto& r : pending) {
      (void)hipEventDestroy(r.e0);
      (void)hipEventDestroy(r.e1);
    }
    for (auto e : pool) (void)hipEventDestroy(e);
  }
};

}  // namespace lslam

struct lslam_context {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;
  lslam::KernelTimer timer;
  hipDeviceProp_t prop;
  void* d_small = nullptr;  // 256 bytes of device memory that exist as long as the context does (collective staging)
  // work that objects of this context have deferred and that must be on the stream before a synchronise means
  // "everything is done" (the log-odds map's pipelined apply): (object, flush function)
  std::vector<std::pair<void*, int (*)(void*)>> pre_sync;
  // checks that can only be made once the stream has drained (a counter kernels bump in pinned host memory): run by
  // lslam_synchronize AFTER the wait; the first failure is the call's result
  std::vector<std::pair<void*, int (*)(void*)>> post_sync;

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error = buf;
    lslam::g_last_error = buf;
    return code;
  }
};

namespace lslam {

#define LSLAM_HIP(ctx, expr)                                                                   \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return (ctx)->fail(LSLAM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                                  \
  } while (0)

// Growable device buffer (never shrinks; sized for 288 GB of HBM, so we keep everything resident).  Growing does NOT
// free the old allocation: kernels already on the stream may still read it, and hipFree would wait for the whole device
// (a first big batch after small ones stalled the stream on it).  Outgrown allocations are kept until release(); growth
// is geometric, so together they are smaller than the live one.  Contents are not carried over.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  std::vector<void*> outgrown;
  DevBuf() = default;
  // owning: a copy would free the same allocations twice (move only)
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap), outgrown(std::move(o.outgrown)) {
    o.p = nullptr;
    o.cap = 0;
    o.outgrown.clear();
  }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      cap = o.cap;
      outgrown = std::move(o.outgrown);
      o.p = nullptr;
      o.cap = 0;
      o.outgrown.clear();
    }
    return *this;
  }
  // Outgrown allocations may only go once nothing on the stream can still read them: callers that have just synchronised
  // their stream (every host-synchronous entry point does, once per call) hand them back here.
  void trim() {
    for (void* q : outgrown) (void)hipFree(q);
    outgrown.clear();
  }
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    T* fresh = nullptr;
    size_t want = n + n / 8 + 64;
    if (want < 2 * cap) want = 2 * cap;  // geometric: the outgrown allocations sum to less than the live one
    hipError_t e = hipMalloc((void**)&fresh, want * sizeof(T));
    if (e != hipSuccess) return e;
    if (p) outgrown.push_back((void*)p);
    p = fresh;
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    for (void* q : outgrown) (void)hipFree(q);
    outgrown.clear();
    p = nullptr;
    cap = 0;
  }
};

// One polite iteration of a host spin loop (the host side of this library builds for x86-64 and AArch64 hosts).
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  asm volatile("" ::: "memory");
#endif
}

// Wait for a ticket a kernel posts in pinned host memory behind its record (single-scan calls: the kernel-completion
// signal and the runtime's wake-up cost more than the last kernel itself).  A bounded busy spin -- it occupies the
// calling host thread's core for the length of the device call, at most `budget_ms` -- with acquire loads; false when
// the ticket did not show up in time (a failed launch, a device fault, a path that posts none): the caller then waits
// for the stream, which also reports the error.
inline bool spin_for_ticket(const int* word, int want, int budget_ms = 20) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) return true;
    cpu_relax();
    if ((spins & 1023u) == 1023u &&
        std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >= budget_ms)
      return false;
  }
}

// ---- phase stamps (diagnostic builds only: -DLSLAM_PHASE_STAMPS, tools/phase_stamps.py) -------------------------------
// Where INSIDE a kernel the time goes: a wave reads the shader-clock counter (s_memtime) at phase boundaries and adds the
// deltas into a per-translation-unit table [kernel][phase] (cycles, visits).  The product build compiles all of it away.
#if defined(LSLAM_PHASE_STAMPS)
constexpr int kStampKernels = 8, kStampSlots = 16384;  // [kernel][wave slot][8 cycle sums | 8 visit counts]
#define LSLAM_STAMP_TABLE(NAME) __device__ unsigned long long NAME##_slots[lslam::kStampKernels][lslam::kStampSlots][16];
struct PhaseClock {  // deltas are summed in registers; ONE flush per wave at the end of the kernel writes them out
  unsigned long long t;
  unsigned long long acc[8];
  unsigned int cnt[8];
  __device__ __forceinline__ PhaseClock() {
#pragma unroll
    for (int i = 0; i < 8; i++) { acc[i] = 0ull; cnt[i] = 0u; }
    t = __builtin_readcyclecounter();
  }
  __device__ __forceinline__ void mark(int i) {
    const unsigned long long n = __builtin_readcyclecounter();
    acc[i] += n - t;
    cnt[i] += 1u;
    t = n;
  }
  // `who`: the one lane that reports for its wave (or block); `slot`: any id that spreads the reporters (wave index) --
  // every reporter adds into its own line of the table, so the flush itself does not queue up behind other waves' flushes
  __device__ __forceinline__ void flush(unsigned long long (*tab)[kStampSlots][16], int k, unsigned slot, bool who) {
    if (!who) return;
    unsigned long long* row = tab[k][slot % (unsigned)kStampSlots];
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (cnt[i]) {
        atomicAdd(&row[i], acc[i]);
        atomicAdd(&row[8 + i], (unsigned long long)cnt[i]);
      }
  }
};
#define LSLAM_PHASE_CLOCK(pc) lslam::PhaseClock pc
#define LSLAM_PHASE_MARK(pc, i) pc.mark(i)
#define LSLAM_PHASE_FLUSH(pc, NAME, k, slot, who) pc.flush(NAME##_slots, k, slot, who)
#else
#define LSLAM_STAMP_TABLE(NAME)
#define LSLAM_PHASE_CLOCK(pc)
#define LSLAM_PHASE_MARK(pc, i)
#define LSLAM_PHASE_FLUSH(pc, NAME, k, slot, who)
#endif

// Launch with optional HIP-event timing on the context stream.
template <typename K, typename... Args>
inline void launch(lslam_context* ctx, const char* name, K kernel, dim3 grid, dim3 block,
                   size_t shmem, Args... args) {
  if (ctx->timer.enabled && (ctx->timer.only.empty() || ctx->timer.only == name)) {
    hipEvent_t e0 = ctx->timer.get(), e1 = ctx->timer.get();
    (void)hipEventRecord(e0, ctx->stream);
    hipLaunchKernelGGL(kernel, grid, block, shmem, ctx->stream, args...);
    (void)hipEventRecord(e1, ctx->stream);
    ctx->timer.pending.push_back({name, e0, e1});
  } else {
    hipLaunchKernelGGL(kernel, grid, block, shmem, ctx->stream, args...);
  }
}

}  // namespace lslam
