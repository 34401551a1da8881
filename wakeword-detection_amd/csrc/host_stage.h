// host_stage.h - the host-only half of the evaluation flows' input path: writing a chunk ("these sample runs of these clips at
// these offsets, zeros in between") into a page-locked buffer with a pool of copy threads.  Plain C++ (no HIP call, no kernel):
// csrc/uploader.hip includes it, and tests/native/host_stage_check.cpp compiles it alone under ThreadSanitizer / AddressSanitizer
// on the CPU (tests/test_host_logic.py) - the GPU boxes of this pool run no sanitizers.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// The destination is a page-locked upload buffer the CPU never reads back, so whole 16-byte lines go out as streaming
// (non-temporal) stores - no read-for-ownership of the destination (a third of the staging traffic) and the clips' cache lines
// stay in the cache for nobody.  Heads and tails up to the next 16-byte boundary are ordinary stores.
#ifndef WW_STAGE_NT_MIN
#define WW_STAGE_NT_MIN 256  // bytes from which a run goes out as streaming stores
#endif
typedef long long stage_v2 __attribute__((vector_size(16), aligned(16)));
typedef long long stage_v2u __attribute__((vector_size(16), aligned(1)));
inline void stage_copy(char *d, const char *s, size_t n) {
  if (n < WW_STAGE_NT_MIN) { memcpy(d, s, n); return; }
  const size_t head = (16 - ((uintptr_t)d & 15)) & 15;
  memcpy(d, s, head);
  d += head; s += head; n -= head;
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const stage_v2 a = *(const stage_v2u *)(s + i), b = *(const stage_v2u *)(s + i + 16), c = *(const stage_v2u *)(s + i + 32),
                   e = *(const stage_v2u *)(s + i + 48);
    __builtin_nontemporal_store(a, (stage_v2 *)(d + i));
    __builtin_nontemporal_store(b, (stage_v2 *)(d + i + 16));
    __builtin_nontemporal_store(c, (stage_v2 *)(d + i + 32));
    __builtin_nontemporal_store(e, (stage_v2 *)(d + i + 48));
  }
  memcpy(d + i, s + i, n - i);
}
inline void stage_fill0(char *d, size_t n) {
  if (n < WW_STAGE_NT_MIN) { memset(d, 0, n); return; }
  const size_t head = (16 - ((uintptr_t)d & 15)) & 15;
  memset(d, 0, head);
  d += head; n -= head;
  const stage_v2 z = {0, 0};
  size_t i = 0;
  for (; i + 16 <= n; i += 16) __builtin_nontemporal_store(z, (stage_v2 *)(d + i));
  memset(d + i, 0, n - i);
}

struct stage_runs {
  int64_t n = 0;
  const int64_t *dst_off = nullptr;
  const int16_t *const *src = nullptr;
  const int64_t *count = nullptr;
};

// ascending, disjoint, inside [0, total)
bool runs_valid(const stage_runs &r, int64_t total) {
  int64_t end = 0;
  for (int64_t j = 0; j < r.n; ++j) {
    if (r.count[j] < 0 || r.dst_off[j] < end || r.dst_off[j] + r.count[j] > total || (r.count[j] > 0 && !r.src[j])) return false;
    end = r.dst_off[j] + r.count[j];
  }
  return true;
}

// dst[lo, hi) := the parts of the runs inside it, zeros elsewhere
void stage_range(int16_t *dst, const stage_runs &r, int64_t lo, int64_t hi) {
  int64_t a = 0, b = r.n;  // first run that ends after lo
  while (a < b) {
    const int64_t m = (a + b) >> 1;
    if (r.dst_off[m] + r.count[m] <= lo) a = m + 1; else b = m;
  }
  int64_t cur = lo;
  for (int64_t j = a; j < r.n && r.dst_off[j] < hi; ++j) {
    const int64_t s = r.dst_off[j] > cur ? r.dst_off[j] : cur, e = r.dst_off[j] + r.count[j] < hi ? r.dst_off[j] + r.count[j] : hi;
    if (s > cur) stage_fill0((char *)(dst + cur), (size_t)(s - cur) * 2);
    if (e > s) stage_copy((char *)(dst + s), (const char *)(r.src[j] + (s - r.dst_off[j])), (size_t)(e - s) * 2);
    cur = e > cur ? e : cur;
  }
  if (hi > cur) stage_fill0((char *)(dst + cur), (size_t)(hi - cur) * 2);
  std::atomic_thread_fence(std::memory_order_seq_cst);  // the streaming stores are globally visible before the thread reports back
}

int clamp_threads(int threads, int64_t span) {
  const int nt = threads < 1 ? 1 : threads > 64 ? 64 : threads;
  return span < (int64_t)(1 << 20) ? 1 : nt;  // a thread costs more than half a megasample of memcpy
}

// slice t of nt over [lo, hi), 64-sample aligned
inline void slice_of(int64_t lo_all, int64_t hi_all, int nt, int t, int64_t *lo, int64_t *hi) {
  const int64_t step = (((hi_all - lo_all) + nt - 1) / nt + 63) & ~(int64_t)63;
  *lo = lo_all + (int64_t)t * step;
  *hi = *lo + step < hi_all ? *lo + step : hi_all;
}

// A bounded spin in front of every sleep of the uploader's threads.  A rank of eight stages six chunks in 3 ms: waking a thread
// through a condition variable costs 30-60 us, and a chunk crosses four such hand-offs (caller -> worker -> copy threads -> worker
// -> caller), i.e. up to a millisecond of wake-ups on the path of a 3 ms job (round 6).  Each waiter polls its word for at most
// a bounded number of pause instructions (~40 ns each: 60-300 us by the waiter's role, no system call) and only then sleeps; the
// sleeping path is the old one, so nothing is ever missed.
template <typename F>
static inline bool spin_until(F &&ready, int rounds) {
  for (int i = 0; i < rounds; ++i) {
    if (ready()) return true;
    __builtin_ia32_pause();
  }
  return ready();
}

// The uploader's copy threads: started once, woken per chunk (a std::thread per slice and chunk costs as much as a small chunk).
class copy_pool {
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable go_, done_;
  std::function<void(int)> job_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<int> left_{0};
  std::atomic<bool> quit_{false};

  void loop(int t) {
    uint64_t seen = 0;
    for (;;) {
      if (!spin_until([&] { return quit_.load(std::memory_order_acquire) || gen_.load(std::memory_order_acquire) != seen; }, 3000)) {
        std::unique_lock<std::mutex> lk(m_);
        go_.wait(lk, [&] { return quit_.load() || gen_.load() != seen; });
      }
      if (quit_.load(std::memory_order_acquire)) return;
      seen = gen_.load(std::memory_order_acquire);
      job_(t);  // (run() leaves job_ alone until every thread has reported back)
      if (left_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> lk(m_);  // (the waiter checks left_ under this lock before it sleeps)
        done_.notify_all();
      }
    }
  }

 public:
  explicit copy_pool(int n) {
    for (int t = 0; t < n; ++t) th_.emplace_back([this, t] { loop(t); });
  }
  int size() const { return (int)th_.size(); }
  // f(t) for t = 0 .. size() - 1, one per thread; returns when all have finished
  void run(const std::function<void(int)> &f) {
    job_ = f;
    left_.store((int)th_.size(), std::memory_order_release);
    {
      std::lock_guard<std::mutex> lk(m_);
      gen_.fetch_add(1, std::memory_order_release);
    }
    go_.notify_all();
    if (spin_until([&] { return left_.load(std::memory_order_acquire) == 0; }, 6000)) return;
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return left_.load() == 0; });
  }
  ~copy_pool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      quit_.store(true, std::memory_order_release);
    }
    go_.notify_all();
    for (auto &t : th_) t.join();
  }
};

}  // namespace
