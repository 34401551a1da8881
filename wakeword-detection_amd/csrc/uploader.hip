// uploader.hip - the host end of the evaluation flows' input path (no kernels in this file).
//
// The reference's evaluator reads one wav after the other and feeds it 20 ms at a time (utils/evaluate_models.py:52-88); here a
// rank's share of a test split travels to the GPU in chunks of minutes of audio.  A chunk is "these sample runs of these clips
// at these offsets, zeros in between" (the 0.5 s paddings around every clip, the 100 ms gaps of the joined negative stream):
//   * ww_host_stage_i16 writes such a chunk into a caller's page-locked buffer with a handful of host threads, once;
//   * ww_uploader does the same on a thread of its own and sends the chunk on through a copy stream - the caller (one Python
//     thread: wwhip/evaluate.py) plans the next chunk and launches the previous one meanwhile, and never holds a lock the
//     copy needs.
#include "common.h"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <thread>

#include "host_stage.h"


struct ww_uploader {
  int device = 0;
  hipStream_t copy_stream = nullptr;
  struct slot_t {
    void *pin = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool busy = false;  // an upload out of the slot has been enqueued and not yet waited for
  };
  std::vector<slot_t> slots;
  unsigned next_slot = 0;
  size_t largest = 0;  // bytes of the largest chunk so far
  struct job_t {
    int64_t ticket = 0, total = 0;
    std::vector<int64_t> dst_off, count, meta;
    std::vector<const int16_t *> src;
    int16_t *d_pcm = nullptr;
    int64_t *d_meta = nullptr;
  };
  struct result_t {
    int rc = WW_OK;
    int slot = -1;
    std::string what;
  };
  std::deque<job_t> queue;
  std::map<int64_t, result_t> results;  // finished tickets nobody has waited for yet
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  int64_t next_ticket = 1, done_ticket = 0;
  std::atomic<int64_t> done_seen{0}, submitted{0};  // done_ticket / the last ticket queued, for the waiters' bounded spins
  bool quit = false;
  std::thread worker;
  copy_pool *pool = nullptr;

  result_t process(job_t &j) {
    result_t r;
    auto fail = [&](int rc, const char *what, hipError_t e) {
      r.rc = rc;
      r.what = std::string(what) + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string());
      return r;
    };
    const stage_runs runs = {(int64_t)j.dst_off.size(), j.dst_off.data(), j.src.data(), j.count.data()};
    if (!runs_valid(runs, j.total)) return fail(WW_EINVAL, "ww_uploader: sample runs overlap, are out of order or leave the chunk", hipSuccess);
    const int si = (int)(next_slot++ % slots.size());
    slot_t &s = slots[si];
    hipError_t e;
    if (s.busy) {  // the upload that last used the slot must have left it
      if ((e = hipEventSynchronize(s.ev)) != hipSuccess) return fail(WW_EHIP, "ww_uploader: hipEventSynchronize", e);
      s.busy = false;
    }
    const size_t b_pcm = ((size_t)j.total * 2 + 63) & ~(size_t)63, b_meta = j.meta.size() * 8;
    // a slot that has to grow grows to the largest chunk ANY slot has seen: chunk sizes differ and the slots go round, so
    // sizing each slot for its own history alone re-allocated page-locked memory (milliseconds) in call after call
    if (b_pcm + b_meta + 64 > largest) largest = b_pcm + b_meta + 64;
    if (b_pcm + b_meta + 64 > s.cap) {
      if (s.pin) hipHostFree(s.pin);
      s.pin = nullptr;
      s.cap = 0;
      const size_t want = largest + largest / 8;
      if ((e = hipHostMalloc(&s.pin, want, hipHostMallocDefault)) != hipSuccess) {
        s.pin = nullptr;
        return fail(WW_ENOMEM, "ww_uploader: cannot allocate a page-locked slot", e);
      }
      s.cap = want;
    }
    int16_t *dst = (int16_t *)s.pin;
    if (b_meta) memcpy((char *)s.pin + b_pcm, j.meta.data(), b_meta);
    // The chunk goes up in slices, each sent as soon as it has been written: a chunk's way to the device takes the longer of
    // staging and upload, not their sum (round 6: a rank of eight is as much transfer-bound as compute-bound, and the GPU sat
    // idle for 0.35 ms behind the first chunk waiting for the second - profiles/r06/share_of_8_pass_anatomy.txt).  At most eight
    // slices of at least a megasample; every slice is written by all copy threads.
    const int64_t n_slices = j.total >= ((int64_t)3 << 20) ? (j.total >> 20 < 8 ? j.total >> 20 : 8) : 1;
    const int64_t step = ((j.total + n_slices - 1) / n_slices + 63) & ~(int64_t)63;
    for (int64_t lo_s = 0; lo_s < j.total; lo_s += step) {
      const int64_t hi_s = lo_s + step < j.total ? lo_s + step : j.total;
      const int nt = clamp_threads(pool->size(), hi_s - lo_s);
      if (nt == 1) {
        stage_range(dst, runs, lo_s, hi_s);
      } else {
        pool->run([&, dst, lo_s, hi_s, nt](int t) {
          if (t >= nt) return;
          int64_t lo, hi;
          slice_of(lo_s, hi_s, nt, t, &lo, &hi);
          if (lo < hi) stage_range(dst, runs, lo, hi);
        });
      }
      if ((e = hipMemcpyAsync(j.d_pcm + lo_s, dst + lo_s, (size_t)(hi_s - lo_s) * 2, hipMemcpyHostToDevice, copy_stream)) != hipSuccess) {
        (void)hipStreamSynchronize(copy_stream);  // (earlier slices may be in flight out of this slot)
        return fail(WW_EHIP, "ww_uploader: hipMemcpyAsync (samples)", e);
      }
    }
    // from here on a copy out of the slot may be in flight: a failure must not hand the slot back (the next chunk that lands
    // on it would rewrite - or free - page-locked memory under that DMA) before the copy stream has drained
    if (b_meta && (e = hipMemcpyAsync(j.d_meta, (char *)s.pin + b_pcm, b_meta, hipMemcpyHostToDevice, copy_stream)) != hipSuccess) {
      (void)hipStreamSynchronize(copy_stream);
      return fail(WW_EHIP, "ww_uploader: hipMemcpyAsync (tables)", e);
    }
    if ((e = hipEventRecord(s.ev, copy_stream)) != hipSuccess) {
      (void)hipStreamSynchronize(copy_stream);
      return fail(WW_EHIP, "ww_uploader: hipEventRecord", e);
    }
    s.busy = true;
    r.slot = si;
    return r;
  }

  // The worker is a thread of the LIBRARY's: nothing may leave it as an exception (std::terminate in the host program).
  // process() and the bookkeeping around it allocate (strings, a map node): whatever they throw becomes a failed ticket, and a
  // ticket whose result could not even be recorded is remembered by number (lost_ticket) so that its waiter gets WW_ENOMEM.
  void loop() {
    const hipError_t dev_err = hipSetDevice(device);
    for (;;) {
      job_t j;
      {
        const int64_t had = done_seen.load(std::memory_order_acquire);
        spin_until([&] { return submitted.load(std::memory_order_acquire) > had; }, 8000);  // (a chunk usually follows its predecessor closely)
        std::unique_lock<std::mutex> lk(m);
        cv_work.wait(lk, [&] { return quit || !queue.empty(); });
        if (queue.empty()) return;  // (quit: the queue is drained first)
        j = std::move(queue.front());
        queue.pop_front();
      }
      result_t r;
      try {
        if (dev_err != hipSuccess) {
          r.rc = WW_EHIP;
          r.what = std::string("ww_uploader: hipSetDevice: ") + hipGetErrorString(dev_err);
        } else {
          r = process(j);
        }
      } catch (...) {
        (void)hipStreamSynchronize(copy_stream);  // (as process()'s own failure paths: nothing may still read a slot)
        r = result_t();
        r.rc = WW_ENOMEM;  // (what is left empty: it would have to allocate)
      }
      {
        std::lock_guard<std::mutex> lk(m);
        try {
          results[j.ticket] = std::move(r);
        } catch (...) {
          lost_ticket = j.ticket;
        }
        done_ticket = j.ticket;
        done_seen.store(j.ticket, std::memory_order_release);
        // results nobody came for (a caller that gave up on its chunks): only the most recent ones are kept
        while (!results.empty() && results.begin()->first + 4096 < j.ticket) results.erase(results.begin());
      }
      cv_done.notify_all();
    }
  }
  int64_t lost_ticket = 0;  // the last ticket whose result could not be stored (out of memory inside the worker)
};

extern "C" {

int ww_host_stage_i16(int16_t *dst, int64_t total, int64_t n_runs, const int64_t *dst_off, const int16_t *const *src,
                      const int64_t *count, int64_t lo_all, int64_t hi_all, int32_t threads) {
  WW_GUARD_BEGIN
  if (total < 0 || n_runs < 0 || (total > 0 && !dst) || (n_runs > 0 && (!dst_off || !src || !count))) return WW_EINVAL;
  if (lo_all < 0 || lo_all > hi_all || hi_all > total) return WW_EINVAL;
  const stage_runs runs = {n_runs, dst_off, src, count};
  if (!runs_valid(runs, total)) return WW_EINVAL;
  if (hi_all == lo_all) return WW_OK;
  const int nt = clamp_threads(threads, hi_all - lo_all);
  if (nt == 1) {
    stage_range(dst, runs, lo_all, hi_all);
    return WW_OK;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t) {
    int64_t lo, hi;
    slice_of(lo_all, hi_all, nt, t, &lo, &hi);
    if (lo >= hi) break;
    pool.emplace_back([=, &runs] { stage_range(dst, runs, lo, hi); });
  }
  for (auto &th : pool) th.join();
  return WW_OK;
  WW_GUARD_END(nullptr)
}

int ww_uploader_create(ww_ctx *ctx, int32_t slots, int32_t copy_threads, ww_uploader **out) {
  WW_GUARD_BEGIN
  if (!ctx || !out) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  *out = nullptr;
  if (slots < 2 || slots > 16) return ww_fail(ctx, WW_EINVAL, "ww_uploader_create: 2 .. 16 slots (got %d)", slots);
  if (copy_threads < 1 || copy_threads > 64) return ww_fail(ctx, WW_EINVAL, "ww_uploader_create: 1 .. 64 copy threads (got %d)", copy_threads);
  WW_ON_DEVICE(ctx, dev_scope);
  ww_uploader *up = new ww_uploader();
  ww_scoped<ww_uploader, ww_uploader_destroy> own(up);  // (stream, events, pool and thread released on every early return)
  up->device = ctx->device;
  hipError_t e = hipStreamCreateWithFlags(&up->copy_stream, hipStreamNonBlocking);
  up->slots.resize((size_t)slots);
  for (auto &s : up->slots)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev, hipEventDisableTiming);
  if (e != hipSuccess) return ww_fail(ctx, WW_EHIP, "ww_uploader_create: %s", hipGetErrorString(e));
  up->pool = new copy_pool(copy_threads);
  up->worker = std::thread([up] { up->loop(); });
  *out = own.release();
  return WW_OK;
  WW_GUARD_END(ctx)
}

int ww_uploader_destroy(ww_uploader *up) {
  WW_GUARD_BEGIN
  if (!up) return WW_OK;
  {
    std::lock_guard<std::mutex> lk(up->m);
    up->quit = true;
  }
  up->cv_work.notify_all();
  if (up->worker.joinable()) up->worker.join();
  delete up->pool;
  ww_device_scope dev_scope(up->device);
  hipStreamSynchronize(up->copy_stream);
  for (auto &s : up->slots) {
    if (s.pin) hipHostFree(s.pin);
    if (s.ev) hipEventDestroy(s.ev);
  }
  hipStreamDestroy(up->copy_stream);
  delete up;
  return WW_OK;
  WW_GUARD_END(nullptr)
}

int ww_uploader_submit(ww_uploader *up, int64_t total, int64_t n_runs, const int64_t *dst_off, const int16_t *const *src,
                       const int64_t *count, int16_t *d_pcm, int64_t n_meta, const int64_t *meta, int64_t *d_meta, int64_t *ticket) {
  WW_GUARD_BEGIN
  if (!up || !ticket) return WW_EINVAL;
  *ticket = 0;
  if (total < 0 || n_runs < 0 || n_meta < 0 || (total > 0 && !d_pcm) || (n_meta > 0 && (!meta || !d_meta)) ||
      (n_runs > 0 && (!dst_off || !src || !count)))
    return WW_EINVAL;
  {  // (the copies of the caller's arrays allocate: WW_GUARD_END turns a std::bad_alloc into WW_ENOMEM)
  ww_uploader::job_t j;
  j.total = total;
  j.dst_off.assign(dst_off, dst_off + n_runs);
  j.count.assign(count, count + n_runs);
  j.src.assign(src, src + n_runs);
  j.meta.assign(meta, meta + n_meta);
  j.d_pcm = d_pcm;
  j.d_meta = d_meta;
  {
    std::lock_guard<std::mutex> lk(up->m);
    if (up->quit) return WW_EINVAL;
    j.ticket = up->next_ticket;
    up->queue.push_back(std::move(j));
    *ticket = up->next_ticket++;
  }
  }
  up->cv_work.notify_one();
  return WW_OK;
  WW_GUARD_END(nullptr)
}

int ww_uploader_poll(ww_uploader *up, int64_t ticket) {
  WW_GUARD_BEGIN
  if (!up || ticket < 1) return WW_EINVAL;
  std::lock_guard<std::mutex> lk(up->m);
  if (ticket >= up->next_ticket) return WW_EINVAL;
  return up->done_ticket >= ticket ? 1 : 0;
  WW_GUARD_END(nullptr)
}

int ww_uploader_wait(ww_uploader *up, int64_t ticket, ww_ctx *ctx) {
  WW_GUARD_BEGIN
  if (!up || !ctx) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (ctx->device != up->device) return ww_fail(ctx, WW_EINVAL, "ww_uploader_wait: the uploader works for device %d, the context for %d", up->device, ctx->device);
  ww_uploader::result_t r;
  spin_until([&] { return up->done_seen.load(std::memory_order_acquire) >= ticket; }, 6000);  // (bounded; the sleeping wait below decides)
  {
    std::unique_lock<std::mutex> lk(up->m);
    if (ticket < 1 || ticket >= up->next_ticket) return ww_fail(ctx, WW_EINVAL, "ww_uploader_wait: no such ticket (%lld)", (long long)ticket);
    up->cv_done.wait(lk, [&] { return up->done_ticket >= ticket; });
    auto it = up->results.find(ticket);
    if (it == up->results.end() && ticket == up->lost_ticket) return ww_fail(ctx, WW_ENOMEM, "ww_uploader: out of memory while chunk %lld was processed", (long long)ticket);
    if (it == up->results.end()) return ww_fail(ctx, WW_EINVAL, "ww_uploader_wait: ticket %lld was waited for before (or left unclaimed for 4,096 chunks)", (long long)ticket);
    r = std::move(it->second);
    up->results.erase(it);
  }
  if (r.rc != WW_OK) return ww_fail(ctx, r.rc, "%s", r.what.c_str());
  WW_ON_DEVICE(ctx, dev_scope);
  // the slot's event: recorded behind this ticket's copies - or, if the slot has gone round since, behind later copies on the
  // same in-order stream (the uploader itself waited for this ticket's before it reused the slot)
  WW_HIP(ctx, hipStreamWaitEvent(ctx->stream, up->slots[(size_t)r.slot].ev, 0));
  return WW_OK;
  WW_GUARD_END(ctx)
}

}  // extern "C"
