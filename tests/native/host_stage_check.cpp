// CPU harness for csrc/host_stage.h (the uploader's staging code: copy threads woken per chunk, a bounded spin in front of every
// sleep, streaming stores into the destination).  Built twice by tests/test_host_logic.py - -fsanitize=thread and
// -fsanitize=address,undefined - and run: random chunks ("runs of clips at offsets, zeros in between") staged by the pool in
// slices, as ww_uploader's worker does, compared with a plain loop; thousands of hand-offs with and without pauses between
// them (so that both the spinning and the sleeping paths are taken).  Exit code 0 = every chunk right, no report.
#include "host_stage.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 400;
  std::mt19937_64 rng(7);
  copy_pool pool(6);
  std::vector<int16_t> clips(1 << 20);
  for (size_t i = 0; i < clips.size(); ++i) clips[i] = (int16_t)(rng() & 0x7fff) - 16384;
  long long checked = 0;
  for (int r = 0; r < rounds; ++r) {
    // a chunk: up to 40 runs, ascending and disjoint, gaps of zeros between them
    const int n_runs = (int)(rng() % 40);
    std::vector<int64_t> dst_off, count;
    std::vector<const int16_t *> src;
    int64_t pos = (int64_t)(rng() % 3000);
    for (int j = 0; j < n_runs; ++j) {
      const int64_t c = (int64_t)(rng() % 60000) + (rng() % 5 == 0 ? 0 : 1);
      dst_off.push_back(pos);
      count.push_back(c);
      src.push_back(clips.data() + rng() % (clips.size() - 60001));
      pos += c + (int64_t)(rng() % 20000);
    }
    const int64_t total = pos + (int64_t)(rng() % 1000);
    const stage_runs runs = {(int64_t)n_runs, dst_off.data(), src.data(), count.data()};
    if (!runs_valid(runs, total)) { fprintf(stderr, "round %d: a valid chunk was refused\n", r); return 2; }
    std::vector<int16_t> got((size_t)total + 64, 77), want((size_t)total + 64, 77);
    int16_t *dst = got.data() + 32;  // (an unaligned destination now and then)
    if (r & 1) dst += 3;
    // the worker's loop: slices, every slice written by all threads
    const int64_t n_slices = 1 + (int64_t)(rng() % 5);
    const int64_t step = ((total + n_slices - 1) / n_slices + 63) & ~(int64_t)63;
    for (int64_t lo_s = 0; lo_s < total; lo_s += step) {
      const int64_t hi_s = lo_s + step < total ? lo_s + step : total;
      const int nt = (r % 7 == 0) ? 1 : pool.size();
      if (nt == 1) {
        stage_range(dst, runs, lo_s, hi_s);
      } else {
        pool.run([&, dst, lo_s, hi_s, nt](int t) {
          if (t >= nt) return;
          int64_t lo, hi;
          slice_of(lo_s, hi_s, nt, t, &lo, &hi);
          if (lo < hi) stage_range(dst, runs, lo, hi);
        });
      }
    }
    int16_t *w = want.data() + (dst - got.data());
    for (int64_t i = 0; i < total; ++i) w[i] = 0;
    for (int j = 0; j < n_runs; ++j) memcpy(w + dst_off[j], src[j], (size_t)count[j] * 2);
    if (memcmp(got.data(), want.data(), got.size() * 2) != 0) { fprintf(stderr, "round %d: staged chunk differs\n", r); return 3; }
    checked += total;
    if (r % 9 == 0) std::this_thread::sleep_for(std::chrono::microseconds(300 + rng() % 700));  // the pool's threads go to sleep
  }
  // bad chunks are refused
  {
    const int64_t off[2] = {10, 5}, cnt[2] = {3, 3};
    const int16_t *sp[2] = {clips.data(), clips.data()};
    const stage_runs bad = {2, off, sp, cnt};
    if (runs_valid(bad, 100)) { fprintf(stderr, "descending runs accepted\n"); return 4; }
  }
  printf("ok %lld samples\n", checked);
  return 0;
}
