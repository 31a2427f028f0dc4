// lcpc_amd/csrc/host_par.h -- fork-join helper for the host-side glue (the reference uses rayon at the same places:
// lcpc-2d/src/lib.rs:923-944, lcpc-brakedown-pc/src/matgen.rs:38-49).
#pragma once
#include <stdint.h>
#include <atomic>
#include <thread>
#include <vector>

namespace lcpc {

// host cores this process may really use: hardware threads capped by the cgroup CPU quota (a container can show 256
// hardware threads and be granted 16 CPUs of time; more threads than that only get throttled)
unsigned usable_cores();

template <typename Fn> void parallel_for(uint64_t n, uint64_t grain, Fn fn, unsigned max_threads = 16) {
  unsigned nt = usable_cores();
  if (nt > max_threads) nt = max_threads;
  if (nt <= 1 || n < 2 * grain) { fn((uint64_t)0, n); return; }
  const uint64_t nchunks = (n + grain - 1) / grain;
  if (nt > nchunks) nt = (unsigned)nchunks;
  std::atomic<uint64_t> next{0};
  auto body = [&] {
    for (;;) {
      const uint64_t c = next.fetch_add(1);
      if (c >= nchunks) return;
      const uint64_t b = c * grain, e = b + grain < n ? b + grain : n;
      fn(b, e);
    }
  };
  std::vector<std::thread> th;
  th.reserve(nt);
  try {
    for (unsigned t = 0; t + 1 < nt; t++) th.emplace_back(body);
  } catch (...) {       // thread creation failed: the calling thread (and whatever started) finishes the work
  }
  body();
  for (auto& x : th) x.join();
}

}  // namespace lcpc
