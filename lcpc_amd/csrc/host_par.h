// lcpc_amd/csrc/host_par.h -- fork-join helper for the host-side glue (the reference uses rayon at the same places:
// lcpc-2d/src/lib.rs:923-944, lcpc-brakedown-pc/src/matgen.rs:38-49).  Like rayon's global pool, the workers are created
// once (lazily) and reused: spawning 15 threads per call costs ~0.8 ms, more than the whole verify of a small proof.
#pragma once
#include <stdint.h>
#include <atomic>
#include <new>

namespace lcpc {

// host cores this process may really use: hardware threads capped by the cgroup CPU quota (a container can show 256
// hardware threads and be granted 16 CPUs of time; more threads than that only get throttled)
unsigned usable_cores();

// one fork-join region: chunks [0, n_chunks) are claimed one at a time by the calling thread and by up to max_workers pool
// threads.  Lives on the caller's stack; par_run returns once no pool thread can touch it any more.
struct ParJob {
  void (*run)(void* ctx, uint64_t chunk) = nullptr;
  void* ctx = nullptr;
  uint64_t n_chunks = 0;
  unsigned max_workers = 0;                     // pool threads beside the caller
  std::atomic<uint64_t> next{0};
  std::atomic<unsigned> attached{0};
  std::atomic<bool> failed{false};              // a chunk threw (std::bad_alloc): rethrown by the caller
};
void par_run(ParJob& job);                      // encoding.cpp

template <typename Fn> void parallel_for(uint64_t n, uint64_t grain, Fn fn, unsigned max_threads = 16) {
  unsigned nt = usable_cores();
  if (nt > max_threads) nt = max_threads;
  if (nt <= 1 || n < 2 * grain) { fn((uint64_t)0, n); return; }
  const uint64_t nchunks = (n + grain - 1) / grain;
  if (nt > nchunks) nt = (unsigned)nchunks;
  struct Ctx { Fn* fn; uint64_t n, grain; } ctx{&fn, n, grain};
  ParJob job;
  job.run = [](void* c, uint64_t chunk) {
    Ctx* x = static_cast<Ctx*>(c);
    const uint64_t b = chunk * x->grain, e = b + x->grain < x->n ? b + x->grain : x->n;
    (*x->fn)(b, e);
  };
  job.ctx = &ctx;
  job.n_chunks = nchunks;
  job.max_workers = nt - 1;
  par_run(job);
  if (job.failed.load()) throw std::bad_alloc();
}

}  // namespace lcpc
