// tests/native/host_par_stress.cpp -- stress of the fork-join pool behind parallel_for (lcpc_amd/csrc/host_par.h, encoding.cpp):
// many regions from many caller threads at once, nested regions, empty and single-chunk regions, a throwing chunk.
// Built and run by tests/test_host_pool.py (g++, no GPU).
#include "host_par.h"
#include <stdio.h>
#include <atomic>
#include <numeric>
#include <thread>
#include <vector>
using namespace lcpc;

static int fail(const char* what) { printf("FAIL %s\n", what); return 1; }

int main() {
  // 1. sums over ranges of many sizes and grains, from 6 caller threads at the same time
  std::atomic<int> bad{0};
  std::vector<std::thread> callers;
  for (int t = 0; t < 6; t++)
    callers.emplace_back([&, t] {
      for (int rep = 0; rep < 300; rep++) {
        const uint64_t n = (uint64_t)((rep * 7919 + t * 104729) % 50000);
        const uint64_t grain = 1 + (uint64_t)((rep * 31 + t) % 700);
        std::vector<uint32_t> hit(n, 0);
        std::atomic<uint64_t> sum{0};
        parallel_for(n, grain, [&](uint64_t b, uint64_t e) {
          uint64_t s = 0;
          for (uint64_t i = b; i < e; i++) { hit[i]++; s += i; }
          sum.fetch_add(s);
        });
        if (sum.load() != (n ? n * (n - 1) / 2 : 0)) bad++;
        for (uint64_t i = 0; i < n; i++) if (hit[i] != 1) { bad++; break; }
      }
    });
  for (auto& c : callers) c.join();
  if (bad.load()) return fail("concurrent regions");
  // 2. nested regions (a chunk of an outer region opens an inner one): the callers work too, so this cannot deadlock
  std::atomic<uint64_t> total{0};
  parallel_for(64, 1, [&](uint64_t b, uint64_t e) {
    for (uint64_t i = b; i < e; i++)
      parallel_for(1000, 10, [&](uint64_t x, uint64_t y) { total.fetch_add(y - x); });
  });
  if (total.load() != 64 * 1000) return fail("nested regions");
  // 3. a chunk that throws std::bad_alloc surfaces in the caller, the pool stays usable
  bool thrown = false;
  try {
    parallel_for(1000, 1, [&](uint64_t b, uint64_t) { if (b == 500) throw std::bad_alloc(); });
  } catch (const std::bad_alloc&) { thrown = true; }
  if (!thrown) return fail("exception not propagated");
  std::atomic<uint64_t> again{0};
  parallel_for(10000, 16, [&](uint64_t b, uint64_t e) { again.fetch_add(e - b); });
  if (again.load() != 10000) return fail("pool unusable after an exception");
  printf("ok (usable cores: %u)\n", usable_cores());
  return 0;
}
