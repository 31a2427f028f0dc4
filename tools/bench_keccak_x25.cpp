#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#define LCPC_AVX512VL __attribute__((target("avx512f,avx512vl")))
#include "keccak_x25_gen.h"
#include "host_crypto.h"
LCPC_AVX512VL void keccak_x25(uint64_t a[25]) {
  __m128i s[25];
  for (int i = 0; i < 25; i++) s[i] = _mm_cvtsi64_si128((long long)a[i]);
  keccak_x25_rounds(s);
  for (int i = 0; i < 25; i++) a[i] = (uint64_t)_mm_cvtsi128_si64(s[i]);
}
int main() {
  uint64_t a[25], b[25];
  for (int i = 0; i < 25; i++) a[i] = b[i] = i * 0x9e3779b97f4a7c15ull + (i << 7);
  for (int r = 0; r < 1000; r++) { keccak_x25(a); lcpc::keccak_f1600_portable(b); if (memcmp(a, b, 200)) { printf("MISMATCH at %d\n", r); return 1; } }
  printf("x25 == scalar\n");
  auto t0 = std::chrono::steady_clock::now();
  const int N = 300000;
  for (int i = 0; i < N; i++) keccak_x25(a);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("keccak x25: %.1f ns/perm (%llx)\n", dt / N * 1e9, (unsigned long long)a[1]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) lcpc::keccak_f1600_portable(a);
  dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("keccak scalar: %.1f ns/perm (%llx)\n", dt / N * 1e9, (unsigned long long)a[1]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) lcpc::keccak_f1600(a);
  dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("keccak planes(zmm): %.1f ns/perm (%llx)\n", dt / N * 1e9, (unsigned long long)a[1]);
}
