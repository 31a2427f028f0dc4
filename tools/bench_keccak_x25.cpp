#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#define LCPC_AVX512VL __attribute__((target("avx512f,avx512vl")))
#include "keccak_x25_gen.h"
#include "host_crypto.h"
#ifdef KECCAK_VARIANTS          // python lcpc_amd/csrc/gen/gen_keccak_x25.py --variants > /tmp/keccak_variants_gen.h; g++ -DKECCAK_VARIANTS -I/tmp ...
#include "keccak_variants_gen.h"
#endif
LCPC_AVX512VL void keccak_x25(uint64_t a[25]) {
  __m128i s[25];
  for (int i = 0; i < 25; i++) s[i] = _mm_cvtsi64_si128((long long)a[i]);
  keccak_x25_rounds(s);
  for (int i = 0; i < 25; i++) a[i] = (uint64_t)_mm_cvtsi128_si64(s[i]);
}
LCPC_AVX512VL void keccak_x25_asm(uint64_t a[25]) { keccak_x25_permute_tern(a); }
LCPC_AVX512VL void keccak_x25_asm_xor(uint64_t a[25]) { keccak_x25_permute_xor(a); }
int main() {
  uint64_t a[25], b[25];
  for (int i = 0; i < 25; i++) a[i] = b[i] = i * 0x9e3779b97f4a7c15ull + (i << 7);
  for (int r = 0; r < 1000; r++) { keccak_x25(a); lcpc::keccak_f1600_portable(b); if (memcmp(a, b, 200)) { printf("MISMATCH at %d\n", r); return 1; } }
  for (int r = 0; r < 1000; r++) { keccak_x25_asm(a); lcpc::keccak_f1600_portable(b); if (memcmp(a, b, 200)) { printf("asm MISMATCH at %d\n", r); return 1; } }
  printf("x25 (intrinsics) == x25 (asm) == scalar\n");
  auto t0 = std::chrono::steady_clock::now();
  const int N = 300000;
  for (int i = 0; i < N; i++) keccak_x25(a);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("keccak x25 intrinsics: %.1f ns/perm (%llx)\n", dt / N * 1e9, (unsigned long long)a[1]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) keccak_x25_asm(a);
  dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("keccak x25 asm (ternlog mix): %.1f ns/perm (%llx)\n", dt / N * 1e9, (unsigned long long)a[1]);
#ifdef KECCAK_VARIANTS
  {
    typedef void (*Fn)(uint64_t*);
    struct V { const char* name; Fn fn; } vs[] = {{"theta via D (xor)", keccak_v_theta_d}, {"theta via D + parity all xor", keccak_v_theta_d_parx},
                                                  {"theta via D + chi andn/xor in 2 rows", keccak_v_theta_d_chi2}, {"no vpternlogq at all", keccak_v_all_xor},
                                                  {"theta via D in 3 columns", keccak_v_d3}, {"theta via D in 4 columns", keccak_v_d4}, {"chi without copies", keccak_v_fresh},
                                                  {"theta via D + chi without copies", keccak_v_d_fresh}, {"theta via D in 3 columns + chi without copies", keccak_v_d3_fresh},
                                                  {"xor mix, every 2nd rho rotate as vpshldq", keccak_v_shld2}, {"xor mix, every 3rd rho rotate as vpshldq", keccak_v_shld3}, {"xor mix, all rho rotates as vpshldq", keccak_v_shld1}};
    for (auto& v : vs) {
      uint64_t c[25], d[25];
      for (int i = 0; i < 25; i++) c[i] = d[i] = i * 0x9e3779b97f4a7c15ull + (i << 7);
      for (int r = 0; r < 100; r++) { v.fn(c); lcpc::keccak_f1600_portable(d); if (memcmp(c, d, 200)) { printf("%s MISMATCH\n", v.name); return 1; } }
      t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; i++) v.fn(c);
      dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      printf("keccak x25 asm, %s: %.1f ns/perm (%llx)\n", v.name, dt / N * 1e9, (unsigned long long)c[1]);
    }
  }
#endif
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) lcpc::keccak_f1600_portable(a);
  dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("keccak scalar: %.1f ns/perm (%llx)\n", dt / N * 1e9, (unsigned long long)a[1]);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) lcpc::keccak_f1600(a);
  dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("keccak library dispatch: %.1f ns/perm (%llx)\n", dt / N * 1e9, (unsigned long long)a[1]);
}
