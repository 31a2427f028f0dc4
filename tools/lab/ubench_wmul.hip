// tools/ubench_wmul.hip -- LAB: what a WAVE-UNIFORM twiddle would buy the Ft255 row NTT (LABNOTES Part 0, "shifted multiples").
//   A  today's multiply: l9::mul = r29_mul1s, 153 mads + 35, per-LANE twiddle (36 B per lane and multiply, coalesced)
//   B  x * w as sum_j x_j * W_j with the nine precomputed W_j = balanced(w 2^(29 j) mod p) of a wave-uniform w (81 dwords by
//      scalar loads inside the statement) + one 32-bit quotient: 90 mads + 26  (lcpc_amd/csrc/gen/gen_wmul_asm.py)
// Each thread runs a dependent chain x <- x * w_i of ITERS multiplies (one accumulator chain per wave and multiply, as in K1s);
// occupancy by __launch_bounds__.  Results of B are checked on the host (r == x * prod w_i mod p, done by the caller script with
// Python integers from the dumped values).
// Build: python lcpc_amd/csrc/gen/gen_wmul_asm.py > tools/wmul_gen.h && hipcc --offload-arch=gfx950 -O3 -I lcpc_amd/csrc tools/ubench_wmul.hip -o tools/ubench_wmul
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "field_dev.h"
using namespace lcpc;
#include "wmul_gen.h"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned __int128 u128;
// ---- host bignum mod p (4 x u64) ----
static const uint64_t PP[4] = {0x02a4f20000000001ull, 0xef73c79086595f30ull, 0xfda9df04b9575969ull, 0x663c799b6e4d2900ull};
static bool ge_p(const uint64_t* a) { for (int i = 3; i >= 0; i--) { if (a[i] > PP[i]) return true; if (a[i] < PP[i]) return false; } return true; }
static void sub_p(uint64_t* a) { u128 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - PP[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; } }
static void dbl(uint64_t* a) { uint64_t c = 0; for (int i = 0; i < 4; i++) { uint64_t n = a[i] >> 63; a[i] = (a[i] << 1) | c; c = n; } if (c || ge_p(a)) sub_p(a); }
// 81 dwords of w: t = 9 k + j = limb k of balanced(w 2^(29 j) mod p)
static void wtable(const uint64_t w[4], uint32_t out[81]) {
  uint64_t v[4]; memcpy(v, w, 32);
  uint64_t half[4]; memcpy(half, PP, 32); for (int i = 0; i < 4; i++) half[i] = (PP[i] >> 1) | (i < 3 ? PP[i + 1] << 63 : 0);
  for (int j = 0; j < 9; j++) {
    uint64_t m[5] = {v[0], v[1], v[2], v[3], 0};
    bool big = false; for (int i = 3; i >= 0; i--) { if (v[i] > half[i]) { big = true; break; } if (v[i] < half[i]) break; }
    if (big) { u128 br = 0; for (int i = 0; i < 5; i++) { u128 d = (u128)m[i] - (i < 4 ? PP[i] : 0) - (uint64_t)br; m[i] = (uint64_t)d; br = (d >> 64) & 1; } }   // v - p (two's complement, 320 bits)
    for (int k = 0; k < 9; k++) {
      const int b = 29 * k, wd = b / 64, sh = b % 64;
      uint64_t x = m[wd] >> sh; if (sh > 35) x |= m[wd + 1] << (64 - sh);
      out[9 * k + j] = k < 8 ? (uint32_t)(x & 0x1fffffffu) : (uint32_t)x;       // limb 8: bits 232.. sign-extended into 32 bits
    }
    for (int s = 0; s < 29; s++) dbl(v);
  }
}

template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_permul(const u32* __restrict__ tw, u32 n_tw, u32 iters, u32* out) {
  L9 x;
  for (int i = 0; i < 9; i++) x.v[i] = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x) & 0x0fffffffu;
  const u32 gid = blockIdx.x * 256 + threadIdx.x;
  for (u32 it = 0; it < iters; it++) {
    const u32 e = (gid + it * 8191u) % n_tw;                         // per-lane entries, consecutive lanes consecutive (as the packs are)
    const uint4* p = reinterpret_cast<const uint4*>(tw + (size_t)e * 12);
    const uint4 a = p[0], b = p[1];
    Fe29 w; w.v[0] = a.x; w.v[1] = a.y; w.v[2] = a.z; w.v[3] = a.w; w.v[4] = b.x; w.v[5] = b.y; w.v[6] = b.z; w.v[7] = b.w; w.v[8] = tw[(size_t)e * 12 + 8];
    x = l9::mul(x, w);
  }
  u32 s = 0; for (int i = 0; i < 9; i++) s ^= x.v[i];
  out[gid] = s;
}

typedef u32 __attribute__((address_space(4))) CU32;
template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_wmul(const u32* __restrict__ wt, u32 n_tw, u32 iters, u32* out, u32* dump) {
  u32 x[9], np2[9];
  for (int i = 0; i < 9; i++) x[i] = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x) & 0x0fffffffu;
  // limbs of -2p as signed 32-bit values
  np2[0] = (u32)(-(int)(2 * P29::limb(0)));
  for (int k = 1; k < 9; k++) np2[k] = (u32)(-(int)(2 * P29::limb(k)));
  const u32 gid = blockIdx.x * 256 + threadIdx.x;
  const u32 wave = __builtin_amdgcn_readfirstlane(gid >> 6);
  if (dump && gid < 64) for (int i = 0; i < 9; i++) dump[gid * 18 + i] = x[i];
  for (u32 it = 0; it < iters; it++) {
    const u32 e = __builtin_amdgcn_readfirstlane((wave * 13u + it * 8191u) % n_tw);
    const u32* w = wt + (size_t)e * 81;
    u32 r[9];
    wmul_u(x, np2, w, r);
    for (int i = 0; i < 9; i++) x[i] = r[i];
  }
  if (dump && gid < 64) for (int i = 0; i < 9; i++) dump[gid * 18 + 9 + i] = x[i];
  u32 s = 0; for (int i = 0; i < 9; i++) s ^= x[i];
  out[gid] = s;
}

int main(int argc, char** argv) {
  const u32 n_tw = argc > 1 ? (u32)atoi(argv[1]) : 4096;            // distinct twiddles (4096 x 324 B = 1.3 MB; 131072 = 42 MB)
  const u32 iters = argc > 2 ? (u32)atoi(argv[2]) : 256;
  const u32 blocks = 256 * 16;
  // twiddles: w_e = 3^(e+1)-ish residues: any values < p will do for throughput; the host check uses the same table
  std::vector<uint64_t> ws((size_t)n_tw * 4);
  uint64_t cur[4] = {5, 0, 0, 0};
  for (u32 e = 0; e < n_tw; e++) { for (int s = 0; s < 3; s++) dbl(cur); cur[0] ^= e; if (ge_p(cur)) sub_p(cur); memcpy(&ws[(size_t)e * 4], cur, 32); }
  std::vector<uint32_t> wt((size_t)n_tw * 81), tw((size_t)n_tw * 12, 0);
  for (u32 e = 0; e < n_tw; e++) {
    wtable(&ws[(size_t)e * 4], &wt[(size_t)e * 81]);
    for (int k = 0; k < 9; k++) { const int b = 29 * k, wd = b / 64, sh = b % 64; uint64_t x = ws[(size_t)e * 4 + wd] >> sh; if (sh > 35 && wd < 3) x |= ws[(size_t)e * 4 + wd + 1] << (64 - sh); tw[(size_t)e * 12 + k] = (uint32_t)(x & 0x1fffffffu); }
  }
  u32 *d_wt, *d_tw, *d_out, *d_dump;
  CHECK(hipMalloc(&d_wt, wt.size() * 4)); CHECK(hipMalloc(&d_tw, tw.size() * 4)); CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4)); CHECK(hipMalloc(&d_dump, 64 * 18 * 4));
  CHECK(hipMemcpy(d_wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto report = [&](const char* name, float ms) {
    printf("{\"kernel\": \"%s\", \"n_twiddles\": %u, \"iters\": %u, \"ms\": %.3f}\n", name, n_tw, iters, ms);
  };
  auto run = [&](const char* name, auto launch) -> int {
    launch(); CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    report(name, best);
    return 0;
  };
  run("A per-lane twiddle, Montgomery 153+35, 4 waves/SIMD", [&] { hipLaunchKernelGGL(k_permul<4>, dim3(blocks), dim3(256), 0, 0, d_tw, n_tw, iters, d_out); });
  run("A per-lane twiddle, Montgomery 153+35, 2 waves/SIMD", [&] { hipLaunchKernelGGL(k_permul<2>, dim3(blocks), dim3(256), 0, 0, d_tw, n_tw, iters, d_out); });
  run("B wave-uniform twiddle, shifted multiples 90+26, 4 waves/SIMD", [&] { hipLaunchKernelGGL(k_wmul<4>, dim3(blocks), dim3(256), 0, 0, d_wt, n_tw, iters, d_out, (u32*)nullptr); });
  run("B wave-uniform twiddle, shifted multiples 90+26, 2 waves/SIMD", [&] { hipLaunchKernelGGL(k_wmul<2>, dim3(blocks), dim3(256), 0, 0, d_wt, n_tw, iters, d_out, (u32*)nullptr); });
  // low occupancy for real: the grid itself has only 1 or 2 waves per SIMD (256 CUs x 4 SIMDs)
  run("A, grid of 1 wave per SIMD (256 workgroups)", [&] { hipLaunchKernelGGL(k_permul<4>, dim3(256), dim3(256), 0, 0, d_tw, n_tw, iters, d_out); });
  run("A, grid of 2 waves per SIMD (512 workgroups)", [&] { hipLaunchKernelGGL(k_permul<4>, dim3(512), dim3(256), 0, 0, d_tw, n_tw, iters, d_out); });
  run("B, grid of 1 wave per SIMD (256 workgroups)", [&] { hipLaunchKernelGGL(k_wmul<4>, dim3(256), dim3(256), 0, 0, d_wt, n_tw, iters, d_out, (u32*)nullptr); });
  run("B, grid of 2 waves per SIMD (512 workgroups)", [&] { hipLaunchKernelGGL(k_wmul<4>, dim3(512), dim3(256), 0, 0, d_wt, n_tw, iters, d_out, (u32*)nullptr); });
  run("B, grid of 4 waves per SIMD (1024 workgroups)", [&] { hipLaunchKernelGGL(k_wmul<4>, dim3(1024), dim3(256), 0, 0, d_wt, n_tw, iters, d_out, (u32*)nullptr); });
  // correctness dump: 64 lanes, 3 multiplies
  hipLaunchKernelGGL(k_wmul<4>, dim3(1), dim3(256), 0, 0, d_wt, n_tw, 3u, d_out, d_dump);
  std::vector<uint32_t> dump(64 * 18);
  CHECK(hipMemcpy(dump.data(), d_dump, dump.size() * 4, hipMemcpyDeviceToHost));
  FILE* f = fopen("wmul_dump.txt", "w");
  if (f) {
    for (int it = 0; it < 3; it++) { const u32 e = (0 * 13u + it * 8191u) % n_tw; fprintf(f, "w %016llx%016llx%016llx%016llx\n", (unsigned long long)ws[(size_t)e * 4 + 3], (unsigned long long)ws[(size_t)e * 4 + 2], (unsigned long long)ws[(size_t)e * 4 + 1], (unsigned long long)ws[(size_t)e * 4]); }
    for (int l = 0; l < 64; l++) { fprintf(f, "lane"); for (int i = 0; i < 18; i++) fprintf(f, " %d", (int)dump[l * 18 + i]); fprintf(f, "\n"); }
    fclose(f);
  }
  return 0;
}
