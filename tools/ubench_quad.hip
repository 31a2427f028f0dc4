// tools/ubench_quad.hip -- issue cost of the radix-4 butterfly of ntt_pass_l9_kernel out of registers (no LDS, no
// global memory in the loop): how many SIMD cycles do the multiplier chains, the limb add/sub and the normalise / clamp
// steps cost when nothing else is in the way?  (profiles/r02_ubench_quad.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "field_dev.h"
using namespace lcpc;
#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e__), __LINE__); exit(1); } } while (0)
#define ITERS 256
enum { M_MUL = 1, M_ADDSUB = 2, M_NORM = 4, M_CLAMP = 8 };

template <int MODE>
__global__ void __launch_bounds__(256, 4) quad_kernel(const u32* in, const u32* qp_g, u32* out) {
  __shared__ u32 qp[64 * 12];
  for (u32 i = threadIdx.x; i < 64 * 12; i += 256) qp[i] = qp_g[i];
  __syncthreads();
  const u32 tid = blockIdx.x * 256 + threadIdx.x;
  L9 x0, x1, x2, x3;
  Fe29 w0, w1, w2;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    x0.v[k] = in[tid * 9 + k] & l9::M; x1.v[k] = in[(tid + 1) * 9 + k] & l9::M; x2.v[k] = in[(tid + 2) * 9 + k] & l9::M; x3.v[k] = in[(tid + 3) * 9 + k] & l9::M;
    w0.v[k] = in[(tid + 4) * 9 + k] & l9::M; w1.v[k] = in[(tid + 5) * 9 + k] & l9::M; w2.v[k] = in[(tid + 6) * 9 + k] & l9::M;
  }
  x0.v[8] &= 0xfffff; x1.v[8] &= 0xfffff; x2.v[8] &= 0xfffff; x3.v[8] &= 0xfffff; w0.v[8] &= 0x3fffff; w1.v[8] &= 0x3fffff; w2.v[8] &= 0x3fffff;
  for (int it = 0; it < ITERS; it++) {
    L9 b0, b1, c0, d1, e0, e1;
    if constexpr (MODE & M_ADDSUB) {
      b0 = l9::add(x0, x2); b1 = l9::add(x1, x3); c0 = l9::add(b0, b1); d1 = l9::sub(b0, b1); e0 = l9::sub(x0, x2); e1 = l9::sub(x1, x3);
    } else { c0 = x0; d1 = x1; e0 = x2; e1 = x3; }
    if constexpr (MODE & M_NORM) l9::normalize(c0);
    if constexpr (MODE & M_CLAMP) l9::clamp(c0, qp);
    L9 c1, b2, b3, c2, c3;
    if constexpr (MODE & M_MUL) { c1 = l9::mul(d1, w2); b2 = l9::mul(e0, w0); b3 = l9::mul(e1, w1); }
    else { c1 = d1; b2 = e0; b3 = e1; }
    if constexpr (MODE & M_ADDSUB) { c2 = l9::add(b2, b3); c3 = l9::sub(b2, b3); } else { c2 = b2; c3 = b3; }
    if constexpr (MODE & M_NORM) l9::normalize(c2);
    if constexpr (MODE & M_MUL) c3 = l9::mul(c3, w2);
    if constexpr (!(MODE & M_NORM)) {      // keep limbs bounded without the carry pass (1 op per limb instead of 3)
#pragma unroll
      for (int k = 0; k < 9; k++) { c0.v[k] &= l9::M; c2.v[k] &= l9::M; }
    }
    x0 = c0; x1 = c1; x2 = c2; x3 = c3;
  }
  u32 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) acc ^= x0.v[k] ^ x1.v[k] ^ x2.v[k] ^ x3.v[k];
  out[tid] = acc;
}

template <int MODE> static void run(const char* name, const u32* in, const u32* qp, u32* out, int n_instr_hint) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = 256 * 4;       // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipLaunchKernelGGL(quad_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, in, qp, out);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(quad_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, in, qp, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  // per SIMD: 4 waves x ITERS quads
  const double ns_per_wave_quad = best * 1e6 / (4.0 * ITERS);
  printf("%-44s %8.3f ms   %8.1f ns per wave-quad per SIMD  (= %6.0f cycles @2.4 GHz, %6.0f @2.1 GHz)\n", name, best, ns_per_wave_quad,
         ns_per_wave_quad * 2.4, ns_per_wave_quad * 2.1);
}

int main() {
  const size_t n = (size_t)256 * 4 * 256 + 16;
  u32 *in, *qp, *out;
  CHECK(hipMalloc(&in, n * 9 * 4)); CHECK(hipMalloc(&qp, 64 * 12 * 4)); CHECK(hipMalloc(&out, n * 4));
  CHECK(hipMemset(in, 0x5a, n * 9 * 4)); CHECK(hipMemset(qp, 0x11, 64 * 12 * 4));
  run<M_MUL | M_ADDSUB | M_NORM | M_CLAMP>("full quad (4 mul, 8 add/sub, 2 norm, clamp)", in, qp, out, 946);
  run<M_MUL | M_ADDSUB | M_NORM>("no clamp", in, qp, out, 0);
  run<M_MUL | M_ADDSUB>("no clamp, no normalize", in, qp, out, 0);
  run<M_MUL>("4 multiplies only", in, qp, out, 0);
  run<M_ADDSUB | M_NORM | M_CLAMP>("no multiplies", in, qp, out, 0);
  run<M_ADDSUB>("add/sub only", in, qp, out, 0);
  return 0;
}
