// tools/ubench_power.hip -- SUSTAINED issue rate, shader clock and socket power of a few instruction mixes on gfx950:
// the short bursts of ubench_valu.hip (about 1 ms) finish before the power controller reacts; the NTT runs for 10 ms
// per commit, back to back, at the board power limit.  Usage: ubench_power <mad|add|mix|alignbit> <seconds>
// (sample `rocm-smi --showclocks --showpower` meanwhile: tools/ubench_power.sh).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
typedef uint32_t u32; typedef uint64_t u64;
#define ITERS 4096

__global__ void __launch_bounds__(256) k_mad(u32* out, u32 seed) {
  u32 x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
  u64 a0 = x, a1 = y, a2 = x + 1, a3 = y + 1;
  for (int it = 0; it < ITERS; ++it)
    asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                 "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3);
}
__global__ void __launch_bounds__(256) k_add(u32* out, u32 seed) {
  u32 x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
  u32 a0 = x, a1 = y, a2 = x + 1, a3 = y + 1;
  for (int it = 0; it < ITERS; ++it)
    asm volatile("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_xor_b32 %3, %3, %5\n"
                 "v_add_u32 %0, %0, %5\n v_xor_b32 %1, %1, %4\n v_add_u32 %2, %2, %5\n v_xor_b32 %3, %3, %4\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
__global__ void __launch_bounds__(256) k_alignbit(u32* out, u32 seed) {
  u32 x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
  u32 a0 = x, a1 = y, a2 = x + 1, a3 = y + 1;
  for (int it = 0; it < ITERS; ++it)
    asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %5, 12\n v_alignbit_b32 %2, %2, %4, 8\n v_alignbit_b32 %3, %3, %5, 16\n"
                 "v_alignbit_b32 %0, %0, %5, 7\n v_alignbit_b32 %1, %1, %4, 12\n v_alignbit_b32 %2, %2, %5, 8\n v_alignbit_b32 %3, %3, %4, 16\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
// the NTT's mix: two mads for every cheap VOP2 op
__global__ void __launch_bounds__(256) k_mix(u32* out, u32 seed) {
  u32 x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;
  u64 a0 = x, a1 = y;
  u32 b0 = x + 1, b1 = y + 1;
  for (int it = 0; it < ITERS; ++it)
    asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_add_u32 %2, %2, %4\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n"
                 "v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_and_b32 %3, %3, %5\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_sub_u32 %2, %2, %5\n"
                 : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) : "v"(x), "v"(y) : "vcc");
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 ^ a1) ^ b0 ^ b1;
}

// the same mad loop with one multiplicand in an SGPR (what a wave-uniform twiddle or the modulus limbs look like) and with
// 29-bit multiplicands (the limb size the NTT uses): does the operand source / width change the energy per mad?
__global__ void __launch_bounds__(256) k_mad_sgpr(u32* out, u32 seed) {
  u32 x = threadIdx.x * 2654435761u + seed;
  const u32 y = __builtin_amdgcn_readfirstlane(seed * 2654435761u + 12345u);
  u64 a0 = x, a1 = x + 7, a2 = x + 1, a3 = x + 9;
  for (int it = 0; it < ITERS; ++it)
    asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                 "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(y) : "vcc");
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3);
}
__global__ void __launch_bounds__(256) k_mad29(u32* out, u32 seed) {
  u32 x = (threadIdx.x * 2654435761u + seed) & 0x1fffffffu, y = (x ^ 0x9e3779b9u) & 0x1fffffffu;
  u64 a0 = x, a1 = y, a2 = x + 1, a3 = y + 1;
  for (int it = 0; it < ITERS; ++it) {
    asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                 "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");
    if ((it & 3) == 3) { a0 >>= 29; a1 >>= 29; a2 >>= 29; a3 >>= 29; }      // keep the accumulators in the NTT's range
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3);
}

// LDS round trips like the NTT's: one ds_write_b128 + one ds_read_b128 per iteration (32 B per lane), conflict-free
__global__ void __launch_bounds__(256) k_lds(u32* out, u32 seed) {
  __shared__ uint4 buf[256 * 8];
  uint4 v = make_uint4(threadIdx.x + seed, seed, 3, 4);
  u32 acc = 0;
  for (int it = 0; it < ITERS; ++it) {
    const int slot = (it & 7) * 256 + threadIdx.x;
    buf[slot] = v;
    __syncthreads();
    const uint4 r = buf[(it & 7) * 256 + ((threadIdx.x + 64) & 255)];
    acc ^= r.x;
    v.x = r.y + it;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "mad";
  const double secs = argc > 2 ? atof(argv[2]) : 3.0;
  u32* d;
  if (hipMalloc(&d, 256 * 16 * 256 * 4) != hipSuccess) return 1;
  const int blocks = 256 * 4 * 4;          // 4 waves/SIMD, 4 rounds of blocks per launch
  double ops_per_thread = 0;
  auto launch = [&](u32 s) {
    if (!strcmp(which, "mad")) { hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, d, s); ops_per_thread = 8.0 * ITERS; }
    else if (!strcmp(which, "add")) { hipLaunchKernelGGL(k_add, dim3(blocks), dim3(256), 0, 0, d, s); ops_per_thread = 8.0 * ITERS; }
    else if (!strcmp(which, "alignbit")) { hipLaunchKernelGGL(k_alignbit, dim3(blocks), dim3(256), 0, 0, d, s); ops_per_thread = 8.0 * ITERS; }
    else if (!strcmp(which, "mad_sgpr")) { hipLaunchKernelGGL(k_mad_sgpr, dim3(blocks), dim3(256), 0, 0, d, s); ops_per_thread = 8.0 * ITERS; }
    else if (!strcmp(which, "mad29")) { hipLaunchKernelGGL(k_mad29, dim3(blocks), dim3(256), 0, 0, d, s); ops_per_thread = 8.0 * ITERS; }
    else if (!strcmp(which, "lds")) { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, d, s); ops_per_thread = 32.0 * ITERS; }   // bytes
    else { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, d, s); ops_per_thread = 9.0 * ITERS; }
  };
  launch(0);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  double el = 0;
  while (el < secs) {
    for (int i = 0; i < 20; i++) launch((u32)n++);
    hipDeviceSynchronize();
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double lane_ops = (double)n * blocks * 256 * ops_per_thread;
  if (!strcmp(which, "lds")) { printf("lds: %.2f s, %.2f TB/s of LDS traffic (write + read) sustained\n", el, lane_ops / el / 1e12); return 0; }
  printf("%s: %.2f s, %.2f T lane-op/s sustained (%.2f cycles per wave instruction per SIMD at a nominal 2.4 GHz)\n", which, el, lane_ops / el / 1e12,
         2.4e9 * 1024 * 64 / (lane_ops / el));
  return 0;
}
