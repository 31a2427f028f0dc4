// tools/ubench_select.hip -- cost of "conditional subtract p" (8x32-bit limbs) in three encodings on gfx950:
//   V1: v_subbrev chain, then 8 x v_cndmask_b32 (e32, condition in VCC)            [what hipcc emits]
//   V2: v_subbrev chain, mask = 0 - borrow, then 8 x v_bfi_b32 (bitfield select)
//   V3: v_subbrev chain, then 8 x v_cndmask_b32_e64 with the condition copied to an SGPR pair
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;
#define ITERS 4096
#define CHAIN "v_subrev_co_u32 %8, vcc, 1, %0\n v_subbrev_co_u32 %9, vcc, %17, %1, vcc\n v_subbrev_co_u32 %10, vcc, %18, %2, vcc\n v_subbrev_co_u32 %11, vcc, %19, %3, vcc\n" \
              "v_subbrev_co_u32 %12, vcc, %20, %4, vcc\n v_subbrev_co_u32 %13, vcc, %21, %5, vcc\n v_subbrev_co_u32 %14, vcc, %22, %6, vcc\n v_subbrev_co_u32 %15, vcc, %23, %7, vcc\n"
#define OUTS : "+v"(r0),"+v"(r1),"+v"(r2),"+v"(r3),"+v"(r4),"+v"(r5),"+v"(r6),"+v"(r7), "=&v"(d0),"=&v"(d1),"=&v"(d2),"=&v"(d3),"=&v"(d4),"=&v"(d5),"=&v"(d6),"=&v"(d7)
#define INS : "v"(p1),"v"(p2),"v"(p3),"v"(p4),"v"(p5),"v"(p6),"v"(p7)
#define PRE u32 r0=threadIdx.x*77u+s, r1=r0*3u, r2=r0*5u, r3=r0*7u, r4=r0*11u, r5=r0*13u, r6=r0*17u, r7=(r0*19u)&0x7fffffffu; \
  u32 d0,d1,d2,d3,d4,d5,d6,d7; u32 p1=0x02a4f200u,p2=0x86595f30u,p3=0xef73c790u,p4=0xb9575969u,p5=0xfda9df04u,p6=0x6e4d2900u,p7=0x663c799bu;
#define POST out[blockIdx.x*blockDim.x+threadIdx.x]=r0^r1^r2^r3^r4^r5^r6^r7;
__global__ void __launch_bounds__(256) k_v1(u32* out, u32 s){ PRE u32 mk;
  for(int i=0;i<ITERS;i++){ asm volatile(CHAIN
    "v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %9, %1, vcc\n v_cndmask_b32 %2, %10, %2, vcc\n v_cndmask_b32 %3, %11, %3, vcc\n"
    "v_cndmask_b32 %4, %12, %4, vcc\n v_cndmask_b32 %5, %13, %5, vcc\n v_cndmask_b32 %6, %14, %6, vcc\n v_cndmask_b32 %7, %15, %7, vcc\n v_add_u32 %0, %0, %1\n" OUTS, "=&v"(mk) INS : "vcc"); }
  POST }
__global__ void __launch_bounds__(256) k_v2(u32* out, u32 s){ PRE u32 mk;
  for(int i=0;i<ITERS;i++){ asm volatile(CHAIN
    "v_subb_co_u32 %16, vcc, 0, 0, vcc\n"   // mask = all ones iff the chain borrowed (keep r), else 0 (take d)
    "v_bfi_b32 %0, %16, %0, %8\n v_bfi_b32 %1, %16, %1, %9\n v_bfi_b32 %2, %16, %2, %10\n v_bfi_b32 %3, %16, %3, %11\n"
    "v_bfi_b32 %4, %16, %4, %12\n v_bfi_b32 %5, %16, %5, %13\n v_bfi_b32 %6, %16, %6, %14\n v_bfi_b32 %7, %16, %7, %15\n v_add_u32 %0, %0, %1\n" OUTS, "=&v"(mk) INS : "vcc"); }
  POST }
__global__ void __launch_bounds__(256) k_v3(u32* out, u32 s){ PRE u64 sm;
  for(int i=0;i<ITERS;i++){ asm volatile(CHAIN
    "s_mov_b64 %16, vcc\n"
    "v_cndmask_b32_e64 %0, %8, %0, %16\n v_cndmask_b32_e64 %1, %9, %1, %16\n v_cndmask_b32_e64 %2, %10, %2, %16\n v_cndmask_b32_e64 %3, %11, %3, %16\n"
    "v_cndmask_b32_e64 %4, %12, %4, %16\n v_cndmask_b32_e64 %5, %13, %5, %16\n v_cndmask_b32_e64 %6, %14, %6, %16\n v_cndmask_b32_e64 %7, %15, %7, %16\n v_add_u32 %0, %0, %1\n" OUTS, "=&s"(sm) INS : "vcc"); }
  POST }
template <typename K> void run(K k, const char* n, u32* d){ hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize(); float best=1e9;
  for(int r=0;r<5;r++){ hipEventRecord(a); hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, (u32)r); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms; }
  double waves=1024.0*4*ITERS; printf("%-44s %8.3f ms  %7.1f cycles per conditional-subtract per wave (@2.4GHz, 4 waves/SIMD)\n", n, best, best*1e-3*2.4e9*1024/waves); }
int main(){ u32* d; hipMalloc(&d, 1024*256*4);
  run(k_v1,"V1 subb chain + 8 v_cndmask_b32 (vcc)",d); run(k_v2,"V2 subb chain + mask + 8 v_bfi_b32",d); run(k_v3,"V3 subb chain + s_mov + 8 v_cndmask_e64",d); return 0; }
