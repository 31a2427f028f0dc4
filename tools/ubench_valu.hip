// tools/ubench_valu.hip -- integer-VALU throughput microbenchmark for gfx950.
// Measures the issue rate of the instructions a 256-bit Montgomery multiply can be built from,
// so DESIGN.md can price the NTT against a *measured* integer ceiling (the HBM roofline does not
// bind a 255-bit field).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
#define ITERS 2048
#define UNROLL 8

#define KERNEL(name, DECL, BODY, SINK) \
__global__ void __launch_bounds__(256) name(u32* out, u32 seed){ \
  u32 x=threadIdx.x*2654435761u+seed, y=x^0x9e3779b9u; DECL; \
  for(int it=0; it<ITERS; ++it){ _Pragma("unroll") for(int u=0;u<UNROLL;++u){ BODY; } } \
  out[blockIdx.x*blockDim.x+threadIdx.x]=SINK; }

// 8 independent accumulator chains each
KERNEL(k_mad_u64_u32, u64 a0=x;u64 a1=y;u64 a2=x+1;u64 a3=y+1;u64 a4=x+2;u64 a5=y+2;u64 a6=x+3;u64 a7=y+3,
  asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
               "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x),"v"(y) : "vcc"),
  (u32)(a0^a1^a2^a3^a4^a5^a6^a7))
KERNEL(k_mad_addc, u64 a0=x;u64 a1=y;u64 a2=x+1;u64 a3=y+1; u32 h0=0;u32 h1=0;u32 h2=0;u32 h3=0,
  asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n"
               "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_addc_co_u32 %6, vcc, 0, %6, vcc\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_addc_co_u32 %7, vcc, 0, %7, vcc\n"
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(h0),"+v"(h1),"+v"(h2),"+v"(h3) : "v"(x),"v"(y) : "vcc"),
  (u32)(a0^a1^a2^a3)^h0^h1^h2^h3)
KERNEL(k_mad_dep, u64 a0=x,
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
               "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
     : "+v"(a0) : "v"(x),"v"(y) : "vcc"),
  (u32)(a0))
#define OP8(OPSTR) asm volatile(OPSTR " %0, %8, %9\n" OPSTR " %1, %8, %9\n" OPSTR " %2, %8, %9\n" OPSTR " %3, %8, %9\n" OPSTR " %4, %8, %9\n" OPSTR " %5, %8, %9\n" OPSTR " %6, %8, %9\n" OPSTR " %7, %8, %9\n" \
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x),"v"(y))
#define DECL8 u32 a0=x;u32 a1=y;u32 a2=x+1;u32 a3=y+1;u32 a4=x+2;u32 a5=y+2;u32 a6=x+3;u32 a7=y+3
KERNEL(k_mul_lo_u32, DECL8, OP8("v_mul_lo_u32"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_mul_hi_u32, DECL8, OP8("v_mul_hi_u32"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_mul_u32_u24, DECL8, OP8("v_mul_u32_u24"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_mul_hi_u32_u24, DECL8, OP8("v_mul_hi_u32_u24"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_add_u32, DECL8, OP8("v_add_u32"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_xor_b32, DECL8, OP8("v_xor_b32"), a0^a1^a2^a3^a4^a5^a6^a7)
#define OP8_3(OPSTR) asm volatile(OPSTR " %0, %8, %9, %0\n" OPSTR " %1, %8, %9, %1\n" OPSTR " %2, %8, %9, %2\n" OPSTR " %3, %8, %9, %3\n" OPSTR " %4, %8, %9, %4\n" OPSTR " %5, %8, %9, %5\n" OPSTR " %6, %8, %9, %6\n" OPSTR " %7, %8, %9, %7\n" \
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x),"v"(y))
KERNEL(k_mad_u32_u24, DECL8, OP8_3("v_mad_u32_u24"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_alignbit, DECL8, OP8_3("v_alignbit_b32"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_add3_u32, DECL8, OP8_3("v_add3_u32"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_fma_f32, DECL8, OP8_3("v_fma_f32"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_and_b32, DECL8, OP8("v_and_b32"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_lshrrev_b32, DECL8, asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_bfe_u32, DECL8, asm volatile("v_bfe_u32 %0, %0, 3, 29\n v_bfe_u32 %1, %1, 3, 29\n v_bfe_u32 %2, %2, 3, 29\n v_bfe_u32 %3, %3, 3, 29\n v_bfe_u32 %4, %4, 3, 29\n v_bfe_u32 %5, %5, 3, 29\n v_bfe_u32 %6, %6, 3, 29\n v_bfe_u32 %7, %7, 3, 29\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_lshl_or_b32, DECL8, asm volatile("v_lshl_or_b32 %0, %0, 3, %8\n v_lshl_or_b32 %1, %1, 3, %8\n v_lshl_or_b32 %2, %2, 3, %8\n v_lshl_or_b32 %3, %3, 3, %8\n v_lshl_or_b32 %4, %4, 3, %8\n v_lshl_or_b32 %5, %5, 3, %8\n v_lshl_or_b32 %6, %6, 3, %8\n v_lshl_or_b32 %7, %7, 3, %8\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x)), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_cndmask, DECL8, asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x) : "vcc"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_cndmask_cmp, DECL8, asm volatile("v_cmp_gt_u32 vcc, %8, %0\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x) : "vcc"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_cndmask_e64, DECL8; u64 msk = 0x5555aaaa5555aaaaull, asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x), "s"(msk)), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_sel_arith, DECL8; u32 mk = y, asm volatile("v_xor_b32 %0, %0, %8\n v_and_b32 %0, %0, %9\n v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_and_b32 %1, %1, %9\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_and_b32 %2, %2, %9\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x), "v"(mk)), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_addc_chain, DECL8, asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x) : "vcc"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_lshrrev_b64, u64 a0=x;u64 a1=y;u64 a2=x+1;u64 a3=y+1;u64 a4=x+2;u64 a5=y+2;u64 a6=x+3;u64 a7=y+3,
  asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3\n v_lshrrev_b64 %4, 1, %4\n v_lshrrev_b64 %5, 1, %5\n v_lshrrev_b64 %6, 1, %6\n v_lshrrev_b64 %7, 1, %7\n"
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)),
  (u32)(a0^a1^a2^a3^a4^a5^a6^a7))
KERNEL(k_mad_snop, u64 a0=x;u64 a1=y;u64 a2=x+1;u64 a3=y+1,
  asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n s_nop 0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n s_nop 0\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n s_nop 0\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n s_nop 0\n"
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3) : "v"(x),"v"(y) : "vcc"),
  (u32)(a0^a1^a2^a3))
#define DECL8D double a0=x;double a1=y;double a2=x+1;double a3=y+1;double a4=x+2;double a5=y+2;double a6=x+3;double a7=y+3; double dx=1.0+1e-9*x; double dy=1e-9*y
KERNEL(k_fma_f64, DECL8D,
  asm volatile("v_fma_f64 %0, %8, %0, %9\n v_fma_f64 %1, %8, %1, %9\n v_fma_f64 %2, %8, %2, %9\n v_fma_f64 %3, %8, %3, %9\n v_fma_f64 %4, %8, %4, %9\n v_fma_f64 %5, %8, %5, %9\n v_fma_f64 %6, %8, %6, %9\n v_fma_f64 %7, %8, %7, %9\n"
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(dx),"v"(dy)),
  (u32)(a0+a1+a2+a3+a4+a5+a6+a7))
KERNEL(k_lshl_add_u64, u64 a0=x;u64 a1=y;u64 a2=x+1;u64 a3=y+1;u64 a4=x+2;u64 a5=y+2;u64 a6=x+3;u64 a7=y+3; u64 xx=x,
  asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(xx)),
  (u32)(a0^a1^a2^a3^a4^a5^a6^a7))

// SDWA forms and byte permute (candidates for BLAKE3's rotations by 16 / 8), the signed mad and the 64-bit arithmetic shift
#define SDWA8(OPSTR, MODS) asm volatile(OPSTR " %0, %0, %8 " MODS "\n" OPSTR " %1, %1, %8 " MODS "\n" OPSTR " %2, %2, %8 " MODS "\n" OPSTR " %3, %3, %8 " MODS "\n" OPSTR " %4, %4, %8 " MODS "\n" OPSTR " %5, %5, %8 " MODS "\n" OPSTR " %6, %6, %8 " MODS "\n" OPSTR " %7, %7, %8 " MODS "\n" \
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x),"v"(y))
KERNEL(k_xor_sdwa, DECL8, SDWA8("v_xor_b32_sdwa", "dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_perm_b32, DECL8, OP8_3("v_perm_b32"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_mad_i64_i32, u64 a0=x;u64 a1=y;u64 a2=x+1;u64 a3=y+1;u64 a4=x+2;u64 a5=y+2;u64 a6=x+3;u64 a7=y+3,
  asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_mad_i64_i32 %1, vcc, %8, %9, %1\n v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_mad_i64_i32 %3, vcc, %8, %9, %3\n"
               "v_mad_i64_i32 %4, vcc, %8, %9, %4\n v_mad_i64_i32 %5, vcc, %8, %9, %5\n v_mad_i64_i32 %6, vcc, %8, %9, %6\n v_mad_i64_i32 %7, vcc, %8, %9, %7\n"
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x),"v"(y) : "vcc"),
  (u32)(a0^a1^a2^a3^a4^a5^a6^a7))
KERNEL(k_ashrrev_i64, u64 a0=x;u64 a1=y;u64 a2=x+1;u64 a3=y+1;u64 a4=x+2;u64 a5=y+2;u64 a6=x+3;u64 a7=y+3,
  asm volatile("v_ashrrev_i64 %0, 1, %0\n v_ashrrev_i64 %1, 1, %1\n v_ashrrev_i64 %2, 1, %2\n v_ashrrev_i64 %3, 1, %3\n v_ashrrev_i64 %4, 1, %4\n v_ashrrev_i64 %5, 1, %5\n v_ashrrev_i64 %6, 1, %6\n v_ashrrev_i64 %7, 1, %7\n"
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)),
  (u32)(a0^a1^a2^a3^a4^a5^a6^a7))

// cross-lane moves (what a limb-sliced multiply -- one element spread over 9 lanes -- would pay per limb product)
KERNEL(k_dpp_ror, DECL8, asm volatile("v_mov_b32_dpp %0, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                                      "v_mov_b32_dpp %4, %9 row_ror:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %4 row_ror:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %5 row_ror:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 row_ror:2 row_mask:0xf bank_mask:0xf\n"
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(x),"v"(y)), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_bpermute, DECL8; u32 idx = ((threadIdx.x + 9) & 63) * 4, asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
     : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(idx)), a0^a1^a2^a3^a4^a5^a6^a7)

// round 6: the forms the generated multipliers actually use -- one accumulator (a dependent chain), an SGPR multiplicand, a VOP2 with
// a 32-bit literal (the limb mask), the plain VOP2 / VOP1 ops of the carry passes
KERNEL(k_mad_i64_dep, u64 a0=x,
  asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %2, %1, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %2, %1, %0\n"
               "v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %2, %1, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %2, %1, %0\n"
     : "+v"(a0) : "v"(x),"v"(y) : "vcc"),
  (u32)(a0))
KERNEL(k_mad_i64_sgpr, u64 a0=x; u32 sv = seed * 2654435761u + 12345u,
  asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %3, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %3, %2, %0\n"
               "v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %3, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %3, %2, %0\n"
     : "+v"(a0) : "v"(x),"s"(sv),"v"(y) : "vcc"),
  (u32)(a0))
KERNEL(k_and_lit, DECL8, asm volatile("v_and_b32 %0, 0x1fffffff, %0\n v_and_b32 %1, 0x1fffffff, %1\n v_and_b32 %2, 0x1fffffff, %2\n v_and_b32 %3, 0x1fffffff, %3\n v_and_b32 %4, 0x1fffffff, %4\n v_and_b32 %5, 0x1fffffff, %5\n v_and_b32 %6, 0x1fffffff, %6\n v_and_b32 %7, 0x1fffffff, %7\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_sub_u32, DECL8, OP8("v_sub_u32"), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_ashrrev_i32, DECL8, asm volatile("v_ashrrev_i32 %0, 3, %0\n v_ashrrev_i32 %1, 3, %1\n v_ashrrev_i32 %2, 3, %2\n v_ashrrev_i32 %3, 3, %3\n v_ashrrev_i32 %4, 3, %4\n v_ashrrev_i32 %5, 3, %5\n v_ashrrev_i32 %6, 3, %6\n v_ashrrev_i32 %7, 3, %7\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_mov_b32, DECL8, asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)), a0^a1^a2^a3^a4^a5^a6^a7)
KERNEL(k_add_lit, DECL8, asm volatile("v_add_u32 %0, 0x12345678, %0\n v_add_u32 %1, 0x12345678, %1\n v_add_u32 %2, 0x12345678, %2\n v_add_u32 %3, 0x12345678, %3\n v_add_u32 %4, 0x12345678, %4\n v_add_u32 %5, 0x12345678, %5\n v_add_u32 %6, 0x12345678, %6\n v_add_u32 %7, 0x12345678, %7\n" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)), a0^a1^a2^a3^a4^a5^a6^a7)

template <typename K> static double run(K kern, const char* name, int ops_per_body, int waves_per_simd, u32* d){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 256 * waves_per_simd;   // 256 CUs x (waves_per_simd) blocks of 256 threads = 4 waves => waves/SIMD
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
  hipDeviceSynchronize();
  float best=1e30f;
  for(int r=0;r<5;r++){ hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, (u32)r); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(ms<best)best=ms; }
  double lane_ops = (double)blocks*256*ITERS*UNROLL*ops_per_body;
  double rate = lane_ops/(best*1e-3);  // lane-ops / s
  printf("%-22s waves/SIMD=%d  %8.3f ms  %8.2f Tlane-op/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", name, waves_per_simd, best, rate/1e12, 2.4e9*1024*64/rate);
  return rate;
}
int main(int argc, char** argv){
  u32* d; CHECK(hipMalloc(&d, 256*8*256*4));
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p,0)); printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  if (argc > 1 && argv[1][0] == 'm') {      // "mix": the round-6 additions beside their round-2 neighbours, 4 waves per SIMD (the NTT kernels' occupancy)
    const int w = 4;
    for (int rep = 0; rep < 2; rep++) {
      run(k_mad_i64_i32,"v_mad_i64_i32",8,w,d); run(k_mad_i64_dep,"v_mad_i64_i32 dep",8,w,d); run(k_mad_i64_sgpr,"v_mad_i64_i32 dep sgpr",8,w,d);
      run(k_and_b32,"v_and_b32",8,w,d); run(k_and_lit,"v_and_b32 literal",8,w,d); run(k_add_u32,"v_add_u32",8,w,d); run(k_add_lit,"v_add_u32 literal",8,w,d);
      run(k_sub_u32,"v_sub_u32",8,w,d); run(k_ashrrev_i32,"v_ashrrev_i32",8,w,d); run(k_mov_b32,"v_mov_b32",8,w,d);
      run(k_ashrrev_i64,"v_ashrrev_i64",8,w,d); run(k_add3_u32,"v_add3_u32",8,w,d); run(k_addc_chain,"v_addc_co_u32 chain",8,w,d);
    }
    return 0;
  }
  for(int w : {1,2,4}){
    run(k_mad_u64_u32,"v_mad_u64_u32",8,w,d); run(k_mad_addc,"mad_u64+addc (pairs)",4,w,d); run(k_mad_dep,"v_mad_u64_u32 dep",8,w,d);
    run(k_mul_lo_u32,"v_mul_lo_u32",8,w,d); run(k_mul_hi_u32,"v_mul_hi_u32",8,w,d);
    run(k_mul_u32_u24,"v_mul_u32_u24",8,w,d); run(k_mul_hi_u32_u24,"v_mul_hi_u32_u24",8,w,d); run(k_mad_u32_u24,"v_mad_u32_u24",8,w,d);
    run(k_add_u32,"v_add_u32",8,w,d); run(k_xor_b32,"v_xor_b32",8,w,d); run(k_add3_u32,"v_add3_u32",8,w,d); run(k_alignbit,"v_alignbit_b32",8,w,d);
    run(k_and_b32,"v_and_b32",8,w,d); run(k_lshrrev_b32,"v_lshrrev_b32",8,w,d); run(k_bfe_u32,"v_bfe_u32",8,w,d); run(k_lshl_or_b32,"v_lshl_or_b32",8,w,d);
    run(k_cndmask,"v_cndmask_b32 (vcc)",8,w,d); run(k_cndmask_cmp,"v_cmp + 8 cndmask(vcc)",9,w,d); run(k_cndmask_e64,"v_cndmask_b32_e64 sgpr",8,w,d); run(k_sel_arith,"xor/and/xor select",8,w,d); run(k_addc_chain,"v_addc_co_u32 chain",8,w,d); run(k_lshrrev_b64,"v_lshrrev_b64",8,w,d); run(k_mad_snop,"mad_u64 + s_nop 0",4,w,d);
    run(k_lshl_add_u64,"v_lshl_add_u64",8,w,d); run(k_fma_f32,"v_fma_f32",8,w,d); run(k_fma_f64,"v_fma_f64",8,w,d);
    run(k_xor_sdwa,"v_xor_b32_sdwa (word)",8,w,d); run(k_perm_b32,"v_perm_b32",8,w,d); run(k_mad_i64_i32,"v_mad_i64_i32",8,w,d); run(k_ashrrev_i64,"v_ashrrev_i64",8,w,d);
    run(k_dpp_ror,"v_mov_b32_dpp row_ror",8,w,d); run(k_bpermute,"ds_bpermute_b32",8,w,d);
  }
  return 0;
}
