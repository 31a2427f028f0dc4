// lcpc_amd/csrc/rng.hip -- synthetic coefficient vectors on the device.
//
// lcpc_test_fields::random_coeffs (lcpc-test-fields/src/lib.rs:75-97 of /root/reference) fills the vectors its tests and
// benches commit with `Ft::random(&mut rng)` [3P ff_derive: draw L x next_u64 as limbs, mask the top limb to NUM_BITS, accept
// iff the raw integer is < p; the accepted limbs ARE the Montgomery representation].  The reference draws from thread_rng();
// SURVEY.md 8(d) fixes the generator -- ChaCha20Rng::from_seed, stream 0 -- so that the device and any host hold the SAME
// vector by seed, and a 2 GiB input never crosses the bus.
//
// The rule is a serial rejection sampler: element i is the i-th ACCEPTED candidate of the stream.  On the device candidate j
// (u64 words [L j, L j + L) of the keystream) is one thread; two passes over the candidates -- count the accepted ones per
// workgroup, exclusive scan of the counts, then place every accepted candidate at its rank -- recomputing the (cheap) ChaCha
// blocks instead of storing 1.25 x the output as a temporary.
#include "internal.h"

namespace lcpc {
namespace {

typedef uint32_t u32;
typedef uint64_t u64;

struct RngArgs {
  u32 key[8];
  u64 stream_id;
  u64 p[4];
  u64 top_mask;
  u32 L;
  u64 n_cand;
};

__device__ __forceinline__ u32 rotl32(u32 x, int k) { return (x << k) | (x >> (32 - k)); }
#define LCPC_QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);

// one ChaCha20 block in the rand_chacha layout: "expand 32-byte k" | key | 64-bit block counter | 64-bit stream id
__device__ __forceinline__ void chacha20_block(const RngArgs& a, u64 block, u32 out[16]) {
  u32 s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, a.key[0], a.key[1], a.key[2], a.key[3], a.key[4], a.key[5], a.key[6], a.key[7],
               (u32)block, (u32)(block >> 32), (u32)a.stream_id, (u32)(a.stream_id >> 32)};
  u32 x[16];
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = s[i];
#pragma unroll
  for (int r = 0; r < 10; r++) {
    LCPC_QR(x[0], x[4], x[8], x[12]) LCPC_QR(x[1], x[5], x[9], x[13]) LCPC_QR(x[2], x[6], x[10], x[14]) LCPC_QR(x[3], x[7], x[11], x[15])
    LCPC_QR(x[0], x[5], x[10], x[15]) LCPC_QR(x[1], x[6], x[11], x[12]) LCPC_QR(x[2], x[7], x[8], x[13]) LCPC_QR(x[3], x[4], x[9], x[14])
  }
#pragma unroll
  for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

// candidate j: limbs (top one masked) and whether it is accepted.  next_u64 = two consecutive words, low first; a block holds
// eight of them, so a candidate touches at most two blocks (L = 3 straddles them)
__device__ __forceinline__ bool candidate(const RngArgs& a, u64 j, u64 v[4]) {
  u32 blk[16];
  u64 have = ~(u64)0;
  for (u32 i = 0; i < a.L; i++) {
    const u64 k = j * a.L + i, b = k >> 3;
    if (b != have) { chacha20_block(a, b, blk); have = b; }
    const u32 w = (u32)(k & 7) * 2;
    // (dynamic index into a register array would spill: select)
    u32 lo = 0, hi = 0;
#pragma unroll
    for (u32 t = 0; t < 8; t++) { if (w == 2 * t) { lo = blk[2 * t]; hi = blk[2 * t + 1]; } }
    v[i] = (u64)lo | ((u64)hi << 32);
  }
  v[a.L - 1] &= a.top_mask;
  for (int i = (int)a.L - 1; i >= 0; i--) {
    if (v[i] < a.p[i]) return true;
    if (v[i] > a.p[i]) return false;
  }
  return false;                 // == p
}

__global__ void __launch_bounds__(256) rng_count_kernel(RngArgs a, u32* counts) {
  const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
  u64 v[4];
  const bool ok = j < a.n_cand && candidate(a, j, v);
  const int c = __syncthreads_count(ok ? 1 : 0);
  if (threadIdx.x == 0) counts[blockIdx.x] = (u32)c;
}

// exclusive scan of counts[0, n) -> offs[0, n], offs[n] = total.  One workgroup walks the array in tiles of 1024
__global__ void __launch_bounds__(1024) rng_scan_kernel(const u32* counts, u64 n, u64* offs) {
  __shared__ u64 sh[1024];
  __shared__ u64 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (u64 base = 0; base < n; base += 1024) {
    const u64 i = base + threadIdx.x;
    const u64 x = i < n ? counts[i] : 0;
    sh[threadIdx.x] = x;
    __syncthreads();
    for (u32 d = 1; d < 1024; d <<= 1) {                     // Hillis-Steele inclusive scan
      const u64 y = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += y;
      __syncthreads();
    }
    if (i < n) offs[i] = carry + sh[threadIdx.x] - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) offs[n] = carry;
}

__global__ void __launch_bounds__(256) rng_place_kernel(RngArgs a, const u64* offs, u64 n_out, u64* out) {
  __shared__ u32 wave_cnt[4];
  const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
  u64 v[4];
  const bool ok = j < a.n_cand && candidate(a, j, v);
  const u64 bal = __ballot(ok ? 1 : 0);
  const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u32 before = (u32)__popcll(bal & (((u64)1 << lane) - 1));
  if (lane == 0) wave_cnt[wv] = (u32)__popcll(bal);
  __syncthreads();
  u32 wbase = 0;
  for (u32 w = 0; w < wv; w++) wbase += wave_cnt[w];
  if (!ok) return;
  const u64 dst = offs[blockIdx.x] + wbase + before;
  if (dst >= n_out) return;
  for (u32 i = 0; i < a.L; i++) out[dst * a.L + i] = v[i];
}

}  // namespace
}  // namespace lcpc

using namespace lcpc;

extern "C" int lcpc_random_coeffs_device(lcpc_ctx* c, const uint8_t seed[32], uint64_t stream_id, uint64_t n, uint64_t* out_dev, void* stream) {
  if (!c || !seed || !out_dev) return LCPC_ERR_ARG;
  if (n == 0) return 0;
  LCPC_TRY
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  hipStream_t st = (hipStream_t)stream;
  const FieldDesc& f = *c->f;
  RngArgs a{};
  memcpy(a.key, seed, 32);                                 // (little-endian words, as rand_chacha reads its seed)
  a.stream_id = stream_id;
  for (int i = 0; i < 4; i++) a.p[i] = i < f.L ? f.p[i] : 0;
  a.top_mask = f.top_mask;
  a.L = (u32)f.L;
  // acceptance probability p / 2^NUM_BITS (>= 1/2); candidates for n elements with 2 % + 4096 of slack, doubled until enough
  const double p_acc = (double)f.p[f.L - 1] / ((double)f.top_mask + 1.0);
  double slack = 1.02;
  for (int attempt = 0; attempt < 8; attempt++, slack *= 2) {
    a.n_cand = (u64)((double)n / p_acc * slack) + 4096;
    const u64 n_wg = (a.n_cand + 255) / 256;
    u32* d_counts = nullptr;
    u64* d_offs = nullptr;
    int rc = dev_alloc(&c->err, &d_counts, (size_t)n_wg * 4);
    if (!rc) rc = dev_alloc(&c->err, &d_offs, (size_t)(n_wg + 1) * 8);
    if (rc) { dev_free(d_counts); return rc; }
    hipLaunchKernelGGL(rng_count_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, a, d_counts);
    hipLaunchKernelGGL(rng_scan_kernel, dim3(1), dim3(1024), 0, st, d_counts, n_wg, d_offs);
    u64 total = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&total, d_offs + n_wg, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && total >= n) {
      hipLaunchKernelGGL(rng_place_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, a, d_offs, n, out_dev);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(st);   // (the scratch below is freed on return)
    }
    dev_free(d_counts); dev_free(d_offs);
    if (e != hipSuccess) return fail_hip(&c->err, e, "lcpc_random_coeffs_device");
    if (total >= n) return 0;
  }
  c->err = "lcpc_random_coeffs_device: the candidate stream ran short";
  return LCPC_ERR_STATE;
  LCPC_CATCH(c)
}
