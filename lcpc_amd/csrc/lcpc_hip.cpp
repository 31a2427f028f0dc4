// lcpc_amd/csrc/lcpc_hip.cpp -- context + C ABI (include/lcpc_hip.h) of the MI355X lcpc-2d path.
//
// Host orchestration of: commit (lcpc-2d/src/lib.rs:622-671), merkleize (690-704), open_column
// (788-825), prove (1004-1093), collapse_columns (1095-1123), verify (832-1000) and the bincode wire
// layout (186-268, 352-609) of /root/reference.  Every heavy step is a HIP kernel (kernels.hip); there
// is no CPU fallback: without a usable HIP device lcpc_ctx_create fails with LCPC_ERR_NO_DEVICE.
#include "../../include/lcpc_hip.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <memory>
#include <mutex>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "encoding.h"
#include "host_crypto.h"
#include "host_field.h"
#include "kernels.h"

using namespace lcpc;

namespace {

struct DevCsr {
  uint64_t n_in = 0, n_out = 0;
  uint32_t *rowptr = nullptr, *colidx = nullptr, *vals = nullptr;
  uint32_t* vals29 = nullptr;     // Ft255: values in the 29-bit-limb / 2^261 form (lazy29_mac)
};
struct Pass { uint32_t t0, s, log_tj; int log_tile; };

// Ft255 element in ff_derive's Montgomery form (a * 2^256) -> a * 2^261 mod p as 9 limbs of 29 bits, 12-word stride:
// the multiplier format of fe_mul_r29 / lazy29_mac (field_dev.h)
void to_r29(const FieldDesc& f, const uint64_t* in4, uint32_t* out12) {
  uint64_t t[4];
  memcpy(t, in4, 32);
  for (int d = 0; d < 5; d++) h_add(f, t, t, t);
  for (int k = 0; k < 9; k++) {
    const int b = 29 * k, w = b / 64, sh = b % 64;
    uint64_t x = t[w] >> sh;
    if (sh > 35 && w + 1 < 4) x |= t[w + 1] << (64 - sh);
    out12[k] = (uint32_t)(x & ((1u << 29) - 1));
  }
  out12[9] = out12[10] = out12[11] = 0;
}

// small fork-join helper for the host-side glue (the reference uses rayon at the same places: lib.rs:923-944)
// host cores this process may really use: hardware threads capped by the cgroup CPU quota (a container can show 256
// hardware threads and be granted 16 CPUs of time; more threads than that only get throttled)
unsigned usable_cores() {
  static const unsigned cached = [] {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    long long q = -1, per = 100000;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                         // cgroup v2: "<quota|max> <period>"
      char qs[32] = {0};
      if (fscanf(f, "%31s %lld", qs, &per) == 2 && strcmp(qs, "max") != 0) q = atoll(qs);
      fclose(f);
    } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {      // cgroup v1
      if (fscanf(f1, "%lld", &q) != 1) q = -1;
      fclose(f1);
      if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lld", &per) != 1) per = 100000; fclose(f2); }
    }
    if (q > 0 && per > 0) { const unsigned lim = (unsigned)std::max<long long>(1, q / per); if (lim < n) n = lim; }
    return n;
  }();
  return cached;
}

template <typename Fn> void parallel_for(uint64_t n, uint64_t grain, Fn fn, unsigned max_threads = 16) {
  unsigned nt = usable_cores();
  if (nt > max_threads) nt = max_threads;
  if (nt <= 1 || n < 2 * grain) { fn((uint64_t)0, n); return; }
  const uint64_t nchunks = (n + grain - 1) / grain;
  if (nt > nchunks) nt = (unsigned)nchunks;
  std::atomic<uint64_t> next{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&] {
      for (;;) {
        const uint64_t c = next.fetch_add(1);
        if (c >= nchunks) return;
        const uint64_t b = c * grain, e = b + grain < n ? b + grain : n;
        fn(b, e);
      }
    });
  for (auto& x : th) x.join();
}

const uint8_t LBL_DT[] = "$l//DT", LBL_PR[] = "$l//PR", LBL_PE[] = "$l//PE", LBL_CO[] = "$l//CO";  // macros.rs:31-34

}  // namespace

struct lcpc_transcript {
  Transcript t;
  lcpc_transcript(const uint8_t* l, size_t n) : t(l, n) {}
};

struct lcpc_ctx {
  lcpc_params prm{};
  const FieldDesc* f = nullptr;
  int L = 0, NL = 0;
  uint64_t n_per_row = 0, n_cols = 0, np2 = 0;
  uint32_t path_len = 0;
  // Ligero
  unsigned log_n = 0;
  uint32_t* d_roots = nullptr;
  uint32_t* d_roots29 = nullptr;   // Ft255: twiddles in radix-2^29 / R'=2^261 Montgomery form (field_dev.h fe_mul_r29)
  uint32_t* d_qp29 = nullptr;      // Ft255: q*p, q < 32, as 29-bit limbs (l9::clamp); null = packed-form NTT kernel
  uint32_t* d_roots29c = nullptr;  // Ft255 lazy-limb kernel: w^i * 2^5, the table that converts to canonical on the fly
  bool comm_canon = false;         // d_comm of a commit holds canonical values (x * R^-1), not Montgomery form: the column
                                   // hash reads them as they are; every read-out (get_comm, open_columns) converts back
  std::vector<Pass> passes;
  // Brakedown
  SdigSpec spec{};
  std::vector<LevelDims> pre_dims, post_dims;
  std::vector<DevCsr> d_pre, d_post;
  uint32_t* d_r2 = nullptr;
  uint32_t* d_tmp = nullptr;       // last precode output, n_rows x m_last
  uint64_t tmp_cap = 0;
  uint32_t* d_t = nullptr;         // Brakedown: position-major working copy T[pos][row] of the rows being encoded
  uint64_t t_cap = 0;
  bool comm_t = false;             // Brakedown commit with >= 16 local rows: the commitment matrix lives in d_t (position-major,
                                   // element (row, col) at (col * n_rows_local + row)); hash / open read it there, d_comm is only
                                   // filled on demand (lcpc_get_comm) -- no back-transpose on the commit path
  bool comm_rows_valid = false;    // d_comm holds the row-major copy of the commitment in d_t
  // commitment (device resident)
  bool committed = false;
  uint64_t n_rows = 0;             // rows of the whole commitment
  uint64_t row_begin = 0, n_rows_local = 0;
  uint64_t chunk_begin = 0, chunk_end = 0, n_chunks = 0;
  uint32_t *d_coeffs = nullptr, *d_comm = nullptr, *d_hashes = nullptr, *d_cvs = nullptr;
  uint32_t* d_node_tab = nullptr;  // sharded finish: node_slot[0..n) then node_log[0..n)
  uint64_t node_tab_key = 0;       // (n_chunks << 16 | slots_per_rank) the table was built for
  uint64_t cap_rows = 0, cap_cvs = 0;
  // scratch for prove / collapse / open
  uint32_t* d_scratch = nullptr;
  uint64_t scratch_cap = 0;
  uint32_t* d_t29 = nullptr;       // collapse: tensors in the 29-bit-limb form
  uint64_t t29_cap = 0;
  // timing
  bool timing = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t s_copy = nullptr, s_comp = nullptr;   // lcpc_commit (host pointer): H2D of row batch b+1 overlaps the NTTs of batch b
  hipEvent_t ev_batch[16] = {nullptr};
  lcpc_timings last{};
  uint32_t launches[3] = {0, 0, 0};
  std::string err;
  std::mutex mu;
};

namespace {

int fail_hip(lcpc_ctx* c, hipError_t e, const char* what) {
  if (c) c->err = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? LCPC_ERR_NOMEM : LCPC_ERR_HIP;
}
#define HIPCHK(c, call)                                   \
  do {                                                    \
    hipError_t e__ = (call);                              \
    if (e__ != hipSuccess) return fail_hip(c, e__, #call); \
  } while (0)

template <typename T> int dev_alloc(lcpc_ctx* c, T** p, size_t bytes) {
  *p = nullptr;
  if (bytes == 0) bytes = 16;
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(p), bytes));
  return 0;
}
void dev_free(void* p) { if (p) (void)hipFree(p); }

size_t elem_bytes(const lcpc_ctx* c) { return (size_t)8 * c->L; }

// ---- NTT pass plan (DESIGN.md "K1") ----------------------------------------------------------------
void plan_passes(lcpc_ctx* c) {
  const unsigned k = c->log_n;
  const int NL = c->NL;
  const int lt_small = NL >= 6 ? 10 : (NL == 4 ? 11 : 12);      // 32 KiB (24 KiB for NL=6) tiles
  const int lt_big = NL >= 6 ? 11 : 12;
  unsigned ltj_min = 0;                                          // >= 128 B contiguous runs in strided passes
  while (((size_t)NL * 4 << ltj_min) < 128) ltj_min++;
  c->passes.clear();
  if ((int)k <= lt_small) {
    c->passes.push_back({0, k, 0, lt_small});
    return;
  }
  int LT = lt_small;
  auto n_pass = [&](int lt) { unsigned per = lt - ltj_min, rest = k - lt; return 1 + (rest + per - 1) / per; };
  if (n_pass(lt_small) > 2 && n_pass(lt_big) < n_pass(lt_small)) LT = lt_big;
  const unsigned P = n_pass(LT);
  const unsigned s_final = LT;
  unsigned rem = k - s_final, t0 = 0;
  for (unsigned i = 0; i + 1 < P; i++) {
    const unsigned left = P - 1 - i;
    const unsigned s = (rem + left - 1) / left;
    c->passes.push_back({t0, s, (uint32_t)LT - s, LT});
    t0 += s;
    rem -= s;
  }
  c->passes.push_back({t0, s_final, 0u, LT});
}

int ensure_buffers(lcpc_ctx* c, uint64_t n_rows_local) {
  if (n_rows_local > c->cap_rows || !c->d_comm) {
    dev_free(c->d_coeffs); dev_free(c->d_comm);
    c->d_coeffs = c->d_comm = nullptr;
    c->cap_rows = 0;
    const size_t eb = elem_bytes(c);
    int rc;
    if ((rc = dev_alloc(c, &c->d_coeffs, (size_t)n_rows_local * c->n_per_row * eb))) return rc;
    if ((rc = dev_alloc(c, &c->d_comm, (size_t)n_rows_local * c->n_cols * eb))) return rc;
    c->cap_rows = n_rows_local;
  }
  if (!c->d_hashes) {
    int rc = dev_alloc(c, &c->d_hashes, (size_t)(2 * c->np2 - 1) * 32);
    if (rc) return rc;
  }
  if (c->prm.encoding == LCPC_ENC_SDIG) {
    const uint64_t need = n_rows_local * c->pre_dims.back().m;
    if (need > c->tmp_cap) {
      dev_free(c->d_tmp);
      int rc = dev_alloc(c, &c->d_tmp, (size_t)need * elem_bytes(c));
      if (rc) return rc;
      c->tmp_cap = need;
    }
  }
  return 0;
}
int ensure_cvs(lcpc_ctx* c, uint64_t n_chunks) {
  const uint64_t need = n_chunks * c->n_cols;
  if (need > c->cap_cvs) {
    dev_free(c->d_cvs);
    int rc = dev_alloc(c, &c->d_cvs, (size_t)need * 32);
    if (rc) return rc;
    c->cap_cvs = need;
  }
  return 0;
}
int ensure_scratch(lcpc_ctx* c, uint64_t bytes) {
  if (bytes > c->scratch_cap) {
    dev_free(c->d_scratch);
    int rc = dev_alloc(c, &c->d_scratch, (size_t)bytes);
    if (rc) return rc;
    c->scratch_cap = bytes;
  }
  return 0;
}

// leaf message = 32 + F * n_rows bytes -> BLAKE3 chunks of 1 KiB
uint64_t leaf_chunks(const lcpc_ctx* c, uint64_t n_rows) { return (32 + elem_bytes(c) * n_rows + 1023) / 1024; }

// rows / chunks of shard `rank` (DESIGN.md "multi-GPU"): chunk-aligned row blocks
void shard_layout_of(const lcpc_ctx* c, uint64_t g, uint64_t n_rows, uint64_t* rb, uint64_t* re, uint64_t* cb, uint64_t* ce, uint64_t* nch) {
  const uint64_t n_chunks = leaf_chunks(c, n_rows);
  const uint64_t G = c->prm.shard_count > 1 ? c->prm.shard_count : 1;
  if (G == 1) g = 0;
  const uint64_t c0 = n_chunks * g / G, c1 = n_chunks * (g + 1) / G;
  const uint64_t F = elem_bytes(c);
  auto first_row = [&](uint64_t chunk) -> uint64_t {   // first row whose bytes start in or after this chunk
    if (chunk == 0) return 0;
    const uint64_t byte = chunk * 1024 - 32;             // F | 1024 is enforced for sharded contexts
    uint64_t r = byte / F;
    return r < n_rows ? r : n_rows;
  };
  *cb = c0; *ce = c1; *nch = n_chunks;
  *rb = first_row(c0);
  *re = c1 >= n_chunks ? n_rows : first_row(c1);
  if (c0 == c1) *re = *rb;
}
void shard_layout(const lcpc_ctx* c, uint64_t n_rows, uint64_t* rb, uint64_t* re, uint64_t* cb, uint64_t* ce, uint64_t* nch) {
  shard_layout_of(c, c->prm.shard_rank, n_rows, rb, re, cb, ce, nch);
}

// aligned power-of-two decomposition of the chunk range [c0, c1): the subtree nodes a shard exchanges
int shard_nodes(uint64_t c0, uint64_t c1, uint64_t* first, uint32_t* lg) {
  int n = 0;
  for (uint64_t pos = c0; pos < c1;) {
    uint32_t l = 0;
    while ((pos == 0 || (pos & (((uint64_t)2 << l) - 1)) == 0) && pos + ((uint64_t)2 << l) <= c1) l++;
    first[n] = pos; lg[n] = l; n++;
    pos += (uint64_t)1 << l;
  }
  return n;
}

// ---- encode all local rows: coeffs -> comm ----------------------------------------------------------
// canon_out: dst receives canonical values instead of Montgomery form (commit paths of a comm_canon context only)
int encode_rows_device(lcpc_ctx* c, const uint32_t* src, uint64_t src_stride, uint64_t n_valid, uint32_t* dst,
                       uint64_t n_rows, hipStream_t st, uint64_t n_src_total = ~(uint64_t)0, uint32_t* copy_dst = nullptr,
                       bool canon_out = false, bool keep_t = false) {
  if (n_rows == 0) return 0;
  if (c->prm.encoding == LCPC_ENC_LIGERO) {
    bool first = true;
    for (const Pass& p : c->passes) {
      NttPassArgs a;
      a.roots29c = canon_out ? c->d_roots29c : nullptr;
      // the trailing stages multiply by 1 only: a final radix-4 round leaves 4 elements per row unconverted, a final
      // radix-2 stage 2 (ntt_pass_l9_kernel)
      a.mont_prefix = (canon_out && p.t0 + p.s == c->log_n) ? (c->log_n == 0 ? 1u : (p.s % 2 == 0 ? 4u : 2u)) : 0u;
      a.src = first ? src : dst;
      a.dst = dst;
      a.roots = c->d_roots;
      a.roots29 = c->d_roots29;
      a.qp29 = c->d_qp29;
      a.src_stride = first ? src_stride : c->n_cols;
      a.dst_stride = c->n_cols;
      a.n_valid = first ? n_valid : c->n_cols;
      a.n_src_total = first ? n_src_total : ~(uint64_t)0;
      a.copy_dst = first ? copy_dst : nullptr;
      a.n_rows = n_rows;
      a.log_n = c->log_n; a.t0 = p.t0; a.s = p.s; a.log_tj = p.log_tj;
      HIPCHK(c, launch_ntt_pass(c->NL, p.log_tile, a, st));
      c->launches[0]++;
      first = false;
    }
    return 0;
  }
  // Brakedown: systematic part, then precodes down, R-S base case, postcodes up (encode.rs:36-94)
  const size_t t = c->d_pre.size();
  const DevCsr& pl = c->d_pre[t - 1];
  {
    const uint64_t need = n_rows * pl.n_out;
    if (need > c->tmp_cap) {
      dev_free(c->d_tmp);
      int rc = dev_alloc(c, &c->d_tmp, (size_t)need * elem_bytes(c));
      if (rc) return rc;
      c->tmp_cap = need;
    }
  }
  if (n_rows >= 16 && (keep_t || !(c->comm_t && c->committed))) {       // (d_t may be holding a live commitment)
    // fast path: position-major working copy T[pos][row] (lane = row: contiguous gathers, wave-uniform matrix)
    const uint64_t need = n_rows * c->n_cols;
    if (need > c->t_cap) {
      dev_free(c->d_t);
      int rc = dev_alloc(c, &c->d_t, (size_t)need * elem_bytes(c));
      if (rc) return rc;
      c->t_cap = need;
    }
    HIPCHK(c, launch_transpose_to_t(c->NL, src, src_stride, n_valid, n_rows, c->d_t, st, n_src_total, copy_dst));
    c->launches[0]++;
    uint64_t in_start = 0;
    SpmmTArgs a{};
    a.t = c->d_t; a.n_rows = n_rows;
    auto set_mat = [&](const DevCsr& m) { a.rowptr = m.rowptr; a.colidx = m.colidx; a.vals = m.vals; a.vals29 = m.vals29; a.m = m.n_out; };
    for (size_t i = 0; i + 1 < t; i++) {
      const uint64_t in_end = in_start + c->d_pre[i].n_in;
      a.out_alt = nullptr; a.in_off = in_start; a.out_off = in_end;
      set_mat(c->d_pre[i]);
      HIPCHK(c, launch_spmm_t(c->NL, a, st));
      c->launches[0]++;
      in_start = in_end;
    }
    const uint64_t in_end = in_start + pl.n_in;
    a.out_alt = c->d_tmp; a.in_off = in_start; a.out_off = 0;
    set_mat(pl);
    HIPCHK(c, launch_spmm_t(c->NL, a, st));
    const uint64_t out_end = in_end + c->d_post[t - 1].n_in;
    HIPCHK(c, launch_sdig_rs_t(c->NL, c->d_tmp, (uint32_t)pl.n_out, c->d_t, in_end, (uint32_t)c->d_post[t - 1].n_in, n_rows, c->d_r2, st));
    c->launches[0] += 2;
    in_start = in_end + pl.n_out;
    uint64_t out_start = out_end;
    for (size_t ii = t; ii-- > 0;) {
      in_start -= c->d_pre[ii].n_out;
      a.out_alt = nullptr; a.in_off = in_start; a.out_off = out_start;
      set_mat(c->d_post[ii]);
      HIPCHK(c, launch_spmm_t(c->NL, a, st));
      c->launches[0]++;
      out_start += c->d_post[ii].n_out;
    }
    if (keep_t) {               // commit: the position-major copy IS the commitment (hash_columns / open_column read it)
      c->comm_t = true;
      c->comm_rows_valid = false;
      return 0;
    }
    HIPCHK(c, launch_transpose_from_t(c->NL, c->d_t, c->n_cols, n_rows, dst, c->n_cols, st));
    c->launches[0]++;
    return 0;
  }
  // few rows (the verifier's 1 + n_degree_tests single-row encodes): lane = output on the row-major rows
  if (src != dst || src_stride != c->n_cols) {
    HIPCHK(c, launch_pad_rows(c->NL, src, src_stride, dst, c->n_cols, n_valid, n_rows, st));
    c->launches[0]++;
  }
  uint64_t in_start = 0;
  SpmvArgs a{};
  a.mat = dst; a.stride = c->n_cols; a.n_rows = n_rows;
  for (size_t i = 0; i + 1 < t; i++) {
    const uint64_t in_end = in_start + c->d_pre[i].n_in;
    a.out_alt = nullptr; a.in_off = in_start; a.out_off = in_end;
    a.rowptr = c->d_pre[i].rowptr; a.colidx = c->d_pre[i].colidx; a.vals = c->d_pre[i].vals; a.m = c->d_pre[i].n_out;
    HIPCHK(c, launch_spmv(c->NL, a, st));
    c->launches[0]++;
    in_start = in_end;
  }
  const uint64_t in_end = in_start + pl.n_in;
  a.out_alt = c->d_tmp; a.out_alt_stride = pl.n_out; a.in_off = in_start; a.out_off = 0;
  a.rowptr = pl.rowptr; a.colidx = pl.colidx; a.vals = pl.vals; a.m = pl.n_out;
  HIPCHK(c, launch_spmv(c->NL, a, st));
  const uint64_t out_end = in_end + c->d_post[t - 1].n_in;
  HIPCHK(c, launch_sdig_rs(c->NL, c->d_tmp, pl.n_out, (uint32_t)pl.n_out, dst, c->n_cols, in_end,
                           (uint32_t)c->d_post[t - 1].n_in, n_rows, c->d_r2, st));
  c->launches[0] += 2;
  in_start = in_end + pl.n_out;
  uint64_t out_start = out_end;
  for (size_t ii = t; ii-- > 0;) {
    in_start -= c->d_pre[ii].n_out;
    a.out_alt = nullptr; a.in_off = in_start; a.out_off = out_start;
    a.rowptr = c->d_post[ii].rowptr; a.colidx = c->d_post[ii].colidx; a.vals = c->d_post[ii].vals; a.m = c->d_post[ii].n_out;
    HIPCHK(c, launch_spmv(c->NL, a, st));
    c->launches[0]++;
    out_start += c->d_post[ii].n_out;
  }
  return 0;
}

// hash_columns + merkle_tree on the local comm (unsharded) -- lib.rs:690-704
int merkleize_device(lcpc_ctx* c, hipStream_t st) {
  const uint64_t n_chunks = leaf_chunks(c, c->n_rows);
  LeafArgs la{};
  la.comm = c->d_comm; la.canon_in = c->comm_canon ? 1u : 0u; la.row_stride = c->n_cols; la.col_stride = 1; la.n_cols = c->n_cols; la.row_base = 0;
  la.n_rows_total = c->n_rows; la.chunk_begin = 0; la.n_chunks_local = (uint32_t)n_chunks; la.n_chunks_total = (uint32_t)n_chunks;
  if (c->comm_t) { la.comm = c->d_t; la.row_stride = 1; la.col_stride = c->n_rows_local; }
  if (n_chunks == 1) {
    la.out = c->d_hashes;
    HIPCHK(c, launch_leaf_chunks(c->NL, la, st));
    c->launches[1]++;
  } else {
    int rc = ensure_cvs(c, n_chunks);
    if (rc) return rc;
    la.out = c->d_cvs;
    HIPCHK(c, launch_leaf_chunks(c->NL, la, st));
    HIPCHK(c, launch_leaf_finish(c->d_cvs, (uint32_t)n_chunks, c->n_cols, c->d_hashes, st));
    c->launches[1] += 2;
  }
  if (c->timing) HIPCHK(c, hipEventRecord(c->ev[2], st));
  if (c->np2 > c->n_cols)   // hashes[n_cols..np2) stay zero (lib.rs:656-666)
    HIPCHK(c, hipMemsetAsync(c->d_hashes + c->n_cols * 8, 0, (size_t)(c->np2 - c->n_cols) * 32, st));
  if (c->np2 > 1) {
    HIPCHK(c, launch_merkle_tree(c->d_hashes, c->np2, st));
    c->launches[2]++;
  }
  return 0;
}

int finish_timing(lcpc_ctx* c, hipStream_t st) {
  if (!c->timing) return 0;
  HIPCHK(c, hipEventRecord(c->ev[3], st));
  HIPCHK(c, hipEventSynchronize(c->ev[3]));
  (void)hipEventElapsedTime(&c->last.encode_ms, c->ev[0], c->ev[1]);
  (void)hipEventElapsedTime(&c->last.hash_ms, c->ev[1], c->ev[2]);
  (void)hipEventElapsedTime(&c->last.merkle_ms, c->ev[2], c->ev[3]);
  (void)hipEventElapsedTime(&c->last.total_ms, c->ev[0], c->ev[3]);
  c->last.encode_launches = c->launches[0];
  c->last.hash_launches = c->launches[1];
  c->last.merkle_launches = c->launches[2];
  return 0;
}

// commit with the padded coefficient matrix already in c->d_coeffs
int commit_resident(lcpc_ctx* c, hipStream_t st, uint8_t* root, const uint32_t* ext_src = nullptr, uint64_t n_ext = 0) {
  c->launches[0] = c->launches[1] = c->launches[2] = 0;
  if (c->timing) HIPCHK(c, hipEventRecord(c->ev[0], st));
  c->comm_t = false;
  int rc = ext_src ? encode_rows_device(c, ext_src, c->n_per_row, c->n_per_row, c->d_comm, c->n_rows_local, st, n_ext, c->d_coeffs,
                                        c->comm_canon, true)
                   : encode_rows_device(c, c->d_coeffs, c->n_per_row, c->n_per_row, c->d_comm, c->n_rows_local, st, ~(uint64_t)0,
                                        nullptr, c->comm_canon, true);
  if (rc) return rc;
  if (c->timing) HIPCHK(c, hipEventRecord(c->ev[1], st));
  if ((rc = merkleize_device(c, st))) return rc;
  if ((rc = finish_timing(c, st))) return rc;
  c->committed = true;
  if (root) {
    HIPCHK(c, hipMemcpyAsync(root, c->d_hashes + (2 * c->np2 - 2) * 8, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
  }
  return 0;
}

int set_rows(lcpc_ctx* c, uint64_t n_coeffs) {
  if (n_coeffs == 0) return LCPC_ERR_ARG;
  c->n_rows = (n_coeffs + c->n_per_row - 1) / c->n_per_row;    // get_dims (ligero lib.rs:166-169)
  c->row_begin = 0;
  c->n_rows_local = c->n_rows;
  return ensure_buffers(c, c->n_rows);
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

int lcpc_abi_version(void) { return LCPC_ABI_VERSION; }

const char* lcpc_strerror(int s) {
  switch (s) {
    case LCPC_OK: return "ok";
    case LCPC_ERR_TOO_BIG: return "n_cols is too large for this encoding";
    case LCPC_ERR_ENCODE: return "encoding error";
    case LCPC_ERR_COMMIT: return "inconsistent commitment fields";
    case LCPC_ERR_COLUMN_NUMBER: return "bad column number";
    case LCPC_ERR_OUTER_TENSOR: return "outer tensor: wrong size";
    case LCPC_ERR_DIMS: return "dimensions not valid for this encoding";
    case LCPC_ERR_ARG: return "invalid argument";
    case LCPC_ERR_STATE: return "no commitment in this context";
    case LCPC_ERR_HIP: return "HIP runtime error";
    case LCPC_ERR_NOMEM: return "out of device memory";
    case LCPC_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case LCPC_ERR_XCHG: return "all-gather callback of the sharded prove failed";
    case LCPC_VERR_NUM_COL_OPENS: return "wrong number of column openings in proof";
    case LCPC_VERR_COLUMN_PATH: return "column verification: merkle path failed";
    case LCPC_VERR_COLUMN_EVAL: return "column verification: eval dot product failed";
    case LCPC_VERR_COLUMN_DEGREE: return "column verification: degree test dot product failed";
    case LCPC_VERR_OUTER_TENSOR: return "outer tensor: wrong size";
    case LCPC_VERR_INNER_TENSOR: return "inner tensor: wrong size";
    case LCPC_VERR_ENCODING_DIMS: return "encoding dimension mismatch";
    case LCPC_VERR_ENCODE: return "encoding error";
    case LCPC_VERR_MALFORMED: return "malformed proof bytes";
  }
  return "unknown status";
}
const char* lcpc_last_error(const lcpc_ctx* c) { return c ? c->err.c_str() : ""; }

int lcpc_static_get_dims(const lcpc_params* p, uint64_t* nr, uint64_t* np, uint64_t* nc) {
  if (!p || !nr || !np || !nc) return LCPC_ERR_ARG;
  const FieldDesc* f = field_desc((int)p->field);
  if (!f || p->n_coeffs == 0) return LCPC_ERR_ARG;
  if (p->encoding == LCPC_ENC_LIGERO) {
    if (p->rho_num == 0 || p->rho_num >= p->rho_den) return LCPC_ERR_ARG;
    return ligero_get_dims(*f, p->n_coeffs, p->rho_num, p->rho_den, nr, np, nc) ? LCPC_ERR_TOO_BIG : 0;
  } else if (p->encoding == LCPC_ENC_SDIG) {
    SdigSpec s;
    const int code = p->sdig_code ? (int)p->sdig_code : 3;
    uint64_t npr;
    if (!sdig_spec(code, &s) || !sdig_n_per_row(*f, p->n_coeffs, code, &npr)) return LCPC_ERR_ARG;
    std::vector<LevelDims> pre, post;
    if (!sdig_level_dims(s, npr, (double)f->flog2(), pre, post)) return LCPC_ERR_DIMS;
    *nr = (p->n_coeffs + npr - 1) / npr; *np = npr; *nc = sdig_codeword_length(pre, post);
    return 0;
  }
  return LCPC_ERR_ARG;
}

// new_ml (ligero lib.rs:128-135, brakedown lib.rs:114-123): dims for a multilinear polynomial in n_vars variables
int lcpc_static_get_dims_ml(const lcpc_params* p, uint32_t n_vars, uint64_t* nr, uint64_t* np, uint64_t* nc) {
  if (!p || !nr || !np || !nc || n_vars >= 63) return LCPC_ERR_ARG;
  const FieldDesc* f = field_desc((int)p->field);
  if (!f) return LCPC_ERR_ARG;
  const uint64_t n = (uint64_t)1 << n_vars;
  if (p->encoding == LCPC_ENC_LIGERO) {
    lcpc_params q = *p;
    q.n_coeffs = n;
    int rc = lcpc_static_get_dims(&q, nr, np, nc);
    if (rc) return rc;
    // the reference's assert!s (lib.rs:131-133)
    if ((*nr & (*nr - 1)) || (*np & (*np - 1)) || *nr * *np != n) return LCPC_ERR_DIMS;
    return 0;
  } else if (p->encoding == LCPC_ENC_SDIG) {
    SdigSpec s;
    const int code = p->sdig_code ? (int)p->sdig_code : 3;
    uint64_t npr;
    if (!sdig_spec(code, &s) || !sdig_n_per_row(*f, n, code, &npr, true)) return LCPC_ERR_ARG;
    std::vector<LevelDims> pre, post;
    if (!sdig_level_dims(s, npr, (double)f->flog2(), pre, post)) return LCPC_ERR_DIMS;
    *nr = (n + npr - 1) / npr; *np = npr; *nc = sdig_codeword_length(pre, post);
    return 0;
  }
  return LCPC_ERR_ARG;
}

int lcpc_ctx_create(const lcpc_params* p, lcpc_ctx** out) {
  if (!p || !out) return LCPC_ERR_ARG;
  *out = nullptr;
  const FieldDesc* f = field_desc((int)p->field);
  if (!f || p->hash != LCPC_HASH_BLAKE3) return LCPC_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device < 0 || p->device >= ndev) return LCPC_ERR_NO_DEVICE;
  if (hipSetDevice(p->device) != hipSuccess) return LCPC_ERR_NO_DEVICE;
  lcpc_ctx* c = new lcpc_ctx();
  c->prm = *p;
  c->f = f; c->L = f->L; c->NL = 2 * f->L;
  if (c->prm.sdig_code == 0) c->prm.sdig_code = 3;
  if (c->prm.shard_count > 1 && (c->prm.shard_rank >= c->prm.shard_count || 1024 % (8 * f->L) != 0)) {
    delete c;
    return LCPC_ERR_ARG;      // row sharding needs rows that do not straddle BLAKE3 chunks (F | 1024)
  }
  int rc = 0;
  if (p->encoding == LCPC_ENC_LIGERO) {
    if (p->rho_num == 0 || p->rho_num >= p->rho_den) { delete c; return LCPC_ERR_ARG; }
    uint64_t nr, np_, nc;
    if (p->n_per_row && p->n_cols) { np_ = p->n_per_row; nc = p->n_cols; }      // new_from_dims (ligero lib.rs:138-148)
    else if ((rc = lcpc_static_get_dims(p, &nr, &np_, &nc))) { delete c; return rc; }
    if (!(np_ < nc) || (nc & (nc - 1)) || log2_ceil(nc) > f->S) { delete c; return LCPC_ERR_DIMS; }   // _dims_ok + precomp_fft
    if (log2_ceil(nc) > 30) { delete c; return LCPC_ERR_TOO_BIG; }      // device kernels index a row with 32 bits
    c->n_per_row = np_; c->n_cols = nc; c->log_n = (unsigned)log2_ceil(nc);
    {
      // precomp_fft (fffft [3P]): w = ROOT_OF_UNITY^(2^(S - log_n)); the host only computes the log_n - 1 squares
      // w^(2^j), the n/2-entry tables are filled on the device (kernels.hip roots_kernel)
      const unsigned log_half = c->log_n ? c->log_n - 1 : 0;
      const size_t n_roots = (size_t)1 << log_half;
      std::vector<uint64_t> pw((size_t)(log_half + 1) * f->L);
      uint64_t w[MAXL];
      memcpy(w, f->rou, 8 * f->L);
      for (unsigned i = 0; i < f->S - c->log_n; i++) h_mul(*f, w, w, w);
      for (unsigned j = 0; j <= log_half; j++) { memcpy(&pw[(size_t)j * f->L], w, 8 * f->L); h_mul(*f, w, w, w); }
      uint32_t *d_pw = nullptr, *d_one = nullptr;
      if ((rc = dev_alloc(c, &d_pw, pw.size() * 8)) || (rc = dev_alloc(c, &d_one, 8 * f->L)) ||
          (rc = dev_alloc(c, &c->d_roots, n_roots * 8 * f->L)) ||
          (f->L == 4 && (rc = dev_alloc(c, &c->d_roots29, n_roots * 48))) ||
          (f->L == 4 && (rc = dev_alloc(c, &c->d_roots29c, n_roots * 48)))) {
        dev_free(d_pw); dev_free(d_one); lcpc_ctx_destroy(c); return rc;
      }
      hipError_t he = hipMemcpy(d_pw, pw.data(), pw.size() * 8, hipMemcpyHostToDevice);
      if (he == hipSuccess) he = hipMemcpy(d_one, f->r, 8 * f->L, hipMemcpyHostToDevice);
      if (he == hipSuccess) he = launch_roots(c->NL, d_pw, log_half, d_one, c->d_roots, c->d_roots29, c->d_roots29c, nullptr);
      if (he == hipSuccess) he = hipDeviceSynchronize();
      dev_free(d_pw); dev_free(d_one);
      if (he != hipSuccess) { lcpc_ctx_destroy(c); return LCPC_ERR_HIP; }
    }
    if (f->L == 4 && !getenv("LCPC_NTT_PACKED")) {
      // (i - 24) * p for i < 64 as normalised signed 29-bit limbs (limbs 0..7 in [0, 2^29), limb 8 two's complement):
      // the table behind l9::clamp (lazy-limb NTT kernel; QOFF in field_dev.h)
      std::vector<uint32_t> tab(64 * 12, 0);
      for (int i = 0; i < 64; i++) {
        const int q = i - 24;
        uint64_t mag[5] = {0, 0, 0, 0, 0};                      // |q| * p
        unsigned __int128 cy = 0;
        for (int w = 0; w < 5; w++) { cy += (unsigned __int128)(w < 4 ? f->p[w] : 0) * (uint64_t)(q < 0 ? -q : q); mag[w] = (uint64_t)cy; cy >>= 64; }
        if (q < 0) {                                            // two's complement over 320 bits
          unsigned __int128 c2 = 1;
          for (int w = 0; w < 5; w++) { c2 += (unsigned __int128)(~mag[w]); mag[w] = (uint64_t)c2; c2 >>= 64; }
        }
        for (int k = 0; k < 9; k++) {
          const int b = 29 * k, w = b / 64, sh = b % 64;
          uint64_t x = mag[w] >> sh;
          if (sh > 35) x |= mag[w + 1] << (64 - sh);
          tab[i * 12 + k] = k < 8 ? (uint32_t)(x & ((1u << 29) - 1)) : (uint32_t)x;      // limb 8: bits 232..263, sign-extended
        }
      }
      if ((rc = dev_alloc(c, &c->d_qp29, tab.size() * 4))) { lcpc_ctx_destroy(c); return rc; }
      if (hipMemcpy(c->d_qp29, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { lcpc_ctx_destroy(c); return LCPC_ERR_HIP; }
      c->comm_canon = !getenv("LCPC_COMM_MONT");
      if ((rc = dev_alloc(c, &c->d_r2, 8 * f->L))) { lcpc_ctx_destroy(c); return rc; }
      if (hipMemcpy(c->d_r2, f->r2, 8 * f->L, hipMemcpyHostToDevice) != hipSuccess) { lcpc_ctx_destroy(c); return LCPC_ERR_HIP; }
    }
    plan_passes(c);
  } else if (p->encoding == LCPC_ENC_SDIG) {
    if (!sdig_spec((int)c->prm.sdig_code, &c->spec)) { delete c; return LCPC_ERR_ARG; }
    uint64_t npr = p->n_per_row;
    if (!(p->n_per_row && p->n_cols)) {
      if (!sdig_n_per_row(*f, p->n_coeffs, (int)c->prm.sdig_code, &npr)) { delete c; return LCPC_ERR_ARG; }
    }
    std::vector<CsrMatrix> pre, post;
    if (!sdig_generate(*f, c->spec, npr, p->seed, pre, post, c->pre_dims, c->post_dims)) { delete c; return LCPC_ERR_DIMS; }
    c->n_per_row = npr;
    c->n_cols = sdig_codeword_length(c->pre_dims, c->post_dims);
    if (p->n_per_row && p->n_cols && p->n_cols != c->n_cols) { delete c; return LCPC_ERR_DIMS; }   // new_from_dims assert
    auto upload = [&](const CsrMatrix& m, DevCsr& d) -> int {
      d.n_in = m.n_in; d.n_out = m.n_out;
      int r;
      if ((r = dev_alloc(c, &d.rowptr, m.rowptr.size() * 4))) return r;
      if ((r = dev_alloc(c, &d.colidx, m.colidx.size() * 4))) return r;
      if ((r = dev_alloc(c, &d.vals, m.vals.size() * 8))) return r;
      HIPCHK(c, hipMemcpy(d.rowptr, m.rowptr.data(), m.rowptr.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(d.colidx, m.colidx.data(), m.colidx.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(d.vals, m.vals.data(), m.vals.size() * 8, hipMemcpyHostToDevice));
      if (f->L == 4) {
        const size_t nnz = m.colidx.size();
        std::vector<uint32_t> v29(nnz * 12 + 12);
        for (size_t k = 0; k < nnz; k++) to_r29(*f, &m.vals[k * 4], &v29[k * 12]);
        if ((r = dev_alloc(c, &d.vals29, v29.size() * 4))) return r;
        HIPCHK(c, hipMemcpy(d.vals29, v29.data(), v29.size() * 4, hipMemcpyHostToDevice));
      }
      return 0;
    };
    c->d_pre.resize(pre.size());
    c->d_post.resize(post.size());
    for (size_t i = 0; i < pre.size() && !rc; i++) { rc = upload(pre[i], c->d_pre[i]); if (!rc) rc = upload(post[i], c->d_post[i]); }
    if (!rc) rc = dev_alloc(c, &c->d_r2, 8 * f->L);
    if (!rc && hipMemcpy(c->d_r2, f->r2, 8 * f->L, hipMemcpyHostToDevice) != hipSuccess) rc = LCPC_ERR_HIP;
    if (rc) { lcpc_ctx_destroy(c); return rc; }
  } else {
    delete c;
    return LCPC_ERR_ARG;
  }
  c->np2 = next_pow2(c->n_cols);
  c->path_len = (uint32_t)log2_ceil(c->n_cols);
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) { lcpc_ctx_destroy(c); return LCPC_ERR_HIP; }
  *out = c;
  return 0;
}

void lcpc_ctx_destroy(lcpc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->prm.device);
  dev_free(c->d_roots); dev_free(c->d_roots29); dev_free(c->d_roots29c); dev_free(c->d_qp29); dev_free(c->d_r2); dev_free(c->d_tmp); dev_free(c->d_t);
  for (auto* v : {&c->d_pre, &c->d_post})
    for (auto& d : *v) { dev_free(d.rowptr); dev_free(d.colidx); dev_free(d.vals); dev_free(d.vals29); }
  dev_free(c->d_coeffs); dev_free(c->d_comm); dev_free(c->d_hashes); dev_free(c->d_cvs); dev_free(c->d_scratch); dev_free(c->d_node_tab); dev_free(c->d_t29);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : c->ev_batch) if (e) (void)hipEventDestroy(e);
  if (c->s_copy) (void)hipStreamDestroy(c->s_copy);
  if (c->s_comp) (void)hipStreamDestroy(c->s_comp);
  delete c;
}

int lcpc_get_dims(const lcpc_ctx* c, uint64_t len, uint64_t* nr, uint64_t* np, uint64_t* nc) {
  if (!c || len == 0) return LCPC_ERR_ARG;
  if (nr) *nr = (len + c->n_per_row - 1) / c->n_per_row;
  if (np) *np = c->n_per_row;
  if (nc) *nc = c->n_cols;
  return 0;
}
int lcpc_dims_ok(const lcpc_ctx* c, uint64_t n_per_row, uint64_t n_cols) {
  if (!c) return 0;
  bool ok = n_per_row < n_cols && n_per_row == c->n_per_row && n_cols == c->n_cols;
  if (c->prm.encoding == LCPC_ENC_LIGERO) ok = ok && (n_cols & (n_cols - 1)) == 0;
  return ok ? 1 : 0;
}
uint64_t lcpc_get_n_col_opens(const lcpc_ctx* c) {
  if (!c) return 0;
  return c->prm.encoding == LCPC_ENC_LIGERO ? ligero_n_col_opens(c->prm.rho_num, c->prm.rho_den) : sdig_n_col_opens((int)c->prm.sdig_code);
}
uint64_t lcpc_get_n_degree_tests(const lcpc_ctx* c) { return c ? n_degree_tests(128, c->n_cols, c->f->flog2()) : 0; }
uint32_t lcpc_field_limbs(const lcpc_ctx* c) { return c ? (uint32_t)c->L : 0; }

int lcpc_encode_rows(lcpc_ctx* c, uint64_t* rows, uint64_t n_rows) {
  if (!c || !rows) return LCPC_ERR_ARG;
  if (n_rows == 0) return 0;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  const size_t bytes = (size_t)n_rows * c->n_cols * elem_bytes(c);
  int rc = ensure_scratch(c, bytes);
  if (rc) return rc;
  if (c->prm.encoding == LCPC_ENC_SDIG) {
    const uint64_t need = n_rows * c->pre_dims.back().m;
    if (need > c->tmp_cap) {
      dev_free(c->d_tmp);
      if ((rc = dev_alloc(c, &c->d_tmp, (size_t)need * elem_bytes(c)))) return rc;
      c->tmp_cap = need;
    }
  }
  HIPCHK(c, hipMemcpy(c->d_scratch, rows, bytes, hipMemcpyHostToDevice));
  // the trait contract (lib.rs:651-652): entries >= n_per_row are zero on entry; they are read as given here
  if ((rc = encode_rows_device(c, c->d_scratch, c->n_cols, c->n_cols, c->d_scratch, n_rows, nullptr))) return rc;
  HIPCHK(c, hipMemcpy(rows, c->d_scratch, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int lcpc_commit_device(lcpc_ctx* c, const uint64_t* coeffs_dev, uint64_t n_coeffs, void* stream, uint8_t* root) {
  if (!c || !coeffs_dev) return LCPC_ERR_ARG;
  if (c->prm.shard_count > 1) return LCPC_ERR_STATE;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  hipStream_t st = (hipStream_t)stream;
  int rc = set_rows(c, n_coeffs);
  if (rc) return rc;
  const size_t eb = elem_bytes(c);
  if (c->prm.encoding == LCPC_ENC_LIGERO || c->n_rows >= 16) {
    // the padded local copy of coeffs (lib.rs:636-645; LcCommit keeps it for prove) is written by the first
    // NTT pass (Ligero) / the input transpose (Brakedown, >= 16 rows) while it streams the caller's buffer:
    // no separate D2D copy
    return commit_resident(c, st, root, reinterpret_cast<const uint32_t*>(coeffs_dev), n_coeffs);
  }
  HIPCHK(c, hipMemcpyAsync(c->d_coeffs, coeffs_dev, (size_t)n_coeffs * eb, hipMemcpyDeviceToDevice, st));
  const uint64_t padded = c->n_rows * c->n_per_row;
  if (padded > n_coeffs)
    HIPCHK(c, hipMemsetAsync(reinterpret_cast<uint8_t*>(c->d_coeffs) + (size_t)n_coeffs * eb, 0, (size_t)(padded - n_coeffs) * eb, st));
  return commit_resident(c, st, root);
}

int lcpc_commit(lcpc_ctx* c, const uint64_t* coeffs, uint64_t n_coeffs, uint8_t* root) {
  if (!c || !coeffs) return LCPC_ERR_ARG;
  if (c->prm.shard_count > 1) return LCPC_ERR_STATE;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  int rc = set_rows(c, n_coeffs);
  if (rc) return rc;
  const size_t eb = elem_bytes(c);
  const uint64_t padded = c->n_rows * c->n_per_row;
  const size_t total_bytes = (size_t)n_coeffs * eb;
  // Small inputs, Brakedown (whole-matrix transposes) and timing runs: one copy, then the resident path.
  if (c->prm.encoding != LCPC_ENC_LIGERO || total_bytes < ((size_t)64 << 20) || c->n_rows < 16 || c->timing) {
    HIPCHK(c, hipMemcpyAsync(c->d_coeffs, coeffs, total_bytes, hipMemcpyHostToDevice, nullptr));
    if (padded > n_coeffs)
      HIPCHK(c, hipMemsetAsync(reinterpret_cast<uint8_t*>(c->d_coeffs) + total_bytes, 0, (size_t)(padded - n_coeffs) * eb, nullptr));
    return commit_resident(c, nullptr, root);
  }
  // Large Ligero commit from host memory: rows are independent (lcpc-2d lib.rs:648-653), so the matrix is
  // uploaded in row batches on a copy stream while the previous batch runs its NTT passes on a compute stream;
  // column hashing starts once the last batch is encoded.  The end-to-end time tends to the PCIe time.
  if (!c->s_copy) HIPCHK(c, hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking));
  if (!c->s_comp) HIPCHK(c, hipStreamCreateWithFlags(&c->s_comp, hipStreamNonBlocking));
  constexpr int NB = 16;
  for (auto& e : c->ev_batch)
    if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(c, hipDeviceSynchronize());                       // earlier work of this context (null stream) is done
  if (padded > n_coeffs)
    HIPCHK(c, hipMemsetAsync(reinterpret_cast<uint8_t*>(c->d_coeffs) + total_bytes, 0, (size_t)(padded - n_coeffs) * eb, c->s_copy));
  c->launches[0] = c->launches[1] = c->launches[2] = 0;
  const uint64_t rows_per = (c->n_rows + NB - 1) / NB;
  for (int b = 0; b < NB; b++) {
    const uint64_t r0 = (uint64_t)b * rows_per;
    if (r0 >= c->n_rows) break;
    const uint64_t r1 = r0 + rows_per < c->n_rows ? r0 + rows_per : c->n_rows;
    const uint64_t e0 = r0 * c->n_per_row, e1 = r1 * c->n_per_row < n_coeffs ? r1 * c->n_per_row : n_coeffs;
    if (e1 > e0)
      HIPCHK(c, hipMemcpyAsync(reinterpret_cast<uint8_t*>(c->d_coeffs) + (size_t)e0 * eb, reinterpret_cast<const uint8_t*>(coeffs) + (size_t)e0 * eb,
                               (size_t)(e1 - e0) * eb, hipMemcpyHostToDevice, c->s_copy));
    HIPCHK(c, hipEventRecord(c->ev_batch[b], c->s_copy));
    HIPCHK(c, hipStreamWaitEvent(c->s_comp, c->ev_batch[b], 0));
    rc = encode_rows_device(c, c->d_coeffs + (size_t)r0 * c->n_per_row * c->NL, c->n_per_row, c->n_per_row,
                            c->d_comm + (size_t)r0 * c->n_cols * c->NL, r1 - r0, c->s_comp, ~(uint64_t)0, nullptr, c->comm_canon);
    if (rc) return rc;
  }
  if ((rc = merkleize_device(c, c->s_comp))) return rc;
  c->committed = true;
  if (root) HIPCHK(c, hipMemcpyAsync(root, c->d_hashes + (2 * c->np2 - 2) * 8, 32, hipMemcpyDeviceToHost, c->s_comp));
  HIPCHK(c, hipStreamSynchronize(c->s_comp));              // later calls use the null stream / caller streams
  return 0;
}

int lcpc_commit_from_parts(lcpc_ctx* c, const uint64_t* comm, const uint64_t* coeffs, uint64_t n_rows, uint8_t* root) {
  if (!c || !comm || n_rows == 0) return LCPC_ERR_ARG;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  c->n_rows = n_rows; c->row_begin = 0; c->n_rows_local = n_rows;
  int rc = ensure_buffers(c, n_rows);
  if (rc) return rc;
  const size_t eb = elem_bytes(c);
  c->comm_t = false;
  HIPCHK(c, hipMemcpy(c->d_comm, comm, (size_t)n_rows * c->n_cols * eb, hipMemcpyHostToDevice));
  if (c->comm_canon) HIPCHK(c, launch_to_canon(c->NL, c->d_comm, n_rows * c->n_cols, c->d_comm, nullptr));
  if (coeffs) HIPCHK(c, hipMemcpy(c->d_coeffs, coeffs, (size_t)n_rows * c->n_per_row * eb, hipMemcpyHostToDevice));
  else HIPCHK(c, hipMemset(c->d_coeffs, 0, (size_t)n_rows * c->n_per_row * eb));
  c->launches[0] = c->launches[1] = c->launches[2] = 0;
  if (c->timing) { HIPCHK(c, hipEventRecord(c->ev[0], nullptr)); HIPCHK(c, hipEventRecord(c->ev[1], nullptr)); }
  if ((rc = merkleize_device(c, nullptr))) return rc;
  if ((rc = finish_timing(c, nullptr))) return rc;
  c->committed = true;
  if (root) HIPCHK(c, hipMemcpy(root, c->d_hashes + (2 * c->np2 - 2) * 8, 32, hipMemcpyDeviceToHost));
  return 0;
}

int lcpc_get_root(lcpc_ctx* c, uint8_t root[32]) {
  if (!c || !root) return LCPC_ERR_ARG;
  if (!c->committed) return LCPC_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->prm.device));
  HIPCHK(c, hipMemcpy(root, c->d_hashes + (2 * c->np2 - 2) * 8, 32, hipMemcpyDeviceToHost));
  return 0;
}
int lcpc_commit_dims(const lcpc_ctx* c, uint64_t* nr, uint64_t* np, uint64_t* nc, uint64_t* nh) {
  if (!c) return LCPC_ERR_ARG;
  if (!c->committed) return LCPC_ERR_STATE;
  if (nr) *nr = c->n_rows;
  if (np) *np = c->n_per_row;
  if (nc) *nc = c->n_cols;
  if (nh) *nh = 2 * c->np2 - 1;
  return 0;
}
int lcpc_get_hashes(lcpc_ctx* c, uint8_t* hashes) {
  if (!c || !hashes) return LCPC_ERR_ARG;
  if (!c->committed) return LCPC_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->prm.device));
  HIPCHK(c, hipMemcpy(hashes, c->d_hashes, (size_t)(2 * c->np2 - 1) * 32, hipMemcpyDeviceToHost));
  return 0;
}
int lcpc_get_comm(lcpc_ctx* c, uint64_t row0, uint64_t n, uint64_t* out) {
  if (!c || !out) return LCPC_ERR_ARG;
  if (!c->committed) return LCPC_ERR_STATE;
  if (row0 < c->row_begin || row0 + n > c->row_begin + c->n_rows_local) return LCPC_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->prm.device));
  const size_t eb = elem_bytes(c);
  if (c->comm_t && !c->comm_rows_valid) {      // Brakedown: the commitment is position-major; make the row-major view once
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, launch_transpose_from_t(c->NL, c->d_t, c->n_cols, c->n_rows_local, c->d_comm, c->n_cols, nullptr));
    HIPCHK(c, hipStreamSynchronize(nullptr));
    c->comm_rows_valid = true;
  }
  const uint32_t* src = c->d_comm + (size_t)(row0 - c->row_begin) * c->n_cols * c->NL;
  if (!c->comm_canon) {
    HIPCHK(c, hipMemcpy(out, src, (size_t)n * c->n_cols * eb, hipMemcpyDeviceToHost));
    return 0;
  }
  // canonical on the device, Montgomery form (as ff_derive stores elements) at the ABI: convert a row batch at a time
  const uint64_t batch = std::max<uint64_t>(1, ((uint64_t)64 << 20) / (c->n_cols * eb));
  uint32_t* tmp = nullptr;
  int rc = dev_alloc(c, &tmp, (size_t)std::min(batch, n ? n : 1) * c->n_cols * eb);
  if (rc) return rc;
  for (uint64_t r = 0; r < n; r += batch) {
    const uint64_t nb = std::min(batch, n - r);
    hipError_t he = launch_to_mont(c->NL, src + (size_t)r * c->n_cols * c->NL, nb * c->n_cols, c->d_r2, tmp, nullptr);
    if (he == hipSuccess)
      he = hipMemcpy(reinterpret_cast<uint8_t*>(out) + (size_t)r * c->n_cols * eb, tmp, (size_t)nb * c->n_cols * eb, hipMemcpyDeviceToHost);
    if (he != hipSuccess) { dev_free(tmp); return fail_hip(c, he, "lcpc_get_comm"); }
  }
  dev_free(tmp);
  return 0;
}
int lcpc_get_coeffs(lcpc_ctx* c, uint64_t row0, uint64_t n, uint64_t* out) {
  if (!c || !out) return LCPC_ERR_ARG;
  if (!c->committed) return LCPC_ERR_STATE;
  if (row0 < c->row_begin || row0 + n > c->row_begin + c->n_rows_local) return LCPC_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->prm.device));
  const size_t eb = elem_bytes(c);
  HIPCHK(c, hipMemcpy(out, reinterpret_cast<uint8_t*>(c->d_coeffs) + (size_t)(row0 - c->row_begin) * c->n_per_row * eb,
                      (size_t)n * c->n_per_row * eb, hipMemcpyDeviceToHost));
  return 0;
}

// ---- collapse / open ---------------------------------------------------------------------------------
// split the row range so that the grid has >= ~2k workgroups even for narrow matrices
static uint32_t collapse_splits(const lcpc_ctx* c) {
  const uint64_t col_blocks = (c->n_per_row + 255) / 256;
  uint32_t n_splits = 1;
  while (n_splits < 64 && col_blocks * n_splits < 2048 && c->n_rows_local / (n_splits * 2) >= 16) n_splits *= 2;
  return n_splits;
}
static int collapse_local(lcpc_ctx* c, const uint32_t* d_tensors, uint32_t n_tensors, hipStream_t st, uint32_t* d_out) {
  const uint32_t n_splits = collapse_splits(c);
  CollapseArgs a{};
  a.coeffs = c->d_coeffs; a.tensors = d_tensors; a.n_rows = c->n_rows_local; a.n_per_row = c->n_per_row;
  a.n_tensors = n_tensors; a.n_splits = n_splits;
  if (c->NL == 8) {
    const uint64_t ne = (uint64_t)n_tensors * c->n_rows_local;
    if (ne > c->t29_cap) {
      dev_free(c->d_t29);
      int rc = dev_alloc(c, &c->d_t29, (size_t)ne * 48);
      if (rc) return rc;
      c->t29_cap = ne;
    }
    HIPCHK(c, launch_to_r29(d_tensors, ne, c->d_t29, st));
    a.tensors29 = c->d_t29;
  }
  if (n_splits == 1) {
    a.out = d_out;
    HIPCHK(c, launch_collapse(c->NL, a, st));
    return 0;
  }
  const size_t eb = elem_bytes(c);
  const size_t part_bytes = (size_t)n_splits * n_tensors * c->n_per_row * eb;
  // partials live at the end of scratch (callers reserve it)
  uint32_t* d_part = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(c->d_scratch) + c->scratch_cap - part_bytes);
  a.out = d_part;
  HIPCHK(c, launch_collapse(c->NL, a, st));
  HIPCHK(c, launch_field_sum(c->NL, d_part, n_splits, (uint64_t)n_tensors * c->n_per_row, d_out, st));
  return 0;
}
static size_t collapse_scratch_bytes(const lcpc_ctx* c, uint32_t n_tensors) {
  return (size_t)collapse_splits(c) * n_tensors * c->n_per_row * elem_bytes(c);
}

// scratch layout for collapse: [tensors (host entry only)] [polys (host entry only)] ... [partials at the end]
static int collapse_run(lcpc_ctx* c, const uint32_t* d_tensors, uint32_t n_tensors, hipStream_t st, uint32_t* d_polys) {
  for (uint32_t t = 0; t < n_tensors; t += 2) {
    const uint32_t nt = (n_tensors - t) >= 2 ? 2 : 1;
    int rc = collapse_local(c, d_tensors + (size_t)t * c->n_rows_local * c->NL, nt, st, d_polys + (size_t)t * c->n_per_row * c->NL);
    if (rc) return rc;
  }
  return 0;
}

int lcpc_collapse_device(lcpc_ctx* c, const uint64_t* tensors_dev, uint32_t n_tensors, void* stream, uint64_t* polys_dev) {
  if (!c || !tensors_dev || !polys_dev || n_tensors == 0) return LCPC_ERR_ARG;
  if (!c->committed) return LCPC_ERR_STATE;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  int rc = ensure_scratch(c, collapse_scratch_bytes(c, 2));
  if (rc) return rc;
  return collapse_run(c, reinterpret_cast<const uint32_t*>(tensors_dev), n_tensors, (hipStream_t)stream,
                      reinterpret_cast<uint32_t*>(polys_dev));
}

int lcpc_collapse(lcpc_ctx* c, const uint64_t* tensors, uint32_t n_tensors, uint64_t* polys) {
  if (!c || !tensors || !polys || n_tensors == 0) return LCPC_ERR_ARG;
  if (!c->committed) return LCPC_ERR_STATE;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  const size_t eb = elem_bytes(c);
  const size_t tb = ((size_t)n_tensors * c->n_rows_local * eb + 255) & ~(size_t)255;
  const size_t pb = ((size_t)n_tensors * c->n_per_row * eb + 255) & ~(size_t)255;
  int rc = ensure_scratch(c, tb + pb + collapse_scratch_bytes(c, 2) + 256);
  if (rc) return rc;
  uint8_t* base = reinterpret_cast<uint8_t*>(c->d_scratch);
  uint32_t* d_t = reinterpret_cast<uint32_t*>(base);
  uint32_t* d_p = reinterpret_cast<uint32_t*>(base + tb);
  HIPCHK(c, hipMemcpyAsync(d_t, tensors, (size_t)n_tensors * c->n_rows_local * eb, hipMemcpyHostToDevice, nullptr));
  if ((rc = collapse_run(c, d_t, n_tensors, nullptr, d_p))) return rc;
  HIPCHK(c, hipMemcpy(polys, d_p, (size_t)n_tensors * c->n_per_row * eb, hipMemcpyDeviceToHost));
  return 0;
}

int lcpc_field_sum_device(lcpc_ctx* c, const uint64_t* parts, uint32_t n_parts, uint64_t n_elems, void* stream, uint64_t* out) {
  if (!c || !parts || !out || n_parts == 0) return LCPC_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->prm.device));
  HIPCHK(c, launch_field_sum(c->NL, reinterpret_cast<const uint32_t*>(parts), n_parts, n_elems, reinterpret_cast<uint32_t*>(out),
                             (hipStream_t)stream));
  return 0;
}

int lcpc_open_columns(lcpc_ctx* c, const uint64_t* cols, uint32_t n, uint64_t* col_vals, uint8_t* paths) {
  if (!c || !cols || (!col_vals && !paths)) return LCPC_ERR_ARG;
  if (!c->committed) return LCPC_ERR_STATE;
  for (uint32_t i = 0; i < n; i++)
    if (cols[i] >= c->n_cols) return LCPC_ERR_COLUMN_NUMBER;        // lib.rs:797-799
  if (n == 0) return 0;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  const size_t eb = elem_bytes(c);
  const size_t vb = (size_t)n * c->n_rows_local * eb, pb = (size_t)n * c->path_len * 32, cb = (size_t)n * 8;
  int rc = ensure_scratch(c, vb + pb + cb + 64);
  if (rc) return rc;
  uint8_t* base = reinterpret_cast<uint8_t*>(c->d_scratch);
  uint64_t* d_cols = reinterpret_cast<uint64_t*>(base);
  uint32_t* d_vals = reinterpret_cast<uint32_t*>(base + ((cb + 31) & ~(size_t)31));
  uint32_t* d_paths = reinterpret_cast<uint32_t*>(base + ((cb + 31) & ~(size_t)31) + ((vb + 31) & ~(size_t)31));
  HIPCHK(c, hipMemcpy(d_cols, cols, cb, hipMemcpyHostToDevice));
  if (col_vals) {
    if (c->comm_t)
      HIPCHK(c, launch_gather_columns(c->NL, c->d_t, c->n_rows_local, 1, c->n_rows_local, d_cols, n, d_vals, nullptr, nullptr));
    else
      HIPCHK(c, launch_gather_columns(c->NL, c->d_comm, c->n_rows_local, c->n_cols, 1, d_cols, n, d_vals, c->comm_canon ? c->d_r2 : nullptr, nullptr));
    HIPCHK(c, hipMemcpy(col_vals, d_vals, vb, hipMemcpyDeviceToHost));
  }
  if (paths && c->path_len) {
    HIPCHK(c, launch_gather_paths(c->d_hashes, c->np2, c->path_len, d_cols, n, d_paths, nullptr));
    HIPCHK(c, hipMemcpy(paths, d_paths, pb, hipMemcpyDeviceToHost));
  }
  return 0;
}

// ---- transcript ------------------------------------------------------------------------------------------
lcpc_transcript* lcpc_transcript_new(const uint8_t* label, size_t len) { return new lcpc_transcript(label, len); }
lcpc_transcript* lcpc_transcript_clone(const lcpc_transcript* t) { return t ? new lcpc_transcript(*t) : nullptr; }
void lcpc_transcript_append_message(lcpc_transcript* t, const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen) {
  if (t) t->t.append_message(label, llen, msg, mlen);
}
void lcpc_transcript_challenge_bytes(lcpc_transcript* t, const uint8_t* label, size_t llen, uint8_t* out, size_t n) {
  if (t) t->t.challenge_bytes(label, llen, out, n);
}
void lcpc_transcript_free(lcpc_transcript* t) { delete t; }

// ---- prove (lib.rs:1004-1093) ----------------------------------------------------------------------------
static void absorb_poly(Transcript& tr, const uint8_t* label, const FieldDesc& f, const uint64_t* poly, uint64_t n) {
  // to_repr (Montgomery -> canonical little-endian, lib.rs:47-57) is independent per element: done in parallel;
  // only the STROBE absorb itself is serial (lib.rs:1045-1047)
  const int L = f.L;
  std::vector<uint64_t> canon(n * L);
  parallel_for(n, 4096, [&](uint64_t b, uint64_t e) { for (uint64_t i = b; i < e; i++) h_canon(f, &canon[i * L], poly + i * L); });
  for (uint64_t i = 0; i < n; i++) tr.append_message(label, 6, reinterpret_cast<const uint8_t*>(&canon[i * L]), 8 * L);
}

// The exchange of a row-sharded prove: every rank contributes `bytes` from send_dev, receives all
// ranks' blocks in rank order in recv_dev.
namespace {
struct ShardXchg {
  uint8_t *send_dev, *recv_dev;
  uint64_t max_bytes;
  lcpc_allgather_fn fn;
  void* user;
};
// collapse over ALL rows of a sharded commitment: local partial sums, all-gather, sum mod p (lib.rs:1095-1123 split by rows)
int collapse_sharded(lcpc_ctx* c, const ShardXchg& x, const uint64_t* tensors_full, uint32_t nt, uint64_t* polys) {
  const size_t eb = elem_bytes(c);
  const int L = c->L;
  const uint64_t bytes = (uint64_t)nt * c->n_per_row * eb;
  if (bytes > x.max_bytes) return LCPC_ERR_ARG;
  {
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->prm.device));
    if (c->n_rows_local == 0) {
      HIPCHK(c, hipMemsetAsync(x.send_dev, 0, bytes, nullptr));
    } else {
      std::vector<uint64_t> loc((size_t)nt * c->n_rows_local * L);
      for (uint32_t t = 0; t < nt; t++)
        memcpy(&loc[(size_t)t * c->n_rows_local * L], tensors_full + ((size_t)t * c->n_rows + c->row_begin) * L, c->n_rows_local * eb);
      const size_t tb = (loc.size() * 8 + 255) & ~(size_t)255;
      int rc = ensure_scratch(c, tb + collapse_scratch_bytes(c, 2) + 256);
      if (rc) return rc;
      uint32_t* d_t = c->d_scratch;
      HIPCHK(c, hipMemcpyAsync(d_t, loc.data(), loc.size() * 8, hipMemcpyHostToDevice, nullptr));
      if ((rc = collapse_run(c, d_t, nt, nullptr, reinterpret_cast<uint32_t*>(x.send_dev)))) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(nullptr));
  }
  if (x.fn(x.user, bytes) != 0) return LCPC_ERR_XCHG;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  HIPCHK(c, launch_field_sum(c->NL, reinterpret_cast<const uint32_t*>(x.recv_dev), c->prm.shard_count, (uint64_t)nt * c->n_per_row,
                             reinterpret_cast<uint32_t*>(x.send_dev), nullptr));
  HIPCHK(c, hipMemcpy(polys, x.send_dev, bytes, hipMemcpyDeviceToHost));
  return 0;
}
// open_column for n columns of a sharded commitment: every rank gathers its rows, one all-gather, columns assembled in
// row order on the host; the Merkle paths come from the (replicated) tree
int open_sharded(lcpc_ctx* c, const ShardXchg& x, const uint64_t* cols, uint32_t n, uint64_t* vals, uint8_t* paths) {
  const size_t eb = elem_bytes(c);
  const uint32_t G = c->prm.shard_count;
  std::vector<uint64_t> rb(G), re(G);
  uint64_t max_rows = 0;
  for (uint32_t g = 0; g < G; g++) {
    uint64_t cb, ce, nch;
    shard_layout_of(c, g, c->n_rows, &rb[g], &re[g], &cb, &ce, &nch);
    max_rows = std::max(max_rows, re[g] - rb[g]);
  }
  const uint64_t bytes = (uint64_t)n * max_rows * eb;
  if (bytes > x.max_bytes) return LCPC_ERR_ARG;
  std::vector<uint64_t> loc((size_t)n * std::max<uint64_t>(c->n_rows_local, 1) * c->L);
  int rc = lcpc_open_columns(c, cols, n, c->n_rows_local ? loc.data() : nullptr, paths);      // local rows + paths
  if (rc) return rc;
  {
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->prm.device));
    if (c->n_rows_local) HIPCHK(c, hipMemcpy(x.send_dev, loc.data(), (size_t)n * c->n_rows_local * eb, hipMemcpyHostToDevice));
  }
  if (x.fn(x.user, bytes) != 0) return LCPC_ERR_XCHG;
  std::vector<uint8_t> all((size_t)G * bytes);
  {
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->prm.device));
    HIPCHK(c, hipMemcpy(all.data(), x.recv_dev, all.size(), hipMemcpyDeviceToHost));
  }
  for (uint32_t g = 0; g < G; g++) {
    const uint64_t nr_g = re[g] - rb[g];
    for (uint32_t k = 0; k < n && nr_g; k++)       // rank g's block: [k][its rows], contiguous
      memcpy(reinterpret_cast<uint8_t*>(vals) + ((size_t)k * c->n_rows + rb[g]) * eb, &all[(size_t)g * bytes + (size_t)k * nr_g * eb], nr_g * eb);
  }
  return 0;
}
}  // namespace

static int prove_impl(lcpc_ctx* c, const uint64_t* outer, uint64_t n_outer, lcpc_transcript* trw, uint8_t** proof, uint64_t* proof_len,
                      uint64_t* cols_opened, const ShardXchg* xchg) {
  if (!c || !outer || !trw || !proof || !proof_len) return LCPC_ERR_ARG;
  if (!c->committed) return LCPC_ERR_STATE;
  if ((c->prm.shard_count > 1) != (xchg != nullptr)) return LCPC_ERR_STATE;   // sharded contexts prove through lcpc_prove_sharded
  auto collapse = [&](const uint64_t* tensors, uint32_t nt, uint64_t* polys) -> int {
    return xchg ? collapse_sharded(c, *xchg, tensors, nt, polys) : lcpc_collapse(c, tensors, nt, polys);
  };
  if (!lcpc_dims_ok(c, c->n_per_row, c->n_cols)) return LCPC_ERR_COMMIT;      // check_comm lib.rs:1015
  if (n_outer != c->n_rows) return LCPC_ERR_OUTER_TENSOR;                     // lib.rs:1016-1018
  const FieldDesc& f = *c->f;
  const int L = f.L;
  Transcript& tr = trw->t;
  const uint64_t n_deg = lcpc_get_n_degree_tests(c), n_open = lcpc_get_n_col_opens(c);
  const uint64_t np = c->n_per_row, nr = c->n_rows;
  const bool dbg = getenv("LCPC_DEBUG_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tp[8] = {now(), 0, 0, 0, 0, 0, 0, 0};
  double t_collapse = 0, t_absorb = 0;
  std::vector<uint64_t> tensors(2 * nr * L), polys(2 * np * L), p_eval(np * L);
  std::vector<std::vector<uint64_t>> p_random(n_deg);
  bool have_eval = false;
  for (uint64_t i = 0; i < n_deg; i++) {                                      // lib.rs:1024-1050
    uint8_t key[32];
    tr.challenge_bytes(LBL_DT, 6, key, 32);
    ChaCha20Rng rng(key);
    for (uint64_t r = 0; r < nr; r++) rng.field_random(f, &tensors[r * L]);
    uint32_t nt = 1;
    if (i == 0) {   // the eval tensor is independent of the transcript: fuse it into the first pass over coeffs
      memcpy(&tensors[nr * L], outer, nr * L * 8);
      nt = 2;
    }
    double t0 = now();
    int rc = collapse(tensors.data(), nt, polys.data());
    if (rc) return rc;
    t_collapse += now() - t0;
    p_random[i].assign(polys.begin(), polys.begin() + np * L);
    if (nt == 2) { memcpy(p_eval.data(), &polys[np * L], np * L * 8); have_eval = true; }
    t0 = now();
    absorb_poly(tr, LBL_PR, f, p_random[i].data(), np);
    t_absorb += now() - t0;
  }
  if (!have_eval) {                                                           // lib.rs:1053-1064
    int rc = collapse(outer, 1, p_eval.data());
    if (rc) return rc;
  }
  tp[1] = now();
  absorb_poly(tr, LBL_PE, f, p_eval.data(), np);                              // lib.rs:1066-1068
  tp[2] = now();
  uint8_t key[32];
  tr.challenge_bytes(LBL_CO, 6, key, 32);                                     // lib.rs:1071-1080
  ChaCha20Rng rng(key);
  std::vector<uint64_t> cols(n_open);
  for (auto& x : cols) x = rng.uniform(c->n_cols);
  if (cols_opened) memcpy(cols_opened, cols.data(), n_open * 8);
  // (uninitialised buffers: a Brakedown proof opens 6593 columns, tens of MB that are overwritten anyway)
  std::unique_ptr<uint64_t[]> vals(new uint64_t[(size_t)n_open * nr * L + 1]);
  std::unique_ptr<uint8_t[]> paths(new uint8_t[(size_t)n_open * c->path_len * 32 + 32]);
  tp[3] = now();
  int rc = xchg ? open_sharded(c, *xchg, cols.data(), (uint32_t)n_open, vals.get(), paths.get())
                : lcpc_open_columns(c, cols.data(), (uint32_t)n_open, vals.get(), paths.get());   // lib.rs:1081-1084
  if (rc) return rc;
  tp[4] = now();
  // bincode 1.3 of WrappedLcEvalProof (lib.rs:550-560): n_cols, p_eval, p_random_vec, columns -- the size is known
  // up front, so the proof is written once, straight into the buffer the caller receives
  const size_t pbytes = np * L * 8;
  const size_t total = 8 + (8 + pbytes) + 8 + n_deg * (8 + pbytes) + 8 + n_open * (8 + nr * L * 8 + 8 + (size_t)c->path_len * 40);
  uint8_t* out = static_cast<uint8_t*>(malloc(total ? total : 1));
  if (!out) return LCPC_ERR_NOMEM;
  uint8_t* w = out;
  auto w64 = [&](uint64_t v) { memcpy(w, &v, 8); w += 8; };
  auto wbytes = [&](const void* d, size_t n) { memcpy(w, d, n); w += n; };
  w64(c->n_cols);
  w64(np); wbytes(p_eval.data(), pbytes);
  w64(n_deg);
  for (auto& pr : p_random) { w64(np); wbytes(pr.data(), pbytes); }
  w64(n_open);
  for (uint64_t k = 0; k < n_open; k++) {
    w64(nr); wbytes(&vals[k * nr * L], nr * L * 8);
    w64(c->path_len);
    for (uint32_t l = 0; l < c->path_len; l++) { w64(32); wbytes(&paths[((size_t)k * c->path_len + l) * 32], 32); }
  }
  if ((size_t)(w - out) != total) { free(out); return LCPC_ERR_STATE; }
  *proof = out; *proof_len = total;
  if (dbg)
    fprintf(stderr, "[lcpc_prove] collapse %.2f ms, absorb p_random %.2f, absorb p_eval %.2f, challenges+alloc %.2f, open %.2f, bincode %.2f, total %.2f\n",
            t_collapse, t_absorb, tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], now() - tp[4], now() - tp[0]);
  return 0;
}

int lcpc_prove(lcpc_ctx* c, const uint64_t* outer, uint64_t n_outer, lcpc_transcript* trw, uint8_t** proof, uint64_t* proof_len,
               uint64_t* cols_opened) {
  return prove_impl(c, outer, n_outer, trw, proof, proof_len, cols_opened, nullptr);
}

uint64_t lcpc_prove_sharded_bytes(const lcpc_ctx* c, uint64_t n_rows_total) {
  if (!c || n_rows_total == 0) return 0;
  const uint32_t G = c->prm.shard_count > 1 ? c->prm.shard_count : 1;
  uint64_t max_rows = 0;
  for (uint32_t g = 0; g < G; g++) {
    uint64_t rb, re, cb, ce, nch;
    shard_layout_of(c, g, n_rows_total, &rb, &re, &cb, &ce, &nch);
    max_rows = std::max(max_rows, re - rb);
  }
  const uint64_t eb = elem_bytes(c);
  return std::max<uint64_t>(2 * c->n_per_row * eb, lcpc_get_n_col_opens(c) * max_rows * eb);
}

int lcpc_prove_sharded(lcpc_ctx* c, const uint64_t* outer, uint64_t n_outer, lcpc_transcript* trw, uint8_t* send_dev, uint8_t* recv_dev,
                       uint64_t max_bytes, lcpc_allgather_fn fn, void* user, uint8_t** proof, uint64_t* proof_len,
                       uint64_t* cols_opened) {
  if (!c || !send_dev || !recv_dev || !fn) return LCPC_ERR_ARG;
  if (c->prm.shard_count <= 1) return LCPC_ERR_STATE;
  if (c->committed && max_bytes < lcpc_prove_sharded_bytes(c, c->n_rows)) return LCPC_ERR_ARG;
  const ShardXchg x{send_dev, recv_dev, max_bytes, fn, user};
  return prove_impl(c, outer, n_outer, trw, proof, proof_len, cols_opened, &x);
}

// ---- verify (lib.rs:832-1000) ----------------------------------------------------------------------------
namespace {
struct Rd {
  const uint8_t* p; uint64_t len, pos = 0; bool bad = false;
  uint64_t u64_() { uint64_t v = 0; if (pos + 8 > len) { bad = true; return 0; } memcpy(&v, p + pos, 8); pos += 8; return v; }
  const uint8_t* take(uint64_t n) { if (n > len - pos) { bad = true; return nullptr; } const uint8_t* q = p + pos; pos += n; return q; }
};
void hash_column_host(const FieldDesc& f, const uint64_t* col, uint64_t n_rows, uint8_t out[32]) {
  std::vector<uint8_t> msg(32 + n_rows * 8 * f.L, 0);
  for (uint64_t r = 0; r < n_rows; r++) {
    uint64_t t[MAXL];
    h_canon(f, t, col + r * f.L);
    memcpy(&msg[32 + r * 8 * f.L], t, 8 * f.L);
  }
  blake3_host(msg.data(), msg.size(), out);
}
}  // namespace

int lcpc_verify(lcpc_ctx* c, const uint8_t root[32], const uint64_t* outer, uint64_t n_outer, const uint64_t* inner, uint64_t n_inner,
                const uint8_t* proof, uint64_t proof_len, lcpc_transcript* trw, uint64_t* eval_out) {
  if (!c || !root || !outer || !inner || !proof || !trw || !eval_out) return LCPC_ERR_ARG;
  const FieldDesc& f = *c->f;
  const int L = f.L;
  const uint64_t F = 8 * L;
  Transcript& tr = trw->t;
  const bool dbg = getenv("LCPC_DEBUG_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tv[6] = {now(), 0, 0, 0, 0, 0};
  // every field of the wire layout sits at a multiple of 8 bytes, so the vectors are read in place (a proof handed over
  // at an odd address is copied once)
  std::vector<uint64_t> realigned;
  if (reinterpret_cast<uintptr_t>(proof) % 8 != 0) {
    realigned.resize((proof_len + 7) / 8);
    memcpy(realigned.data(), proof, proof_len);
    proof = reinterpret_cast<const uint8_t*>(realigned.data());
  }
  Rd r{proof, proof_len};
  const uint64_t n_cols = r.u64_();
  const uint64_t n_per_row = r.u64_();
  if (r.bad || n_per_row > proof_len / F) return LCPC_VERR_MALFORMED;
  struct View { const uint64_t* p = nullptr; uint64_t n = 0; const uint64_t* data() const { return p; } uint64_t size() const { return n; } };
  View p_eval;
  { const uint8_t* q = r.take(n_per_row * F); if (!q) return LCPC_VERR_MALFORMED; p_eval = View{reinterpret_cast<const uint64_t*>(q), n_per_row * L}; }
  const uint64_t n_deg_pf = r.u64_();
  if (r.bad || n_deg_pf > 4096) return LCPC_VERR_MALFORMED;
  std::vector<View> p_random(n_deg_pf);
  for (auto& v : p_random) {
    const uint64_t l = r.u64_();
    if (r.bad || l > proof_len / F) return LCPC_VERR_MALFORMED;
    const uint8_t* q = r.take(l * F);
    if (!q) return LCPC_VERR_MALFORMED;
    v = View{reinterpret_cast<const uint64_t*>(q), l * L};
  }
  const uint64_t n_columns = r.u64_();
  if (r.bad || n_columns > proof_len / 8) return LCPC_VERR_MALFORMED;
  std::vector<View> cols(n_columns);
  struct PathView { const uint8_t* p = nullptr; uint64_t n = 0; };      // n entries of (u64 32, 32 bytes): digest k at p + 40 k + 8
  std::vector<PathView> paths(n_columns);
  for (uint64_t i = 0; i < n_columns; i++) {
    const uint64_t l = r.u64_();
    if (r.bad || l > proof_len / F) return LCPC_VERR_MALFORMED;
    const uint8_t* q = r.take(l * F);
    if (!q) return LCPC_VERR_MALFORMED;
    cols[i] = View{reinterpret_cast<const uint64_t*>(q), l * L};
    const uint64_t pl = r.u64_();
    if (r.bad || pl > proof_len / 40) return LCPC_VERR_MALFORMED;
    paths[i].p = proof + r.pos;
    paths[i].n = pl;
    for (uint64_t k = 0; k < pl; k++) {
      const uint64_t dl = r.u64_();
      const uint8_t* d = r.take(32);
      if (r.bad || dl != 32 || !d) return LCPC_VERR_MALFORMED;     // Output<D> is 32 bytes
    }
  }
  if (r.pos != proof_len) return LCPC_VERR_MALFORMED;
  const uint64_t n_col_opens = lcpc_get_n_col_opens(c);                        // lib.rs:845-860
  if (n_col_opens != n_columns || n_col_opens == 0) return LCPC_VERR_NUM_COL_OPENS;
  const uint64_t n_rows = cols[0].size() / L;
  if (n_inner != n_per_row) return LCPC_VERR_INNER_TENSOR;
  if (n_outer != n_rows) return LCPC_VERR_OUTER_TENSOR;
  if (!lcpc_dims_ok(c, n_per_row, n_cols)) return LCPC_VERR_ENCODING_DIMS;
  const uint64_t n_deg = lcpc_get_n_degree_tests(c);
  if (n_deg_pf < n_deg) return LCPC_VERR_MALFORMED;                            // reference indexes p_random_vec[i] (would panic)
  for (uint64_t i = 0; i < n_deg; i++) if (p_random[i].size() != n_per_row * L) return LCPC_VERR_MALFORMED;
  for (auto& cv : cols) if (cv.size() != n_rows * L) return LCPC_VERR_MALFORMED;
  // step 2 first: the 1 + n_deg row encodes (lib.rs:886, 918) depend only on the proof, not on the transcript, so they
  // run on the GPU (own thread: upload, kernels, download) while this thread does step 1, the serial transcript work
  std::vector<uint64_t> enc((n_deg + 1) * n_cols * L, 0);
  for (uint64_t i = 0; i < n_deg; i++) memcpy(&enc[i * n_cols * L], p_random[i].data(), n_per_row * F);
  memcpy(&enc[n_deg * n_cols * L], p_eval.data(), n_per_row * F);
  int enc_rc = 0;
  double t_enc = 0;
  tv[1] = now();
  std::thread enc_thread([&] { const double t0 = now(); enc_rc = lcpc_encode_rows(c, enc.data(), n_deg + 1); t_enc = now() - t0; });
  // step 1: random tensors, transcript (lib.rs:868-920)
  std::vector<std::vector<uint64_t>> rand_tensors(n_deg, std::vector<uint64_t>(n_rows * L));
  for (uint64_t i = 0; i < n_deg; i++) {
    uint8_t key[32];
    tr.challenge_bytes(LBL_DT, 6, key, 32);
    ChaCha20Rng rng(key);
    for (uint64_t k = 0; k < n_rows; k++) rng.field_random(f, &rand_tensors[i][k * L]);
    absorb_poly(tr, LBL_PR, f, p_random[i].data(), n_per_row);
  }
  absorb_poly(tr, LBL_PE, f, p_eval.data(), n_per_row);
  uint8_t key[32];
  tr.challenge_bytes(LBL_CO, 6, key, 32);
  ChaCha20Rng rng(key);
  tv[2] = now();
  enc_thread.join();
  tv[3] = now();
  if (enc_rc) return enc_rc == LCPC_ERR_ENCODE ? LCPC_VERR_ENCODE : enc_rc;
  // step 3: per-column checks (lib.rs:923-944), in parallel over columns like the reference's par_iter;
  // the error reported is that of the first failing column, with the reference's precedence degree > eval > path
  std::vector<uint64_t> cols_to_open(n_columns);
  for (auto& x : cols_to_open) x = rng.uniform(n_cols);
  std::vector<int> status(n_columns, 0);
  parallel_for(n_columns, 4, [&](uint64_t b, uint64_t e) {
    for (uint64_t i = b; i < e; i++) {
      const uint64_t cn = cols_to_open[i];
      bool rnd = true, evl = true;
      for (uint64_t d = 0; d <= n_deg; d++) {
        const uint64_t* tensor = d < n_deg ? rand_tensors[d].data() : outer;
        uint64_t acc[MAXL] = {0, 0, 0, 0}, t[MAXL];
        for (uint64_t k = 0; k < n_rows; k++) { h_mul(f, t, tensor + k * L, cols[i].data() + k * L); h_add(f, acc, acc, t); }
        const bool ok = h_eq(f, acc, &enc[(d * n_cols + cn) * L]);           // verify_column_value lib.rs:985-1000
        if (d < n_deg) rnd = rnd && ok; else evl = ok;
      }
      uint8_t h[32], blk[64];                                                  // verify_column_path lib.rs:955-982
      hash_column_host(f, cols[i].data(), n_rows, h);
      uint64_t cc = cn;
      for (uint64_t k = 0; k < paths[i].n; k++) {
        const uint8_t* pk = paths[i].p + 40 * k + 8;
        if (cc % 2 == 0) { memcpy(blk, h, 32); memcpy(blk + 32, pk, 32); } else { memcpy(blk, pk, 32); memcpy(blk + 32, h, 32); }
        blake3_host(blk, 64, h);
        cc >>= 1;
      }
      const bool pth = memcmp(h, root, 32) == 0;
      status[i] = !rnd ? LCPC_VERR_COLUMN_DEGREE : (!evl ? LCPC_VERR_COLUMN_EVAL : (!pth ? LCPC_VERR_COLUMN_PATH : 0));
    }
  }, n_columns * n_rows > ((uint64_t)1 << 19) ? 64u : 16u);       // Brakedown opens 6593 columns: ~0.1 s of work single-threaded
  tv[4] = now();
  for (uint64_t i = 0; i < n_columns; i++)
    if (status[i]) return status[i];
  if (dbg)
    fprintf(stderr, "[lcpc_verify] parse %.2f ms, transcript %.2f (row encodes on the GPU meanwhile: %.2f), wait for encodes %.2f, column checks %.2f\n",
            tv[1] - tv[0], tv[2] - tv[1], t_enc, tv[3] - tv[2], tv[4] - tv[3]);
  uint64_t acc[MAXL] = {0, 0, 0, 0}, t[MAXL];                                  // lib.rs:947-951
  for (uint64_t k = 0; k < n_per_row; k++) { h_mul(f, t, inner + k * L, p_eval.data() + k * L); h_add(f, acc, acc, t); }
  memcpy(eval_out, acc, F);
  return 0;
}

void lcpc_root_bincode(const uint8_t root[32], uint8_t out[40]) {
  const uint64_t l = 32;
  memcpy(out, &l, 8);
  memcpy(out + 8, root, 32);
}
void lcpc_free(void* p) { free(p); }

// ---- row-sharded commit ------------------------------------------------------------------------------------
int lcpc_shard_layout(const lcpc_ctx* c, uint64_t n_rows_total, uint64_t* rb, uint64_t* re, uint64_t* cb, uint64_t* ce, uint64_t* nch) {
  if (!c || n_rows_total == 0) return LCPC_ERR_ARG;
  uint64_t a, b, x, y, z;
  shard_layout(c, n_rows_total, &a, &b, &x, &y, &z);
  if (rb) *rb = a;
  if (re) *re = b;
  if (cb) *cb = x;
  if (ce) *ce = y;
  if (nch) *nch = z;
  return 0;
}

int lcpc_shard_nodes(uint64_t n_chunks, uint32_t G, uint32_t g, uint32_t* n_nodes, uint64_t* first, uint32_t* lg) {
  if (!n_nodes || !first || !lg || n_chunks == 0) return LCPC_ERR_ARG;
  if (G <= 1) { G = 1; g = 0; }
  if (g >= G) return LCPC_ERR_ARG;
  *n_nodes = (uint32_t)shard_nodes(n_chunks * g / G, n_chunks * (g + 1) / G, first, lg);
  return 0;
}

int lcpc_commit_shard_device(lcpc_ctx* c, const uint64_t* coeffs_local, uint64_t n_rows_total, void* stream, uint8_t* nodes_dev) {
  if (!c || n_rows_total == 0 || !nodes_dev) return LCPC_ERR_ARG;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  hipStream_t st = (hipStream_t)stream;
  uint64_t rb, re, cb, ce, nch;
  shard_layout(c, n_rows_total, &rb, &re, &cb, &ce, &nch);
  c->n_rows = n_rows_total; c->row_begin = rb; c->n_rows_local = re - rb;
  c->chunk_begin = cb; c->chunk_end = ce; c->n_chunks = nch;
  int rc = ensure_buffers(c, c->n_rows_local ? c->n_rows_local : 1);
  if (rc) return rc;
  c->launches[0] = c->launches[1] = c->launches[2] = 0;
  c->comm_t = false;
  if (c->timing) HIPCHK(c, hipEventRecord(c->ev[0], st));
  if (c->n_rows_local) {
    if (!coeffs_local) return LCPC_ERR_ARG;
    if (c->prm.encoding == LCPC_ENC_LIGERO) {
      rc = encode_rows_device(c, reinterpret_cast<const uint32_t*>(coeffs_local), c->n_per_row, c->n_per_row, c->d_comm,
                              c->n_rows_local, st, ~(uint64_t)0, c->d_coeffs, c->comm_canon);   // coeffs copy fused into pass 1
    } else {
      HIPCHK(c, hipMemcpyAsync(c->d_coeffs, coeffs_local, (size_t)c->n_rows_local * c->n_per_row * elem_bytes(c), hipMemcpyDeviceToDevice, st));
      rc = encode_rows_device(c, c->d_coeffs, c->n_per_row, c->n_per_row, c->d_comm, c->n_rows_local, st, ~(uint64_t)0, nullptr, false, true);
    }
    if (rc) return rc;
  }
  if (c->timing) HIPCHK(c, hipEventRecord(c->ev[1], st));
  if (ce > cb) {
    uint64_t first[64];
    uint32_t lg[64];
    const int n_nodes = shard_nodes(cb, ce, first, lg);
    bool all_single = true;
    for (int k = 0; k < n_nodes; k++) all_single = all_single && lg[k] == 0;
    LeafArgs la{};
    la.comm = c->d_comm; la.canon_in = c->comm_canon ? 1u : 0u; la.row_stride = c->n_cols; la.col_stride = 1; la.n_cols = c->n_cols; la.row_base = (int64_t)rb;
    if (c->comm_t) { la.comm = c->d_t; la.row_stride = 1; la.col_stride = c->n_rows_local; }
    la.n_rows_total = n_rows_total; la.chunk_begin = (uint32_t)cb; la.n_chunks_local = (uint32_t)(ce - cb);
    la.n_chunks_total = (uint32_t)nch;
    if (all_single) {                       // nothing to pre-merge: chunk CVs are the nodes
      la.out = reinterpret_cast<uint32_t*>(nodes_dev);
      HIPCHK(c, launch_leaf_chunks(c->NL, la, st));
      c->launches[1]++;
    } else {
      if ((rc = ensure_cvs(c, ce - cb))) return rc;
      la.out = c->d_cvs;
      HIPCHK(c, launch_leaf_chunks(c->NL, la, st));
      c->launches[1]++;
      for (int k = 0; k < n_nodes; k++) {   // one subtree CV per aligned block of chunks
        uint32_t* blk = c->d_cvs + (first[k] - cb) * c->n_cols * 8;
        uint32_t* out = reinterpret_cast<uint32_t*>(nodes_dev) + (size_t)k * c->n_cols * 8;
        HIPCHK(c, launch_leaf_finish_nodes(blk, nullptr, nullptr, 1u << lg[k], c->n_cols, out, false, st));
        c->launches[1]++;
      }
    }
  }
  if (c->timing) HIPCHK(c, hipEventRecord(c->ev[2], st));
  return 0;
}

int lcpc_commit_finish_device(lcpc_ctx* c, uint8_t* gathered, uint64_t n_rows_total, uint32_t slots_per_rank, void* stream, uint8_t* root) {
  if (!c || !gathered || n_rows_total != c->n_rows || slots_per_rank == 0) return LCPC_ERR_ARG;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  hipStream_t st = (hipStream_t)stream;
  const uint64_t nch = leaf_chunks(c, n_rows_total);
  const uint32_t G = c->prm.shard_count > 1 ? c->prm.shard_count : 1;
  // node table over all ranks, in chunk order: slot in the gathered buffer + log2(size)
  std::vector<uint32_t> tab_slot, tab_log;
  for (uint32_t r = 0; r < G; r++) {
    uint64_t first[64];
    uint32_t lg[64];
    const int n = shard_nodes(nch * r / G, nch * (r + 1) / G, first, lg);
    if ((uint32_t)n > slots_per_rank) return LCPC_ERR_ARG;
    for (int k = 0; k < n; k++) { tab_slot.push_back(r * slots_per_rank + (uint32_t)k); tab_log.push_back(lg[k]); }
  }
  const uint32_t n_nodes = (uint32_t)tab_slot.size();
  const uint64_t key = (nch << 24) ^ ((uint64_t)slots_per_rank << 8) ^ G;
  if (!c->d_node_tab || c->node_tab_key != key) {
    dev_free(c->d_node_tab);
    c->d_node_tab = nullptr;
    int rc = dev_alloc(c, &c->d_node_tab, (size_t)n_nodes * 8);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_node_tab, tab_slot.data(), n_nodes * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_node_tab + n_nodes, tab_log.data(), n_nodes * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));      // tab_* are stack temporaries
    c->node_tab_key = key;
  }
  if (nch == 1) {   // single-chunk message: the one "node" already carries ROOT (leaf_chunk_kernel)
    HIPCHK(c, hipMemcpyAsync(c->d_hashes, gathered + (size_t)tab_slot[0] * c->n_cols * 32, (size_t)c->n_cols * 32, hipMemcpyDeviceToDevice, st));
  } else {
    HIPCHK(c, launch_leaf_finish_nodes(reinterpret_cast<uint32_t*>(gathered), c->d_node_tab, c->d_node_tab + n_nodes, n_nodes, c->n_cols,
                                       c->d_hashes, true, st));
    c->launches[1]++;
  }
  if (c->np2 > c->n_cols)
    HIPCHK(c, hipMemsetAsync(c->d_hashes + c->n_cols * 8, 0, (size_t)(c->np2 - c->n_cols) * 32, st));
  if (c->np2 > 1) { HIPCHK(c, launch_merkle_tree(c->d_hashes, c->np2, st)); c->launches[2]++; }
  c->committed = true;
  if (c->timing) {
    HIPCHK(c, hipEventRecord(c->ev[3], st));
    HIPCHK(c, hipEventSynchronize(c->ev[3]));
    (void)hipEventElapsedTime(&c->last.encode_ms, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&c->last.hash_ms, c->ev[1], c->ev[2]);
    (void)hipEventElapsedTime(&c->last.merkle_ms, c->ev[2], c->ev[3]);   // includes the caller's exchange
    (void)hipEventElapsedTime(&c->last.total_ms, c->ev[0], c->ev[3]);
    c->last.encode_launches = c->launches[0]; c->last.hash_launches = c->launches[1]; c->last.merkle_launches = c->launches[2];
  }
  if (root) {
    HIPCHK(c, hipMemcpyAsync(root, c->d_hashes + (2 * c->np2 - 2) * 8, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
  }
  return 0;
}

int lcpc_set_timing(lcpc_ctx* c, int enable) { if (!c) return LCPC_ERR_ARG; c->timing = enable != 0; return 0; }
int lcpc_get_timings(lcpc_ctx* c, lcpc_timings* out) { if (!c || !out) return LCPC_ERR_ARG; *out = c->last; return 0; }

}  // extern "C"
