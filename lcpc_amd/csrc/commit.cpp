// lcpc_amd/csrc/commit.cpp -- lcpc_commit: an LcCommit<D, E> resident in HBM.
//
// commit (lcpc-2d/src/lib.rs:622-671), merkleize (690-704), check_comm (673-688), open_column (788-825) and
// collapse_columns (1095-1123) of /root/reference.  Many commitments may be live under one encoder context
// (lib.rs:299-311 borrow `&E`); each object owns its comm / coeffs / hashes and its scratch.
#include "internal.h"
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

using namespace lcpc;

namespace lcpc {

int ensure_scratch(lcpc_commit_t* m, uint64_t bytes) { return ensure_dev(&m->err, &m->d_scratch, &m->scratch_cap, bytes); }

int ensure_cvs(lcpc_commit_t* m, uint64_t n_chunks) {
  uint64_t cap_b = m->cap_cvs * 32;
  int rc = ensure_dev(&m->err, &m->d_cvs, &cap_b, n_chunks * m->enc->n_cols * 32);
  m->cap_cvs = rc ? 0 : cap_b / 32;
  return rc;
}

// comm (unless the commitment will live position-major in ws.d_t), coeffs (unless borrowed), hashes
int ensure_commit_buffers(lcpc_commit_t* m, uint64_t n_rows_local, bool own_coeffs) {
  const lcpc_ctx* c = m->enc;
  const size_t eb = elem_bytes(c);
  const uint64_t rows = n_rows_local ? n_rows_local : 1;
  int rc;
  if (own_coeffs && (rows > m->cap_coeff_rows || !m->d_coeffs)) {
    dev_free(m->d_coeffs); m->d_coeffs = nullptr; m->cap_coeff_rows = 0;
    if ((rc = dev_alloc(&m->err, &m->d_coeffs, (size_t)rows * c->n_per_row * eb))) return rc;
    m->cap_coeff_rows = rows;
  }
  const bool need_comm = !(c->prm.encoding == LCPC_ENC_SDIG && n_rows_local >= SDIG_T_MIN_ROWS);   // else made on demand (lcpc_get_comm)
  if (need_comm && (rows > m->cap_comm_rows || !m->d_comm)) {
    dev_free(m->d_comm); m->d_comm = nullptr; m->cap_comm_rows = 0;
    if ((rc = dev_alloc(&m->err, &m->d_comm, (size_t)rows * c->n_cols * eb))) return rc;
    m->cap_comm_rows = rows;
  }
  if (!m->d_hashes && (rc = dev_alloc(&m->err, &m->d_hashes, (size_t)(2 * c->np2 - 1) * 32))) return rc;
  return 0;
}

static LeafArgs leaf_args(const lcpc_commit_t* m) {
  const lcpc_ctx* c = m->enc;
  LeafArgs la{};
  la.comm = m->d_comm; la.canon_in = c->comm_canon ? 1u : 0u; la.row_stride = c->n_cols; la.col_stride = 1; la.n_cols = c->n_cols;
  if (m->comm_t) { la.comm = m->ws.d_t; la.canon_in = 1u; la.row_stride = 1; la.col_stride = m->n_rows_local; }
  la.n_rows_total = m->n_rows;
  return la;
}

int merkle_top(lcpc_commit_t* m, hipStream_t st, uint32_t levels_done) {
  const lcpc_ctx* c = m->enc;
  if (c->np2 > c->n_cols)   // hashes[n_cols..np2) stay zero (lib.rs:656-666)
    HIPCHK(m, hipMemsetAsync(m->d_hashes + c->n_cols * 8, 0, (size_t)(c->np2 - c->n_cols) * 32, st));
  if (c->np2 > 1) {
    if (!m->h_root) {             // once per object; without the mapping the root is copied out as before
      void* hp = nullptr;
      if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) { m->h_root = static_cast<uint32_t*>(hp); m->d_root_alias = static_cast<uint32_t*>(dp); }
        else (void)hipHostFree(hp);
      }
      if (!m->h_root) (void)hipGetLastError();   // the failed call's error must not surface at the next launch check
    }
    HIPCHK(m, launch_merkle_tree_from(m->d_hashes, c->np2, levels_done, st, m->d_root_alias));
    m->launches[2]++;
  }
  return 0;
}
// the root of a commit that was just enqueued on `st`, on the host (synchronises the stream)
int fetch_root(lcpc_commit_t* m, hipStream_t st, uint8_t* root) {
  const lcpc_ctx* c = m->enc;
  if (m->d_root_alias && c->np2 > 1) {
    HIPCHK(m, hipStreamSynchronize(st));
    memcpy(root, m->h_root, 32);
    return 0;
  }
  HIPCHK(m, hipMemcpyAsync(root, m->d_hashes + (2 * c->np2 - 2) * 8, 32, hipMemcpyDeviceToHost, st));
  HIPCHK(m, hipStreamSynchronize(st));
  return 0;
}

// hash_columns + merkle_tree on the local comm (unsharded) -- lib.rs:690-704
static int merkleize_device(lcpc_commit_t* m, hipStream_t st) {
  const lcpc_ctx* c = m->enc;
  const uint64_t n_chunks = leaf_chunks(c, m->n_rows);
  LeafArgs la = leaf_args(m);
  la.row_base = 0; la.chunk_begin = 0; la.n_chunks_local = (uint32_t)n_chunks; la.n_chunks_total = (uint32_t)n_chunks;
  if (leaf_tree_supported(la, c->np2)) {
    // small commitment: leaf digests and the first six levels of the tree in one launch (kernels.hip leaf_tree_kernel)
    HIPCHK(m, launch_leaf_tree(c->NL, la, m->d_hashes, c->np2, st));
    m->launches[1]++;
    if (m->timing) HIPCHK(m, hipEventRecord(m->ev[2], st));
    return merkle_top(m, st, 6);
  }
  if (n_chunks == 1) {
    la.out = m->d_hashes;
    HIPCHK(m, launch_leaf_chunks(c->NL, la, st));
    m->launches[1]++;
  } else {
    int rc = ensure_cvs(m, n_chunks);
    if (rc) return rc;
    la.out = m->d_cvs;
    HIPCHK(m, launch_leaf_chunks(c->NL, la, st));
    HIPCHK(m, launch_leaf_finish(m->d_cvs, (uint32_t)n_chunks, c->n_cols, m->d_hashes, st));
    m->launches[1] += 2;
  }
  if (m->timing) HIPCHK(m, hipEventRecord(m->ev[2], st));
  return merkle_top(m, st);
}

int finish_timing(lcpc_commit_t* m, hipStream_t st) {
  if (!m->timing) return 0;
  HIPCHK(m, hipEventRecord(m->ev[3], st));
  HIPCHK(m, hipEventSynchronize(m->ev[3]));
  (void)hipEventElapsedTime(&m->last.encode_ms, m->ev[0], m->ev[1]);
  (void)hipEventElapsedTime(&m->last.hash_ms, m->ev[1], m->ev[2]);
  (void)hipEventElapsedTime(&m->last.merkle_ms, m->ev[2], m->ev[3]);
  (void)hipEventElapsedTime(&m->last.total_ms, m->ev[0], m->ev[3]);
  m->last.encode_launches = m->launches[0];
  m->last.hash_launches = m->launches[1];
  m->last.merkle_launches = m->launches[2];
  return 0;
}

// a new commit starts: whatever the object held is gone, and stays gone if anything below fails.  `st` -- the stream that
// will rewrite the object's buffers -- is first ordered behind the commit that filled them last (which may still be running
// on another, non-blocking stream of the caller: lcpc_commit_device only enqueues)
static int begin_commit(lcpc_commit_t* m, hipStream_t st, uint64_t n_rows_total, uint64_t row_begin, uint64_t n_rows_local) {
  int rc = order_after_commit(m, st);
  if (rc) return rc;
  m->committed = false;
  m->comm_t = false;
  m->comm_rows_valid = false;
  m->coeffs_view = nullptr;
  m->n_rows = n_rows_total; m->row_begin = row_begin; m->n_rows_local = n_rows_local;
  m->launches[0] = m->launches[1] = m->launches[2] = 0;
  m->last.staged_slices = 0;
  return 0;
}

// encode the local rows of `src` (row-major, n_per_row per row; flat elements >= n_src_total read as zero) into the
// commitment matrix; copy_coeffs: write the padded LcCommit.coeffs copy while the source streams through
static int encode_commit(lcpc_commit_t* m, const uint32_t* src, uint64_t n_src_total, bool copy_coeffs, hipStream_t st) {
  const lcpc_ctx* c = m->enc;
  EncodeJob j;
  j.src = src; j.src_stride = c->n_per_row; j.n_valid = c->n_per_row; j.dst = m->d_comm; j.n_rows = m->n_rows_local;
  j.n_src_total = n_src_total;
  j.copy_dst = copy_coeffs ? m->d_coeffs : nullptr;
  j.canon_out = c->prm.encoding == LCPC_ENC_SDIG ? true : c->comm_canon;
  j.keep_t = true;
  bool kept = false;
  j.kept_t = &kept;
  int rc = encode_rows_device(c, &m->ws, j, st, &m->err, &m->launches[0]);
  if (rc) return rc;
  m->comm_t = kept;
  return 0;
}

// can the first encode pass write the coeffs copy on the fly?  (Ligero: fused into the first NTT pass; Brakedown with
// >= SDIG_T_MIN_ROWS rows: fused into the input transpose)
static bool fused_copy(const lcpc_ctx* c, uint64_t n_rows_local) { return c->prm.encoding == LCPC_ENC_LIGERO || n_rows_local >= SDIG_T_MIN_ROWS; }

static int commit_tail(lcpc_commit_t* m, hipStream_t st, uint8_t* root) {
  if (m->timing) HIPCHK(m, hipEventRecord(m->ev[1], st));
  int rc;
  if ((rc = merkleize_device(m, st))) return rc;
  if ((rc = finish_timing(m, st))) return rc;
  m->committed = true;
  // whatever reads the commitment next (prove, collapse, open, the getters) runs on another stream -- the null stream, or the
  // sharded prover's own -- and a caller's NON-BLOCKING stream is not implicitly ordered before those: they wait for this event
  if (!m->ev_done) HIPCHK(m, hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming));
  HIPCHK(m, hipEventRecord(m->ev_done, st));
  if (root) return fetch_root(m, st, root);
  return 0;
}
// order `st` behind the commit that filled this object (no-op when the commit ran on `st` itself or has long finished)
int order_after_commit(lcpc_commit_t* m, hipStream_t st) {
  if (m->ev_done) HIPCHK(m, hipStreamWaitEvent(st, m->ev_done, 0));
  return 0;
}

static uint32_t collapse_splits(const lcpc_commit_t* m) {
  const uint64_t col_blocks = (m->enc->n_per_row + 255) / 256;
  uint32_t n_splits = 1;   // split the row range so that the grid has >= ~2k workgroups even for narrow matrices
  while (n_splits < 64 && col_blocks * n_splits < 2048 && m->n_rows_local / (n_splits * 2) >= 16) n_splits *= 2;
  return n_splits;
}
size_t collapse_scratch_bytes(const lcpc_commit_t* m, uint32_t n_tensors) {
  return (size_t)collapse_splits(m) * n_tensors * m->enc->n_per_row * elem_bytes(m->enc);
}
// one launch (+ the sum over row splits) for the column range [j0, j1) of n_tensors polynomials; d_out = the polynomials' array
// ([n_tensors][n_per_row] elements); `a` arrives with coeffs / tensors / tensors29 / n_rows / n_per_row / n_tensors filled in.  A range
// narrower than the whole polynomial (prove's first slice) takes only one tensor and splits its rows further, so that it still fills
// the chip: the partials never exceed the reserved collapse_scratch_bytes(m, 2)
static int collapse_range(lcpc_commit_t* m, CollapseArgs a, uint64_t j0, uint64_t j1, hipStream_t st, uint32_t* d_out) {
  const lcpc_ctx* c = m->enc;
  const uint64_t len = j1 - j0;
  uint32_t n_splits = collapse_splits(m);
  const bool whole = len == c->n_per_row;
  if (!whole) {
    if (a.n_tensors != 1) return LCPC_ERR_ARG;
    while (n_splits < 64 && ((len + 255) / 256) * n_splits < 2048 && m->n_rows_local / (n_splits * 2) >= 8) n_splits *= 2;
    while (n_splits > 1 && (uint64_t)n_splits * len > (uint64_t)collapse_splits(m) * 2 * c->n_per_row) n_splits /= 2;
  }
  a.n_splits = n_splits; a.j0 = j0; a.j1 = j1;
  if (n_splits == 1) {
    a.out = d_out + j0 * c->NL; a.out_stride = c->n_per_row;
    HIPCHK(m, launch_collapse(c->NL, a, st));
    return 0;
  }
  const size_t part_bytes = (size_t)n_splits * a.n_tensors * len * elem_bytes(c);
  // partials live at the end of scratch (callers reserve it; scratch_cap is a multiple of 256, part_bytes of 16)
  uint32_t* d_part = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(m->d_scratch) + m->scratch_cap - ((part_bytes + 255) & ~(size_t)255));
  a.out = d_part; a.out_stride = len;
  HIPCHK(m, launch_collapse(c->NL, a, st));
  // (whole: flat over [n_tensors][n_per_row]; a range: one tensor, its outputs start at element j0)
  HIPCHK(m, launch_field_sum(c->NL, d_part, n_splits, (uint64_t)a.n_tensors * len, d_out + j0 * c->NL, st));
  return 0;
}
static int collapse_prepare(lcpc_commit_t* m, const uint32_t* d_tensors, uint32_t n_tensors, hipStream_t st, CollapseArgs* a) {
  const lcpc_ctx* c = m->enc;
  *a = CollapseArgs{};
  a->coeffs = m->coeffs_view; a->tensors = d_tensors; a->n_rows = m->n_rows_local; a->n_per_row = c->n_per_row;
  a->n_tensors = n_tensors;
  if (c->NL == 8) {
    const uint64_t ne = (uint64_t)n_tensors * m->n_rows_local;
    int rc = ensure_dev(&m->err, &m->d_t29, &m->t29_cap, ne * 48);
    if (rc) return rc;
    HIPCHK(m, launch_to_r29(d_tensors, ne, m->d_t29, st));
    a->tensors29 = m->d_t29;
  }
  return 0;
}
static int collapse_local(lcpc_commit_t* m, const uint32_t* d_tensors, uint32_t n_tensors, hipStream_t st, uint32_t* d_out) {
  CollapseArgs a;
  int rc = collapse_prepare(m, d_tensors, n_tensors, st, &a);
  if (rc) return rc;
  return collapse_range(m, a, 0, m->enc->n_per_row, st, d_out);
}
// scratch layout for collapse: [tensors (host entry only)] [polys (host entry only)] ... [partials at the end]
int collapse_run(lcpc_commit_t* m, const uint32_t* d_tensors, uint32_t n_tensors, hipStream_t st, uint32_t* d_polys) {
  const lcpc_ctx* c = m->enc;
  for (uint32_t t = 0; t < n_tensors; t += 2) {
    const uint32_t nt = (n_tensors - t) >= 2 ? 2 : 1;
    int rc = collapse_local(m, d_tensors + (size_t)t * m->n_rows_local * c->NL, nt, st, d_polys + (size_t)t * c->n_per_row * c->NL);
    if (rc) return rc;
  }
  return 0;
}

// collapse_columns for tensors in host memory; polys_canon (optional): the same polynomials as canonical values
// (PrimeField::to_repr limbs, what the transcript absorbs, lib.rs:47-57), converted on the device
int collapse_host(lcpc_commit_t* m, const uint64_t* tensors, uint32_t n_tensors, uint64_t* polys, uint64_t* polys_canon) {
  if (!m || !tensors || !polys || n_tensors == 0) return LCPC_ERR_ARG;
  if (!m->committed) return LCPC_ERR_STATE;
  const lcpc_ctx* c = m->enc;
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  const size_t eb = elem_bytes(c);
  const size_t tb = ((size_t)n_tensors * m->n_rows_local * eb + 255) & ~(size_t)255;
  const size_t pbytes = (size_t)n_tensors * c->n_per_row * eb;
  const size_t pb = (pbytes + 255) & ~(size_t)255;
  int rc = ensure_scratch(m, tb + 2 * pb + collapse_scratch_bytes(m, 2) + 512);
  if (rc) return rc;
  uint8_t* base = reinterpret_cast<uint8_t*>(m->d_scratch);
  uint32_t* d_t = reinterpret_cast<uint32_t*>(base);
  uint32_t* d_p = reinterpret_cast<uint32_t*>(base + tb);
  uint32_t* d_pc = reinterpret_cast<uint32_t*>(base + tb + pb);
  if ((rc = order_after_commit(m, nullptr))) return rc;
  HIPCHK(m, hipMemcpyAsync(d_t, tensors, (size_t)n_tensors * m->n_rows_local * eb, hipMemcpyHostToDevice, nullptr));
  if ((rc = collapse_run(m, d_t, n_tensors, nullptr, d_p))) return rc;
  if (polys_canon) HIPCHK(m, launch_to_canon(c->NL, d_p, (uint64_t)n_tensors * c->n_per_row, d_pc, nullptr));
  HIPCHK(m, hipMemcpyAsync(polys, d_p, pbytes, hipMemcpyDeviceToHost, nullptr));
  if (polys_canon) HIPCHK(m, hipMemcpyAsync(polys_canon, d_pc, pbytes, hipMemcpyDeviceToHost, nullptr));
  HIPCHK(m, hipStreamSynchronize(nullptr));
  return 0;
}

// collapse_columns of ONE tensor for the prover, enqueued without waiting and delivered in two column ranges -- [0, cut) and
// [cut, n_per_row), cut = *cut_out -- each followed by its device-to-host copies (Montgomery form and canonical values) and an event:
// the transcript absorbs the first range (lib.rs:1045-1047 is serial, ~50 ns per coefficient) while the second is still being
// computed, so that of the collapse only the first eighth stays on the prover's critical path.  The caller waits for ev[0] / ev[1]
// (collapse_wait_slice); everything is on the null stream, so later work of this commitment queues up behind it.
int collapse_host_sliced(lcpc_commit_t* m, const uint64_t* tensor, uint64_t* polys, uint64_t* polys_canon, uint64_t* cut_out) {
  if (!m || !tensor || !polys || !polys_canon || !cut_out) return LCPC_ERR_ARG;
  if (!m->committed) return LCPC_ERR_STATE;
  const lcpc_ctx* c = m->enc;
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  const size_t eb = elem_bytes(c);
  const uint64_t np = c->n_per_row;
  const uint64_t cut = ((np / 8) + 255) & ~(uint64_t)255;
  if (cut == 0 || cut >= np) return LCPC_ERR_ARG;
  const size_t tb = ((size_t)m->n_rows_local * eb + 255) & ~(size_t)255;
  const size_t pb = ((size_t)np * eb + 255) & ~(size_t)255;
  int rc = ensure_scratch(m, tb + 2 * pb + collapse_scratch_bytes(m, 2) + 512);
  if (rc) return rc;
  for (auto& e : m->ev_slice)
    if (!e) HIPCHK(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  uint8_t* base = reinterpret_cast<uint8_t*>(m->d_scratch);
  uint32_t* d_t = reinterpret_cast<uint32_t*>(base);
  uint32_t* d_p = reinterpret_cast<uint32_t*>(base + tb);
  uint32_t* d_pc = reinterpret_cast<uint32_t*>(base + tb + pb);
  if ((rc = order_after_commit(m, nullptr))) return rc;
  HIPCHK(m, hipMemcpyAsync(d_t, tensor, (size_t)m->n_rows_local * eb, hipMemcpyHostToDevice, nullptr));
  CollapseArgs a;
  if ((rc = collapse_prepare(m, d_t, 1, nullptr, &a))) return rc;
  const uint64_t lim[3] = {0, cut, np};
  for (int s = 0; s < 2; s++) {
    const uint64_t j0 = lim[s], len = lim[s + 1] - lim[s];
    if ((rc = collapse_range(m, a, j0, j0 + len, nullptr, d_p))) return rc;
    HIPCHK(m, launch_to_canon(c->NL, d_p + j0 * c->NL, len, d_pc + j0 * c->NL, nullptr));
    HIPCHK(m, hipMemcpyAsync(reinterpret_cast<uint8_t*>(polys_canon) + j0 * eb, reinterpret_cast<uint8_t*>(d_pc) + j0 * eb, len * eb, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(m, hipMemcpyAsync(reinterpret_cast<uint8_t*>(polys) + j0 * eb, reinterpret_cast<uint8_t*>(d_p) + j0 * eb, len * eb, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(m, hipEventRecord(m->ev_slice[s], nullptr));
  }
  *cut_out = cut;
  return 0;
}
int collapse_wait_slice(lcpc_commit_t* m, int s) {
  HIPCHK(m, hipEventSynchronize(m->ev_slice[s]));
  return 0;
}

int open_columns_device(lcpc_commit_t* m, const uint64_t* d_cols, uint32_t n, uint32_t* d_vals, uint32_t* d_paths, hipStream_t st) {
  const lcpc_ctx* c = m->enc;
  if (d_vals && m->n_rows_local) {
    if (m->comm_t)
      HIPCHK(m, launch_gather_columns(c->NL, m->ws.d_t, m->n_rows_local, 1, m->n_rows_local, d_cols, n, d_vals, c->d_r2, st));
    else
      HIPCHK(m, launch_gather_columns(c->NL, m->d_comm, m->n_rows_local, c->n_cols, 1, d_cols, n, d_vals, c->comm_canon ? c->d_r2 : nullptr, st));
  }
  if (d_paths && c->path_len) HIPCHK(m, launch_gather_paths(m->d_hashes, c->np2, c->path_len, d_cols, n, d_paths, st));
  return 0;
}

// open_column (lib.rs:788-825) for n columns into host memory.  vals_pitch: bytes between the values of consecutive
// columns (0 = packed, n_rows * F): prove lets the device-to-host copy drop them straight into the bincode layout
int open_columns_host(lcpc_commit_t* m, const uint64_t* cols, uint32_t n, uint64_t* col_vals, size_t vals_pitch, uint8_t* paths) {
  if (!m || !cols || (!col_vals && !paths)) return LCPC_ERR_ARG;
  if (!m->committed) return LCPC_ERR_STATE;
  const lcpc_ctx* c = m->enc;
  for (uint32_t i = 0; i < n; i++)
    if (cols[i] >= c->n_cols) return LCPC_ERR_COLUMN_NUMBER;        // lib.rs:797-799
  if (n == 0) return 0;
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  const size_t eb = elem_bytes(c);
  const size_t col_b = (size_t)m->n_rows_local * eb;
  const size_t vb = (((size_t)n * col_b) + 255) & ~(size_t)255, pb = (((size_t)n * c->path_len * 32) + 255) & ~(size_t)255;
  const size_t cb = (((size_t)n * 8) + 255) & ~(size_t)255;
  int rc = ensure_scratch(m, vb + pb + cb);
  if (rc) return rc;
  uint8_t* base = reinterpret_cast<uint8_t*>(m->d_scratch);
  uint64_t* d_cols = reinterpret_cast<uint64_t*>(base);
  uint32_t* d_vals = reinterpret_cast<uint32_t*>(base + cb);
  uint32_t* d_paths = reinterpret_cast<uint32_t*>(base + cb + vb);
  if ((rc = order_after_commit(m, nullptr))) return rc;
  HIPCHK(m, hipMemcpyAsync(d_cols, cols, (size_t)n * 8, hipMemcpyHostToDevice, nullptr));
  if ((rc = open_columns_device(m, d_cols, n, col_vals ? d_vals : nullptr, paths ? d_paths : nullptr, nullptr))) return rc;
  if (col_vals && m->n_rows_local) {
    if (vals_pitch == 0 || vals_pitch == col_b)
      HIPCHK(m, hipMemcpyAsync(col_vals, d_vals, (size_t)n * col_b, hipMemcpyDeviceToHost, nullptr));
    else
      HIPCHK(m, hipMemcpy2DAsync(col_vals, vals_pitch, d_vals, col_b, col_b, n, hipMemcpyDeviceToHost, nullptr));
  }
  if (paths && c->path_len) HIPCHK(m, hipMemcpyAsync(paths, d_paths, (size_t)n * c->path_len * 32, hipMemcpyDeviceToHost, nullptr));
  HIPCHK(m, hipStreamSynchronize(nullptr));
  return 0;
}

// pinned host memory that lives with the commitment (prove's tensors / polynomials: no page faults, full-speed D2H)
int ensure_pinned(lcpc_commit_t* m, uint64_t bytes) {
  if (bytes > m->h_pin_cap || !m->h_pin) {
    if (m->h_pin) (void)hipHostFree(m->h_pin);
    m->h_pin = nullptr; m->h_pin_cap = 0;
    HIPCHK(m, hipHostMalloc(reinterpret_cast<void**>(&m->h_pin), (size_t)bytes, hipHostMallocDefault));
    m->h_pin_cap = bytes;
  }
  return 0;
}

}  // namespace lcpc

// =====================================================================================================
// C ABI: LcCommit
// =====================================================================================================
extern "C" {

int lcpc_commit_create(lcpc_ctx* enc, lcpc_commit_t** out) {
  if (!enc || !out) return LCPC_ERR_ARG;
  *out = nullptr;
  lcpc_commit_t* m = new (std::nothrow) lcpc_commit_t();
  if (!m) return LCPC_ERR_NOMEM;
  m->enc = enc;
  ctx_ref(enc);
  if (hipSetDevice(enc->prm.device) != hipSuccess) { lcpc_commit_destroy(m); return LCPC_ERR_NO_DEVICE; }
  for (auto& e : m->ev)
    if (hipEventCreate(&e) != hipSuccess) { lcpc_commit_destroy(m); return LCPC_ERR_HIP; }
  *out = m;
  return 0;
}

void lcpc_commit_destroy(lcpc_commit_t* m) {
  if (!m) return;
  (void)hipSetDevice(m->enc->prm.device);
  dev_free(m->d_coeffs); dev_free(m->d_comm); dev_free(m->d_hashes); dev_free(m->d_cvs); dev_free(m->d_scratch);
  for (auto& t : m->node_tabs) dev_free(t.d);
  dev_free(m->d_t29); dev_free(m->ws.d_tmp); dev_free(m->ws.d_t); dev_free(m->ws.d_mid); dev_free(m->d_gather); dev_free(m->d_xsend); dev_free(m->d_xrecv);
  for (auto& e : m->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : m->ev_batch) if (e) (void)hipEventDestroy(e);
  for (auto& e : m->ev_slice) if (e) (void)hipEventDestroy(e);
  if (m->h_pin) (void)hipHostFree(m->h_pin);
  if (m->h_root) (void)hipHostFree(m->h_root);
  if (m->s_prove) (void)hipStreamDestroy(m->s_prove);
  if (m->s_xchg) (void)hipStreamDestroy(m->s_xchg);
  if (m->ev_hashed) (void)hipEventDestroy(m->ev_hashed);
  if (m->ev_done) (void)hipEventDestroy(m->ev_done);
  if (m->s_copy) (void)hipStreamDestroy(m->s_copy);
  if (m->s_comp) (void)hipStreamDestroy(m->s_comp);
  ctx_unref(m->enc);
  delete m;
}
const char* lcpc_commit_last_error(const lcpc_commit_t* m) { return m ? m->err.c_str() : ""; }

int lcpc_commit_device(lcpc_commit_t* m, const uint64_t* coeffs_dev, uint64_t n_coeffs, void* stream, uint32_t flags, uint8_t* root) {
  if (!m || !coeffs_dev || n_coeffs == 0) return LCPC_ERR_ARG;
  const lcpc_ctx* c = m->enc;
  if (c->prm.shard_count > 1) return LCPC_ERR_STATE;
  LCPC_TRY
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  hipStream_t st = (hipStream_t)stream;
  const uint64_t n_rows = (n_coeffs + c->n_per_row - 1) / c->n_per_row;    // get_dims (ligero lib.rs:166-169)
  int rc = begin_commit(m, st, n_rows, 0, n_rows);
  if (rc) return rc;
  const uint64_t padded = n_rows * c->n_per_row;
  const bool borrow = (flags & LCPC_COMMIT_BORROW_COEFFS) && padded == n_coeffs;
  if ((rc = ensure_commit_buffers(m, n_rows, !borrow))) return rc;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(coeffs_dev);
  const size_t eb = elem_bytes(c);
  if (m->timing) HIPCHK(m, hipEventRecord(m->ev[0], st));
  if (borrow) {
    // LcCommit.coeffs IS the caller's buffer: nothing but comm and the digests is written
    if ((rc = encode_commit(m, src, n_coeffs, false, st))) return rc;
    m->coeffs_view = src;
  } else if (fused_copy(c, n_rows)) {
    // the padded local copy of coeffs (lib.rs:636-645; LcCommit keeps it for prove) is written by the first
    // NTT pass (Ligero) / the input transpose (Brakedown, position-major path) while it streams the caller's buffer
    if ((rc = encode_commit(m, src, n_coeffs, true, st))) return rc;
    m->coeffs_view = m->d_coeffs;
  } else {
    HIPCHK(m, hipMemcpyAsync(m->d_coeffs, coeffs_dev, (size_t)n_coeffs * eb, hipMemcpyDeviceToDevice, st));
    if (padded > n_coeffs)
      HIPCHK(m, hipMemsetAsync(reinterpret_cast<uint8_t*>(m->d_coeffs) + (size_t)n_coeffs * eb, 0, (size_t)(padded - n_coeffs) * eb, st));
    if ((rc = encode_commit(m, m->d_coeffs, ~(uint64_t)0, false, st))) return rc;
    m->coeffs_view = m->d_coeffs;
  }
  return commit_tail(m, st, root);
  LCPC_CATCH(m)
}

// ---- host memory -> HBM ---------------------------------------------------------------------------------------------
// LcCommit::commit(&coeffs, &enc) (lcpc-2d/src/lib.rs:299-301, 636-645) takes a slice of ordinary -- pageable -- memory.
// What hipMemcpyAsync does with such a pointer is the runtime's business: ROCm 7 on the MI355X boxes locks the caller's pages in
// place and stays asynchronous (measured, profiles/r05_host_path.jsonl: 40.6 ms for the 2 GiB headline on pages it has locked
// before, 44.4 ms on a buffer it has never seen, 40.7 from pinned memory); older stacks stage through one pinned buffer on the
// calling thread and lose the row-batch overlap below.  upload_host does not depend on either: slices of the source are copied
// into the encoder's own ring of pinned bounce buffers by the host pool (streaming stores: the DMA engine is the only reader)
// while the previous slices cross the bus, and every H2D is a true async copy from memory the library owns -- 41.2 ms whatever
// the pages' history.  The caller's memory is never registered or locked by this library.
// true: the runtime knows this memory and copies from it without staging -- hipHostMalloc'ed or registered by its owner, managed,
// or (a caller's mistake the runtime still handles, which the host pool's memcpy would not) device memory
static bool host_ptr_is_pinned(const void* p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // unknown to the runtime = pageable
  return a.type == hipMemoryTypeHost || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeDevice;
}

static bool host_ptr_is_device(const void* p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice;
}

static void stream_copy(uint8_t* dst, const uint8_t* src, size_t n) {
#if defined(__x86_64__)
  // 64 B per iteration, non-temporal stores (dst is 64-byte aligned: slices start on 4 KiB multiples of a pinned allocation)
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
      __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i)), b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 16));
      __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 32)), d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 48));
      _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i), a); _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 16), b);
      _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 32), c); _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 48), d);
    }
    _mm_sfence();
    if (i < n) memcpy(dst + i, src + i, n - i);
    return;
  }
#endif
  memcpy(dst, src, n);
}

// bytes of src (host) -> dst (device) on stream st.  Pinned sources: one async copy.  Pageable sources: through the encoder's bounce ring.
// `total`: bytes of the whole upload this call is a part of (sizes the ring once).  The ring is shared by every commitment of the
// encoder: c->stage_mu is held per call (one row batch), so two commitments uploading at once interleave batch by batch; a buffer
// is reused only after the copy that last read it has completed (ev_stage), whichever stream that copy was on.
static int upload_host(lcpc_commit_t* m, void* dst, const void* src, size_t bytes, size_t total, hipStream_t st, bool pinned) {
  lcpc_ctx* c = m->enc;
  if (bytes == 0) return 0;
  if (pinned) { HIPCHK(m, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); return 0; }
  std::lock_guard<std::mutex> sg(c->stage_mu);
  // Slice = what one H2D command moves.  Measured at 2^26 Ft255 (2 GiB; tools/bench_host_path.py, profiles/r05_host_path.jsonl): 8 MiB
  // slices 44.4 ms, 16: 42.3, 32: 41.6-42.0, 64: 41.2 (pinned source 40.7; the runtime's own pageable path 40.6 on pages it has
  // locked before, 44.4 on fresh ones) -- every copy command costs ~15 us of bus idle time, a slice's latency is paid once.
  constexpr size_t SLICE = (size_t)64 << 20;
  const size_t want = std::min(SLICE, std::max<size_t>((size_t)4 << 20, (total / 32 + 4095) & ~(size_t)4095));   // small uploads: small buffers
  if (c->stage_cap < want) {                               // a ring of smaller buffers goes; the new ones are made as they are first used
    for (unsigned k = 0; k < lcpc_ctx::N_STAGE; k++) {
      if (c->ev_stage[k]) HIPCHK(m, hipEventSynchronize(c->ev_stage[k]));
      if (c->h_stage[k]) { (void)hipHostFree(c->h_stage[k]); c->h_stage[k] = nullptr; }
    }
    c->stage_cap = want;
  }
  const size_t slice = c->stage_cap;
  const uint8_t* s = static_cast<const uint8_t*>(src);
  uint8_t* d = static_cast<uint8_t*>(dst);
  for (size_t off = 0; off < bytes; off += slice) {
    const size_t len = std::min(slice, bytes - off);
    const unsigned k = c->stage_next++ % lcpc_ctx::N_STAGE;
    if (!c->h_stage[k]) {
      // first use: pinning 64 MiB costs ~6 ms -- paid here, buffer by buffer, while the slices already enqueued cross the bus
      // (all four up front put 24 ms in front of an encoder's first pageable commit)
      if (hipHostMalloc(reinterpret_cast<void**>(&c->h_stage[k]), slice, hipHostMallocDefault) != hipSuccess) {
        // no pinned memory to be had (ulimit -l, a full host): this slice takes the runtime's own pageable path -- slower, not an error
        (void)hipGetLastError();
        c->h_stage[k] = nullptr;
        HIPCHK(m, hipMemcpyAsync(d + off, s + off, len, hipMemcpyHostToDevice, st));
        continue;
      }
      if (!c->ev_stage[k]) HIPCHK(m, hipEventCreateWithFlags(&c->ev_stage[k], hipEventDisableTiming));
    }
    HIPCHK(m, hipEventSynchronize(c->ev_stage[k]));                 // the copy that last read this buffer has left it (a fresh event is complete)
    uint8_t* hb = c->h_stage[k];
    const uint8_t* sp = s + off;
    parallel_for(len, (size_t)256 << 10, [&](uint64_t b, uint64_t e) { stream_copy(hb + b, sp + b, (size_t)(e - b)); });
    HIPCHK(m, hipMemcpyAsync(d + off, hb, len, hipMemcpyHostToDevice, st));
    HIPCHK(m, hipEventRecord(c->ev_stage[k], st));
    m->last.staged_slices++;
  }
  return 0;
}

int lcpc_commit(lcpc_commit_t* m, const uint64_t* coeffs, uint64_t n_coeffs, uint8_t* root) {
  if (!m || !coeffs || n_coeffs == 0) return LCPC_ERR_ARG;
  lcpc_ctx* c = m->enc;
  if (c->prm.shard_count > 1) return LCPC_ERR_STATE;
  LCPC_TRY
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  const uint64_t n_rows = (n_coeffs + c->n_per_row - 1) / c->n_per_row;
  int rc = begin_commit(m, nullptr, n_rows, 0, n_rows);
  if (rc) return rc;
  if ((rc = ensure_commit_buffers(m, n_rows, true))) return rc;
  const size_t eb = elem_bytes(c);
  const uint64_t padded = n_rows * c->n_per_row;
  const size_t total_bytes = (size_t)n_coeffs * eb;
  // pageable source (a Rust Vec, malloc, numpy): staged through pinned bounce buffers by the host pool; small ones are not worth the ring
  const bool pinned = c->sw_host_stage == 0 || (c->sw_host_stage < 0 && (total_bytes < ((size_t)4 << 20) || host_ptr_is_pinned(coeffs))) ||
                      (c->sw_host_stage > 0 && host_ptr_is_device(coeffs));
  // Small inputs, Brakedown (whole-matrix transposes) and timing runs: one copy, then the resident path.
  if (c->prm.encoding != LCPC_ENC_LIGERO || total_bytes < ((size_t)64 << 20) || n_rows < 16 || m->timing) {
    if ((rc = upload_host(m, m->d_coeffs, coeffs, total_bytes, total_bytes, nullptr, pinned))) return rc;
    if (padded > n_coeffs)
      HIPCHK(m, hipMemsetAsync(reinterpret_cast<uint8_t*>(m->d_coeffs) + total_bytes, 0, (size_t)(padded - n_coeffs) * eb, nullptr));
    if (m->timing) HIPCHK(m, hipEventRecord(m->ev[0], nullptr));
    if ((rc = encode_commit(m, m->d_coeffs, ~(uint64_t)0, false, nullptr))) return rc;
    m->coeffs_view = m->d_coeffs;
    if ((rc = commit_tail(m, nullptr, root))) return rc;
    HIPCHK(m, hipStreamSynchronize(nullptr));     // the caller's (possibly pageable) buffer is no longer in use on return
    return 0;
  }
  // Large Ligero commit from host memory: rows are independent (lcpc-2d lib.rs:648-653), so the matrix is
  // uploaded in row batches on a copy stream while the previous batch runs its NTT passes on a compute stream;
  // column hashing starts once the last batch is encoded.  The end-to-end time tends to the PCIe time.
  if (!m->s_copy) HIPCHK(m, hipStreamCreateWithFlags(&m->s_copy, hipStreamNonBlocking));
  if (!m->s_comp) HIPCHK(m, hipStreamCreateWithFlags(&m->s_comp, hipStreamNonBlocking));
  constexpr int NB = 16;
  for (auto& e : m->ev_batch)
    if (!e) HIPCHK(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  // Earlier work on this object's buffers: the last fill is ordered by its event (begin_commit put the null stream behind it;
  // the two streams of this path wait for it as well), and every host-pointer reader (prove, collapse, open, the getters) has
  // synchronised before it returned.  No device-wide synchronisation: other contexts and streams of the process keep running.
  if ((rc = order_after_commit(m, m->s_copy)) || (rc = order_after_commit(m, m->s_comp))) return rc;
  if (padded > n_coeffs)
    HIPCHK(m, hipMemsetAsync(reinterpret_cast<uint8_t*>(m->d_coeffs) + total_bytes, 0, (size_t)(padded - n_coeffs) * eb, m->s_copy));
  const uint64_t rows_per = (n_rows + NB - 1) / NB;
  const uint64_t n_chunks = leaf_chunks(c, n_rows);
  uint64_t chunks_hashed = 0;
  m->comm_t = false;                                        // (Ligero: row-major comm; leaf_args reads it)
  if (n_chunks > 1 && (rc = ensure_cvs(m, n_chunks))) return rc;
  for (int b = 0; b < NB; b++) {
    const uint64_t r0 = (uint64_t)b * rows_per;
    if (r0 >= n_rows) break;
    const uint64_t r1 = r0 + rows_per < n_rows ? r0 + rows_per : n_rows;
    const uint64_t e0 = r0 * c->n_per_row, e1 = r1 * c->n_per_row < n_coeffs ? r1 * c->n_per_row : n_coeffs;
    if (e1 > e0 && (rc = upload_host(m, reinterpret_cast<uint8_t*>(m->d_coeffs) + (size_t)e0 * eb, reinterpret_cast<const uint8_t*>(coeffs) + (size_t)e0 * eb,
                                     (size_t)(e1 - e0) * eb, total_bytes, m->s_copy, pinned))) return rc;
    HIPCHK(m, hipEventRecord(m->ev_batch[b], m->s_copy));
    HIPCHK(m, hipStreamWaitEvent(m->s_comp, m->ev_batch[b], 0));
    EncodeJob j;
    j.src = m->d_coeffs + (size_t)r0 * c->n_per_row * c->NL; j.src_stride = c->n_per_row; j.n_valid = c->n_per_row;
    j.dst = m->d_comm + (size_t)r0 * c->n_cols * c->NL; j.n_rows = r1 - r0; j.canon_out = c->comm_canon;
    if ((rc = encode_rows_device(c, &m->ws, j, m->s_comp, &m->err, &m->launches[0]))) return rc;
    // the column hash chunk by chunk behind the rows it needs (a 1 KiB chunk of the leaf message 0^32 || repr(col[0]) || ... spans
    // 1024 / F rows, lib.rs:719-735): the chunks the encoded rows complete are hashed now, under the upload of the next batches, and
    // only the last batch's chunks, the fold and the tree remain behind the bus -- the reference hashes after its encode loop
    // (lib.rs:648-671), same digests
    if (n_chunks > 1) {
      const uint64_t done = r1 == n_rows ? n_chunks : std::min<uint64_t>(n_chunks, (32 + (uint64_t)eb * r1) / 1024);
      if (done > chunks_hashed) {
        LeafArgs la = leaf_args(m);
        la.row_base = 0; la.n_chunks_total = (uint32_t)n_chunks;
        la.chunk_begin = (uint32_t)chunks_hashed; la.n_chunks_local = (uint32_t)(done - chunks_hashed);
        la.out = m->d_cvs + chunks_hashed * c->n_cols * 8;
        HIPCHK(m, launch_leaf_chunks(c->NL, la, m->s_comp));
        m->launches[1]++;
        chunks_hashed = done;
      }
    }
  }
  m->coeffs_view = m->d_coeffs;
  if (n_chunks > 1) {
    HIPCHK(m, launch_leaf_finish(m->d_cvs, (uint32_t)n_chunks, c->n_cols, m->d_hashes, m->s_comp));
    m->launches[1]++;
    if ((rc = merkle_top(m, m->s_comp))) return rc;
  } else if ((rc = merkleize_device(m, m->s_comp))) return rc;
  if (root) HIPCHK(m, hipMemcpyAsync(root, m->d_hashes + (2 * c->np2 - 2) * 8, 32, hipMemcpyDeviceToHost, m->s_comp));
  HIPCHK(m, hipStreamSynchronize(m->s_comp));              // later calls use the null stream / caller streams
  HIPCHK(m, hipStreamSynchronize(m->s_copy));              // (the tail memset when no batch followed it)
  m->committed = true;
  return 0;
  LCPC_CATCH(m)
}

int lcpc_commit_from_parts(lcpc_commit_t* m, const uint64_t* comm, const uint64_t* coeffs, uint64_t n_rows, uint8_t* root) {
  if (!m || !comm || n_rows == 0) return LCPC_ERR_ARG;
  const lcpc_ctx* c = m->enc;
  LCPC_TRY
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  int rc = begin_commit(m, nullptr, n_rows, 0, n_rows);
  if (rc) return rc;
  const size_t eb = elem_bytes(c);
  if ((rc = ensure_commit_buffers(m, n_rows, true))) return rc;
  if (!m->d_comm || m->cap_comm_rows < n_rows) {           // (Brakedown, position-major path: ensure_commit_buffers leaves comm for later)
    dev_free(m->d_comm); m->d_comm = nullptr; m->cap_comm_rows = 0;
    if ((rc = dev_alloc(&m->err, &m->d_comm, (size_t)n_rows * c->n_cols * eb))) return rc;
    m->cap_comm_rows = n_rows;
  }
  HIPCHK(m, hipMemcpy(m->d_comm, comm, (size_t)n_rows * c->n_cols * eb, hipMemcpyHostToDevice));
  if (c->comm_canon) HIPCHK(m, launch_to_canon(c->NL, m->d_comm, n_rows * c->n_cols, m->d_comm, nullptr));
  if (coeffs) HIPCHK(m, hipMemcpy(m->d_coeffs, coeffs, (size_t)n_rows * c->n_per_row * eb, hipMemcpyHostToDevice));
  else HIPCHK(m, hipMemset(m->d_coeffs, 0, (size_t)n_rows * c->n_per_row * eb));
  m->coeffs_view = m->d_coeffs;
  if (m->timing) { HIPCHK(m, hipEventRecord(m->ev[0], nullptr)); }
  return commit_tail(m, nullptr, root);
  LCPC_CATCH(m)
}

// ---- serde of LcCommit (lib.rs:186-268), bincode 1.3 ---------------------------------------------------------------
uint64_t lcpc_commit_bincode_size(const lcpc_commit_t* m) {
  if (!m || !m->committed || m->enc->prm.shard_count > 1) return 0;
  const lcpc_ctx* c = m->enc;
  const uint64_t eb = elem_bytes(c);
  return 8 + m->n_rows * c->n_cols * eb + 8 + m->n_rows * c->n_per_row * eb + 24 + 8 + (2 * c->np2 - 1) * 40;
}

int lcpc_commit_bincode_write(lcpc_commit_t* m, lcpc_write_fn fn, void* user) {
  if (!m || !fn) return LCPC_ERR_ARG;
  if (!m->committed || m->enc->prm.shard_count > 1) return LCPC_ERR_STATE;
  const lcpc_ctx* c = m->enc;
  LCPC_TRY
  const uint64_t eb = elem_bytes(c);
  auto put64 = [&](uint64_t v) { return fn(user, reinterpret_cast<const uint8_t*>(&v), 8); };
  // comm, then coeffs: row batches of <= 64 MiB through one staging buffer (lcpc_get_comm converts a canonical device copy
  // back to Montgomery form and materialises Brakedown's row-major view)
  struct Part { uint64_t per_row; int (*get)(lcpc_commit_t*, uint64_t, uint64_t, uint64_t*); };
  const Part parts[2] = {{c->n_cols, lcpc_get_comm}, {c->n_per_row, lcpc_get_coeffs}};
  for (const Part& p : parts) {
    if (put64(m->n_rows * p.per_row)) return LCPC_ERR_ARG;
    const uint64_t row_b = p.per_row * eb;
    const uint64_t batch = std::max<uint64_t>(1, ((uint64_t)64 << 20) / row_b);
    std::vector<uint64_t> buf((size_t)(std::min(batch, m->n_rows) * row_b / 8));
    for (uint64_t r = 0; r < m->n_rows; r += batch) {
      const uint64_t nb = std::min(batch, m->n_rows - r);
      int rc = p.get(m, r, nb, buf.data());
      if (rc) return rc;
      for (uint64_t off = 0; off < nb * row_b; off += (uint64_t)64 << 20)      // (a single row may exceed 64 MiB)
        if (fn(user, reinterpret_cast<const uint8_t*>(buf.data()) + off, std::min<uint64_t>((uint64_t)64 << 20, nb * row_b - off))) return LCPC_ERR_ARG;
    }
  }
  if (put64(m->n_rows) || put64(c->n_cols) || put64(c->n_per_row)) return LCPC_ERR_ARG;
  const uint64_t nh = 2 * c->np2 - 1;
  std::vector<uint8_t> h((size_t)nh * 32), w((size_t)nh * 40);
  int rc = lcpc_get_hashes(m, h.data());
  if (rc) return rc;
  for (uint64_t i = 0; i < nh; i++) {
    const uint64_t l = 32;
    memcpy(&w[i * 40], &l, 8);
    memcpy(&w[i * 40 + 8], &h[i * 32], 32);
  }
  if (put64(nh)) return LCPC_ERR_ARG;
  for (uint64_t off = 0; off < w.size(); off += (uint64_t)64 << 20)
    if (fn(user, w.data() + off, std::min<uint64_t>((uint64_t)64 << 20, w.size() - off))) return LCPC_ERR_ARG;
  return 0;
  LCPC_CATCH(m)
}

int lcpc_commit_from_bincode(lcpc_commit_t* m, lcpc_read_fn fn, void* user, uint8_t* root) {
  if (!m || !fn) return LCPC_ERR_ARG;
  const lcpc_ctx* c = m->enc;
  if (c->prm.shard_count > 1) return LCPC_ERR_STATE;
  LCPC_TRY
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  const FieldDesc& f = *c->f;
  const uint64_t eb = elem_bytes(c);
  bool bad_read = false;
  auto get64 = [&]() { uint64_t v = 0; if (fn(user, reinterpret_cast<uint8_t*>(&v), 8)) bad_read = true; return v; };
  const uint64_t n_comm = get64();
  if (bad_read) return LCPC_ERR_ARG;
  if (n_comm == 0 || n_comm % c->n_cols) return LCPC_ERR_COMMIT;             // check_comm: comm.len() == n_rows * n_cols
  const uint64_t n_rows = n_comm / c->n_cols;
  if (n_rows > ((uint64_t)1 << 40) / c->n_cols) return LCPC_ERR_COMMIT;
  {
    // untrusted length: nothing is freed or allocated for a commitment that could not fit this device anyway
    size_t free_b = 0, total_b = 0;
    HIPCHK(m, hipMemGetInfo(&free_b, &total_b));
    if (n_rows * (c->n_cols + c->n_per_row) > total_b / eb) return LCPC_ERR_COMMIT;
  }
  int rc = begin_commit(m, nullptr, n_rows, 0, n_rows);
  if (rc) return rc;
  if ((rc = ensure_commit_buffers(m, n_rows, true))) return rc;
  if (!m->d_comm || m->cap_comm_rows < n_rows) {           // (Brakedown, position-major path: ensure_commit_buffers leaves comm for later)
    dev_free(m->d_comm); m->d_comm = nullptr; m->cap_comm_rows = 0;
    if ((rc = dev_alloc(&m->err, &m->d_comm, (size_t)n_rows * c->n_cols * eb))) return rc;
    m->cap_comm_rows = n_rows;
  }
  std::vector<uint64_t> buf;
  // an element vector of the stream -> device, a piece at a time; untrusted limbs: nothing >= p reaches device arithmetic
  auto upload = [&](uint32_t* dst, uint64_t n_elems) -> int {
    const uint64_t piece = ((uint64_t)64 << 20) / eb;
    buf.resize((size_t)(std::min(piece, n_elems) * eb / 8));
    for (uint64_t e = 0; e < n_elems; e += piece) {
      const uint64_t ne = std::min(piece, n_elems - e);
      if (fn(user, reinterpret_cast<uint8_t*>(buf.data()), ne * eb)) return LCPC_ERR_ARG;
      std::atomic<bool> ok{true};
      parallel_for(ne, 1 << 15, [&](uint64_t b, uint64_t en) {
        for (uint64_t i = b; i < en; i++) if (h_ge_p(f, buf.data() + i * f.L)) { ok.store(false); return; }
      });
      if (!ok.load()) return LCPC_ERR_COMMIT;
      HIPCHK(m, hipMemcpy(reinterpret_cast<uint8_t*>(dst) + e * eb, buf.data(), ne * eb, hipMemcpyHostToDevice));
    }
    return 0;
  };
  if ((rc = upload(m->d_comm, n_comm))) return rc;
  if (c->comm_canon) HIPCHK(m, launch_to_canon(c->NL, m->d_comm, n_comm, m->d_comm, nullptr));
  const uint64_t n_coeffs = get64();
  if (bad_read) return LCPC_ERR_ARG;
  if (n_coeffs != n_rows * c->n_per_row) return LCPC_ERR_COMMIT;           // check_comm: coeffs.len() == n_rows * n_per_row
  if ((rc = upload(m->d_coeffs, n_coeffs))) return rc;
  m->coeffs_view = m->d_coeffs;
  const uint64_t s_rows = get64(), s_cols = get64(), s_per_row = get64(), nh = get64();
  if (bad_read) return LCPC_ERR_ARG;
  if (s_rows != n_rows || s_cols != c->n_cols || s_per_row != c->n_per_row || !lcpc_dims_ok(c, s_per_row, s_cols) || nh != 2 * c->np2 - 1)
    return LCPC_ERR_COMMIT;
  std::vector<uint8_t> w((size_t)nh * 40), have((size_t)nh * 32);
  for (uint64_t off = 0; off < w.size(); off += (uint64_t)64 << 20)
    if (fn(user, w.data() + off, std::min<uint64_t>((uint64_t)64 << 20, w.size() - off))) return LCPC_ERR_ARG;
  if (m->timing) { HIPCHK(m, hipEventRecord(m->ev[0], nullptr)); }
  if ((rc = commit_tail(m, nullptr, nullptr))) return rc;              // hash_columns + merkle_tree from comm, on the device
  m->committed = false;
  HIPCHK(m, hipMemcpy(have.data(), m->d_hashes, have.size(), hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < nh; i++) {
    uint64_t l;
    memcpy(&l, &w[i * 40], 8);
    if (l != 32 || memcmp(&w[i * 40 + 8], &have[i * 32], 32) != 0) return LCPC_ERR_COMMIT;
  }
  m->committed = true;
  if (root) memcpy(root, &have[(size_t)(nh - 1) * 32], 32);
  return 0;
  LCPC_CATCH(m)
}

int lcpc_get_root(lcpc_commit_t* m, uint8_t root[32]) {
  if (!m || !root) return LCPC_ERR_ARG;
  if (!m->committed) return LCPC_ERR_STATE;
  HIPCHK(m, hipSetDevice(m->enc->prm.device));
  { int orc = order_after_commit(m, nullptr); if (orc) return orc; }
  HIPCHK(m, hipMemcpy(root, m->d_hashes + (2 * m->enc->np2 - 2) * 8, 32, hipMemcpyDeviceToHost));
  return 0;
}
int lcpc_commit_dims(const lcpc_commit_t* m, uint64_t* nr, uint64_t* np, uint64_t* nc, uint64_t* nh) {
  if (!m) return LCPC_ERR_ARG;
  if (!m->committed) return LCPC_ERR_STATE;
  if (nr) *nr = m->n_rows;
  if (np) *np = m->enc->n_per_row;
  if (nc) *nc = m->enc->n_cols;
  if (nh) *nh = 2 * m->enc->np2 - 1;
  return 0;
}
int lcpc_get_hashes(lcpc_commit_t* m, uint8_t* hashes) {
  if (!m || !hashes) return LCPC_ERR_ARG;
  if (!m->committed) return LCPC_ERR_STATE;
  HIPCHK(m, hipSetDevice(m->enc->prm.device));
  { int orc = order_after_commit(m, nullptr); if (orc) return orc; }
  HIPCHK(m, hipMemcpy(hashes, m->d_hashes, (size_t)(2 * m->enc->np2 - 1) * 32, hipMemcpyDeviceToHost));
  return 0;
}
int lcpc_get_comm(lcpc_commit_t* m, uint64_t row0, uint64_t n, uint64_t* out) {
  if (!m || !out) return LCPC_ERR_ARG;
  if (!m->committed) return LCPC_ERR_STATE;
  if (row0 < m->row_begin || row0 + n > m->row_begin + m->n_rows_local) return LCPC_ERR_ARG;
  const lcpc_ctx* c = m->enc;
  LCPC_TRY
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  { int orc = order_after_commit(m, nullptr); if (orc) return orc; }
  const size_t eb = elem_bytes(c);
  if (m->comm_t && !m->comm_rows_valid) {      // Brakedown: the commitment is position-major; make the row-major view once
    if (!m->d_comm || m->cap_comm_rows < m->n_rows_local) {
      dev_free(m->d_comm); m->d_comm = nullptr; m->cap_comm_rows = 0;
      int rc = dev_alloc(&m->err, &m->d_comm, (size_t)m->n_rows_local * c->n_cols * eb);
      if (rc) return rc;
      m->cap_comm_rows = m->n_rows_local;
    }
    HIPCHK(m, launch_transpose_from_t(c->NL, m->ws.d_t, c->n_cols, m->n_rows_local, m->d_comm, c->n_cols, nullptr));
    HIPCHK(m, hipStreamSynchronize(nullptr));
    m->comm_rows_valid = true;
  }
  const uint32_t* src = m->d_comm + (size_t)(row0 - m->row_begin) * c->n_cols * c->NL;
  if (!(m->comm_t || c->comm_canon)) {
    HIPCHK(m, hipMemcpy(out, src, (size_t)n * c->n_cols * eb, hipMemcpyDeviceToHost));
    return 0;
  }
  // canonical on the device, Montgomery form (as ff_derive stores elements) at the ABI: convert a row batch at a time
  const uint64_t batch = std::max<uint64_t>(1, ((uint64_t)64 << 20) / (c->n_cols * eb));
  uint32_t* tmp = nullptr;
  int rc = dev_alloc(&m->err, &tmp, (size_t)std::min(batch, n ? n : 1) * c->n_cols * eb);
  if (rc) return rc;
  for (uint64_t r = 0; r < n; r += batch) {
    const uint64_t nb = std::min(batch, n - r);
    hipError_t he = launch_to_mont(c->NL, src + (size_t)r * c->n_cols * c->NL, nb * c->n_cols, c->d_r2, tmp, nullptr);
    if (he == hipSuccess)
      he = hipMemcpy(reinterpret_cast<uint8_t*>(out) + (size_t)r * c->n_cols * eb, tmp, (size_t)nb * c->n_cols * eb, hipMemcpyDeviceToHost);
    if (he != hipSuccess) { dev_free(tmp); return fail_hip(&m->err, he, "lcpc_get_comm"); }
  }
  dev_free(tmp);
  return 0;
  LCPC_CATCH(m)
}
int lcpc_get_coeffs(lcpc_commit_t* m, uint64_t row0, uint64_t n, uint64_t* out) {
  if (!m || !out) return LCPC_ERR_ARG;
  if (!m->committed) return LCPC_ERR_STATE;
  if (row0 < m->row_begin || row0 + n > m->row_begin + m->n_rows_local) return LCPC_ERR_ARG;
  const lcpc_ctx* c = m->enc;
  HIPCHK(m, hipSetDevice(c->prm.device));
  { int orc = order_after_commit(m, nullptr); if (orc) return orc; }
  const size_t eb = elem_bytes(c);
  HIPCHK(m, hipMemcpy(out, reinterpret_cast<const uint8_t*>(m->coeffs_view) + (size_t)(row0 - m->row_begin) * c->n_per_row * eb,
                      (size_t)n * c->n_per_row * eb, hipMemcpyDeviceToHost));
  return 0;
}

// ---- collapse / open ---------------------------------------------------------------------------------
int lcpc_collapse_device(lcpc_commit_t* m, const uint64_t* tensors_dev, uint32_t n_tensors, void* stream, uint64_t* polys_dev) {
  if (!m || !tensors_dev || !polys_dev || n_tensors == 0) return LCPC_ERR_ARG;
  if (!m->committed) return LCPC_ERR_STATE;
  LCPC_TRY
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(m->enc->prm.device));
  int rc = ensure_scratch(m, collapse_scratch_bytes(m, 2) + 256);
  if (rc) return rc;
  if ((rc = order_after_commit(m, (hipStream_t)stream))) return rc;
  return collapse_run(m, reinterpret_cast<const uint32_t*>(tensors_dev), n_tensors, (hipStream_t)stream,
                      reinterpret_cast<uint32_t*>(polys_dev));
  LCPC_CATCH(m)
}

int lcpc_collapse(lcpc_commit_t* m, const uint64_t* tensors, uint32_t n_tensors, uint64_t* polys) {
  LCPC_TRY
  return collapse_host(m, tensors, n_tensors, polys, nullptr);
  LCPC_CATCH(m)
}

int lcpc_open_columns(lcpc_commit_t* m, const uint64_t* cols, uint32_t n, uint64_t* col_vals, uint8_t* paths) {
  LCPC_TRY
  return open_columns_host(m, cols, n, col_vals, 0, paths);
  LCPC_CATCH(m)
}

int lcpc_set_timing(lcpc_commit_t* m, int enable) { if (!m) return LCPC_ERR_ARG; m->timing = enable != 0; return 0; }
int lcpc_get_timings(lcpc_commit_t* m, lcpc_timings* out) { if (!m || !out) return LCPC_ERR_ARG; *out = m->last; return 0; }

}  // extern "C"
