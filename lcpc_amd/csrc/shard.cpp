// lcpc_amd/csrc/shard.cpp -- row-sharded commit / prove across the GPUs of a node.
//
// commit (lcpc-2d/src/lib.rs:622-671) shards by rows: the encode (lib.rs:648-653) is independent per row and the
// column hash (lib.rs:706-745) is a BLAKE3 tree over the leaf message, so a rank that owns a chunk-aligned block of
// rows reduces it to subtree chaining values on its own.  The single exchange step is one all-gather of those
// (here: ncclAllGather on RCCL over xGMI, or a caller-supplied all-gather); every rank then finishes the leaf
// digests and builds the Merkle tree redundantly.  prove (lib.rs:1004-1093): collapse_columns (1095-1123) is a sum
// over rows -> partial sums per rank + all-gather + sum mod p; open_column (788-825) gathers per rank + all-gather.
#include "internal.h"
#include <dlfcn.h>

using namespace lcpc;

namespace lcpc {

// Chunk range of shard g of G for elements of F bytes: the leaf message 0^32 || repr(col[0]) || ... (lib.rs:719-735) is cut where
// a 1 KiB BLAKE3 chunk boundary is also a row boundary, (1024 c - 32) mod F == 0.  For F | 1024 (Ft63 / Ft127 / Ft255) that is
// every chunk and the split is the even one, n_chunks g / G; for Ft191 (F = 24) it is every third chunk -- c = 2 (mod 3), i.e.
// rows = 84 (mod 128): 128 rows are exactly three chunks -- and the even split is moved down to the nearest such boundary.
void shard_chunk_range(uint64_t F, uint64_t n_chunks, uint64_t G, uint64_t g, uint64_t* c0, uint64_t* c1) {
  auto bound = [&](uint64_t i) -> uint64_t {
    if (i == 0) return 0;
    if (i >= G) return n_chunks;
    uint64_t c = n_chunks * i / G;
    while (c > 0 && (c * 1024 - 32) % F != 0) c--;
    return c;
  };
  if (G <= 1) { *c0 = 0; *c1 = n_chunks; return; }
  *c0 = bound(g); *c1 = bound(g + 1);
}

// rows / chunks of shard `g` (DESIGN.md "multi-GPU"): chunk-aligned row blocks
void shard_layout_of(const lcpc_ctx* c, uint64_t g, uint64_t n_rows, uint64_t* rb, uint64_t* re, uint64_t* cb, uint64_t* ce, uint64_t* nch) {
  const uint64_t n_chunks = leaf_chunks(c, n_rows);
  const uint64_t G = c->prm.shard_count > 1 ? c->prm.shard_count : 1;
  if (G == 1) g = 0;
  const uint64_t F = elem_bytes(c);
  uint64_t c0, c1;
  shard_chunk_range(F, n_chunks, G, g, &c0, &c1);
  auto first_row = [&](uint64_t chunk) -> uint64_t {   // the row whose bytes start this chunk (shard_chunk_range: exact)
    if (chunk == 0) return 0;
    const uint64_t byte = chunk * 1024 - 32;
    uint64_t r = byte / F;
    return r < n_rows ? r : n_rows;
  };
  *cb = c0; *ce = c1; *nch = n_chunks;
  *rb = first_row(c0);
  *re = c1 >= n_chunks ? n_rows : first_row(c1);
  if (c0 == c1) *re = *rb;
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

// ---- RCCL, loaded at run time ---------------------------------------------------------------------------
// The library itself does not link librccl: a single-GPU host never needs it, and a process that already carries a
// copy (torch bundles one) must keep using that one.  Prototypes as in rccl.h (NCCL 2.x ABI).
namespace {
typedef struct { char internal[128]; } NcclUniqueId;
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(void** comm, int nranks, NcclUniqueId id, int rank) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) = nullptr;
  int (*Broadcast)(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t st) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    // a copy already in the process (torch bundles one) wins, under whatever name it was loaded: every candidate name with
    // RTLD_NOLOAD first, then the global symbol scope (a copy loaded by full path with RTLD_GLOBAL), and only then a load
    // LCPC_RCCL_LIB=<path>: this library and no other (a site's own RCCL build; the tests' in-process stand-in).  The path is
    // loaded and called as the communicator library: a trusted-environment switch, ignored (secure_getenv) in set-uid / set-gid
    // or capability-raised processes
    if (const char* ov = secure_getenv("LCPC_RCCL_LIB")) {
      x.h = dlopen(ov, RTLD_NOW | RTLD_LOCAL);
      if (!x.h) return x;
    }
    for (const char* n : names) {
      if (x.h) break;
      x.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    }
    if (!x.h) {
      Dl_info info;
      void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
      if (sym && dladdr(sym, &info) && info.dli_fname) x.h = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);
    }
    for (const char* n : names) {
      if (x.h) break;
      x.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!x.h) return x;
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.h, "ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.h, "ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.h, "ncclCommDestroy"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.h, "ncclAllGather"));
    x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(dlsym(x.h, "ncclBroadcast"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.h, "ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.h, "ncclGroupEnd"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.h, "ncclGetErrorString"));
    x.GetVersion = reinterpret_cast<decltype(x.GetVersion)>(dlsym(x.h, "ncclGetVersion"));      // (optional)
    x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.Broadcast && x.GroupStart && x.GroupEnd;
    return x;
  }();
  return r;
}
constexpr int NCCL_UINT8 = 1;      // ncclUint8 (rccl.h)
int fail_nccl(std::string* err, int rc, const char* what) {
  if (err) *err = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "nccl error") ;
  return LCPC_ERR_XCHG;
}
// collectives of one communicator in submission order, whatever streams carry them (call with xchg_mu held)
hipError_t xchg_order_before(lcpc_ctx* c, hipStream_t s) {
  if (!c->ev_xchg) return hipEventCreateWithFlags(&c->ev_xchg, hipEventDisableTiming);
  return hipStreamWaitEvent(s, c->ev_xchg, 0);
}
hipError_t xchg_order_after(lcpc_ctx* c, hipStream_t s) { return hipEventRecord(c->ev_xchg, s); }
// lcpc_allgather_fn on the encoder's communicator: enqueued on the commitment's prove stream, nothing waits on the host
int rccl_allgather_cb(void* user, uint64_t bytes) {
  lcpc_commit_t* m = static_cast<lcpc_commit_t*>(user);
  std::lock_guard<std::mutex> xg(m->enc->xchg_mu);
  if (xchg_order_before(m->enc, m->s_prove) != hipSuccess) { m->err = "hipStreamWaitEvent (exchange order)"; return 1; }
  int rc = rccl().AllGather(m->d_xsend, m->d_xrecv, (size_t)bytes, NCCL_UINT8, m->enc->comm, m->s_prove);
  if (rc != 0) { fail_nccl(&m->err, rc, "ncclAllGather"); return 1; }
  if (xchg_order_after(m->enc, m->s_prove) != hipSuccess) { m->err = "hipEventRecord (exchange order)"; return 1; }
  return 0;
}
}  // namespace

void comm_release(lcpc_ctx* c) {
  if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  c->comm = nullptr;
  if (c->ev_xchg) { (void)hipEventDestroy(c->ev_xchg); c->ev_xchg = nullptr; }
}

// ---- sharded prove pieces ---------------------------------------------------------------------------------
// the commitment's prove stream; (re)ordered behind whatever stream the last sharded commit was enqueued on
int prove_stream(lcpc_commit_t* m, hipStream_t* st) {
  if (!m->s_prove) HIPCHK(m, hipStreamCreateWithFlags(&m->s_prove, hipStreamNonBlocking));
  if (m->ev_done) HIPCHK(m, hipStreamWaitEvent(m->s_prove, m->ev_done, 0));
  *st = m->s_prove;
  return 0;
}

// collapse over ALL rows of a sharded commitment: local partial sums, all-gather, sum mod p (lib.rs:1095-1123 split by rows);
// one host synchronisation, at the end, when the host needs the polynomials
int collapse_sharded(lcpc_commit_t* m, const ShardXchg& x, const uint64_t* tensors_full, uint32_t nt, uint64_t* polys, uint64_t* polys_canon) {
  const lcpc_ctx* c = m->enc;
  const size_t eb = elem_bytes(c);
  const int L = c->L;
  const uint64_t bytes = (uint64_t)nt * c->n_per_row * eb;
  if (bytes > x.max_bytes) return LCPC_ERR_ARG;
  hipStream_t st = nullptr;
  uint32_t* d_canon = nullptr;
  {
    std::lock_guard<std::mutex> g(m->mu);
    HIPCHK(m, hipSetDevice(c->prm.device));
    int rc = prove_stream(m, &st);
    if (rc) return rc;
    const size_t row_b = m->n_rows_local * eb;
    const size_t tb = ((size_t)nt * row_b + 255) & ~(size_t)255, pb = (bytes + 255) & ~(size_t)255;
    if ((rc = ensure_scratch(m, tb + pb + collapse_scratch_bytes(m, 2) + 512))) return rc;
    uint32_t* d_t = m->d_scratch;
    d_canon = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(m->d_scratch) + tb);
    if (m->n_rows_local == 0) {
      HIPCHK(m, hipMemsetAsync(x.send_dev, 0, bytes, st));
    } else {
      for (uint32_t t = 0; t < nt; t++)      // this rank's slice of every tensor, straight from the caller's buffer
        HIPCHK(m, hipMemcpyAsync(reinterpret_cast<uint8_t*>(d_t) + (size_t)t * row_b, tensors_full + ((size_t)t * m->n_rows + m->row_begin) * L, row_b,
                                 hipMemcpyHostToDevice, st));
      if ((rc = collapse_run(m, d_t, nt, st, reinterpret_cast<uint32_t*>(x.send_dev)))) return rc;
    }
    if (!x.stream_ordered) HIPCHK(m, hipStreamSynchronize(st));       // a host-driven exchange reads send_dev next
  }
  if (x.fn(x.user, bytes) != 0) return LCPC_ERR_XCHG;
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  HIPCHK(m, launch_field_sum(c->NL, reinterpret_cast<const uint32_t*>(x.recv_dev), std::max<uint32_t>(1, c->prm.shard_count), (uint64_t)nt * c->n_per_row,
                             reinterpret_cast<uint32_t*>(x.send_dev), st));
  HIPCHK(m, hipMemcpyAsync(polys, x.send_dev, bytes, hipMemcpyDeviceToHost, st));
  if (polys_canon) {
    HIPCHK(m, launch_to_canon(c->NL, reinterpret_cast<const uint32_t*>(x.send_dev), (uint64_t)nt * c->n_per_row, d_canon, st));
    HIPCHK(m, hipMemcpyAsync(polys_canon, d_canon, bytes, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(m, hipStreamSynchronize(st));
  return 0;
}

// open_column for n columns of a sharded commitment: every rank gathers its rows straight into the send buffer, one
// all-gather, the columns are put in row order on the device and land in the caller's (strided) buffer by one copy;
// the Merkle paths come from the (replicated) tree
int open_sharded(lcpc_commit_t* m, const ShardXchg& x, const uint64_t* cols, uint32_t n, uint64_t* vals, size_t vals_pitch, uint8_t* paths) {
  const lcpc_ctx* c = m->enc;
  const size_t eb = elem_bytes(c);
  const uint32_t G = std::max<uint32_t>(1, c->prm.shard_count);
  for (uint32_t i = 0; i < n; i++)
    if (cols[i] >= c->n_cols) return LCPC_ERR_COLUMN_NUMBER;
  std::vector<uint64_t> rb(G + 1);
  uint64_t max_rows = 0;
  for (uint32_t g = 0; g < G; g++) {
    uint64_t re, cb, ce, nch;
    shard_layout_of(c, g, m->n_rows, &rb[g], &re, &cb, &ce, &nch);
    max_rows = std::max(max_rows, re - rb[g]);
    rb[g + 1] = re;
  }
  const uint64_t bytes = (uint64_t)n * max_rows * eb;
  if (bytes > x.max_bytes) return LCPC_ERR_ARG;
  hipStream_t st = nullptr;
  uint32_t* d_out = nullptr;
  uint64_t* d_rb = nullptr;
  const size_t col_b = (size_t)m->n_rows * eb;
  {
    std::lock_guard<std::mutex> g(m->mu);
    HIPCHK(m, hipSetDevice(c->prm.device));
    int rc = prove_stream(m, &st);
    if (rc) return rc;
    const size_t cb = (((size_t)n * 8) + 255) & ~(size_t)255, pb = (((size_t)n * c->path_len * 32) + 255) & ~(size_t)255;
    const size_t rbb = (((size_t)(G + 1) * 8) + 255) & ~(size_t)255, ob = (((size_t)n * col_b) + 255) & ~(size_t)255;
    if ((rc = ensure_scratch(m, cb + pb + rbb + ob))) return rc;
    uint8_t* base = reinterpret_cast<uint8_t*>(m->d_scratch);
    uint64_t* d_cols = reinterpret_cast<uint64_t*>(base);
    uint32_t* d_paths = reinterpret_cast<uint32_t*>(base + cb);
    d_rb = reinterpret_cast<uint64_t*>(base + cb + pb);
    d_out = reinterpret_cast<uint32_t*>(base + cb + pb + rbb);
    HIPCHK(m, hipMemcpyAsync(d_cols, cols, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIPCHK(m, hipMemcpyAsync(d_rb, rb.data(), (size_t)(G + 1) * 8, hipMemcpyHostToDevice, st));
    if ((rc = open_columns_device(m, d_cols, n, reinterpret_cast<uint32_t*>(x.send_dev), paths ? d_paths : nullptr, st))) return rc;
    if (paths && c->path_len) HIPCHK(m, hipMemcpyAsync(paths, d_paths, (size_t)n * c->path_len * 32, hipMemcpyDeviceToHost, st));
    if (!x.stream_ordered) HIPCHK(m, hipStreamSynchronize(st));
  }
  if (x.fn(x.user, bytes) != 0) return LCPC_ERR_XCHG;
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  HIPCHK(m, launch_assemble_columns(c->NL, reinterpret_cast<const uint32_t*>(x.recv_dev), bytes / 4, d_rb, G, n, m->n_rows, d_out, st));
  if (vals_pitch == 0 || vals_pitch == col_b) HIPCHK(m, hipMemcpyAsync(vals, d_out, (size_t)n * col_b, hipMemcpyDeviceToHost, st));
  else HIPCHK(m, hipMemcpy2DAsync(vals, vals_pitch, d_out, col_b, col_b, n, hipMemcpyDeviceToHost, st));
  HIPCHK(m, hipStreamSynchronize(st));        // (also keeps `rb` and `cols` alive until their copies are done)
  return 0;
}

// ---- sharded commit phases ----------------------------------------------------------------------------------
// The commit of a rank is four steps:
//   encode        local rows -> comm                                                     (lib.rs:648-653)
//   hash_cols     the columns of the local rows -> node chaining values [k][n_cols]      (lib.rs:706-745, this rank's part)
//   finish_cols   gathered node CVs -> leaf digests hashes[0, n_cols)                    (lib.rs:706-745, the rest)
//   merkle        the tree above the leaf digests                                        (lib.rs:747-785)
// (Round 4 could slice the middle two by column ranges so that one slice's exchange overlapped the next slice's hashing: measured
// neutral to negative -- LABNOTES "exchange slicing" -- and removed in round 5 together with its four column-range entry points.)
static int shard_encode_phase(lcpc_commit_t* m, const uint64_t* coeffs_local, uint64_t n_rows_total, hipStream_t st, uint32_t flags) {
  const lcpc_ctx* c = m->enc;
  uint64_t rb, re, cb, ce, nch;
  shard_layout_of(c, c->prm.shard_rank, n_rows_total, &rb, &re, &cb, &ce, &nch);
  int rc = order_after_commit(m, st);      // a refill on another stream than the previous fill's: behind that fill
  if (rc) return rc;
  m->committed = false;
  m->comm_t = false; m->comm_rows_valid = false; m->coeffs_view = nullptr;
  m->n_rows = n_rows_total; m->row_begin = rb; m->n_rows_local = re - rb;
  m->chunk_begin = cb; m->chunk_end = ce; m->n_chunks = nch;
  m->launches[0] = m->launches[1] = m->launches[2] = 0;
  m->last.exchange_exposed_ms = 0.f; m->last.exchange_wire_ms = 0.f;
  m->shard_encoded = false;
  const bool fused = c->prm.encoding == LCPC_ENC_LIGERO || m->n_rows_local >= SDIG_T_MIN_ROWS;
  const bool borrow = (flags & LCPC_COMMIT_BORROW_COEFFS) != 0 && m->n_rows_local > 0;   // local rows are always whole rows
  rc = ensure_commit_buffers(m, m->n_rows_local, !borrow);
  if (rc) return rc;
  if (m->timing) HIPCHK(m, hipEventRecord(m->ev[0], st));
  if (m->n_rows_local) {
    if (!coeffs_local) return LCPC_ERR_ARG;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(coeffs_local);
    EncodeJob j;
    j.src_stride = c->n_per_row; j.n_valid = c->n_per_row; j.dst = m->d_comm; j.n_rows = m->n_rows_local;
    j.canon_out = c->prm.encoding == LCPC_ENC_SDIG ? true : c->comm_canon; j.keep_t = true;
    bool kept = false;
    j.kept_t = &kept;
    if (borrow) {
      j.src = src;
      m->coeffs_view = src;
    } else if (fused) {
      j.src = src; j.copy_dst = m->d_coeffs;       // coeffs copy fused into the first pass / the input transpose
      m->coeffs_view = m->d_coeffs;
    } else {
      HIPCHK(m, hipMemcpyAsync(m->d_coeffs, coeffs_local, (size_t)m->n_rows_local * c->n_per_row * elem_bytes(c), hipMemcpyDeviceToDevice, st));
      j.src = m->d_coeffs;
      m->coeffs_view = m->d_coeffs;
    }
    if ((rc = encode_rows_device(c, &m->ws, j, st, &m->err, &m->launches[0]))) return rc;
    m->comm_t = kept;
  } else {
    m->coeffs_view = m->d_coeffs;
  }
  if (m->timing) HIPCHK(m, hipEventRecord(m->ev[1], st));
  if (ce > cb && (rc = ensure_cvs(m, ce - cb))) return rc;      // (all slices share it; sized once so that no slice reallocates)
  m->shard_encoded = true;
  return 0;
}

// the columns of the local rows -> one chaining value per (local node, column): nodes_dev[k][n_cols][32 B]
static int shard_hash_cols(lcpc_commit_t* m, hipStream_t st, uint8_t* nodes_dev) {
  const lcpc_ctx* c = m->enc;
  const uint64_t cb = m->chunk_begin, ce = m->chunk_end, w = c->n_cols;
  if (ce <= cb) return 0;
  uint64_t first[64];
  uint32_t lg[64];
  const int n_nodes = shard_nodes(cb, ce, first, lg);
  bool all_single = true;
  for (int k = 0; k < n_nodes; k++) all_single = all_single && lg[k] == 0;
  LeafArgs la{};
  la.comm = m->d_comm; la.canon_in = c->comm_canon ? 1u : 0u; la.row_stride = c->n_cols; la.col_stride = 1;
  if (m->comm_t) { la.comm = m->ws.d_t; la.canon_in = 1u; la.row_stride = 1; la.col_stride = m->n_rows_local; }
  la.n_cols = w; la.row_base = (int64_t)m->row_begin;
  la.n_rows_total = m->n_rows; la.chunk_begin = (uint32_t)cb; la.n_chunks_local = (uint32_t)(ce - cb);
  la.n_chunks_total = (uint32_t)m->n_chunks;
  if (all_single) {                       // nothing to pre-merge: chunk CVs are the nodes
    la.out = reinterpret_cast<uint32_t*>(nodes_dev);
    HIPCHK(m, launch_leaf_chunks(c->NL, la, st));
    m->launches[1]++;
    return 0;
  }
  uint32_t* cvs = m->d_cvs;
  la.out = cvs;
  HIPCHK(m, launch_leaf_chunks(c->NL, la, st));
  m->launches[1]++;
  // a rank that owns the WHOLE message as one node (a power-of-two number of chunks and nobody else has any: world = 1, or an
  // Ft191 commitment too short for a cut) produces the digest itself -- the parent of the last merge is the tree's root and
  // takes the ROOT flag here; the finish step then only copies it, as for a single-chunk message
  const bool whole = n_nodes == 1 && cb == 0 && ce == m->n_chunks;
  for (int k = 0; k < n_nodes; k++) {   // one subtree CV per aligned block of chunks
    uint32_t* blk = cvs + (first[k] - cb) * w * 8;
    uint32_t* out = reinterpret_cast<uint32_t*>(nodes_dev) + (size_t)k * w * 8;
    HIPCHK(m, launch_leaf_finish_nodes(blk, nullptr, nullptr, 1u << lg[k], w, out, whole, st));
    m->launches[1]++;
  }
  return 0;
}

// node table over all ranks, in chunk order: slot in the gathered buffer + log2(size); cached per shape.  Slot of rank g's
// node k: padded layout (slots_per_rank > 0) g * slots_per_rank + k; compact layout (slots_per_rank == 0, the native
// exchange) g for k = 0 and G + (running index over ranks of their nodes k >= 1) otherwise
static int shard_node_table(lcpc_commit_t* m, uint32_t slots_per_rank) {
  const lcpc_ctx* c = m->enc;
  const uint64_t nch = m->n_chunks;
  const uint32_t G = c->prm.shard_count > 1 ? c->prm.shard_count : 1;
  const uint64_t key = (nch << 24) ^ ((uint64_t)slots_per_rank << 8) ^ G;
  if (m->d_node_tab && m->node_tab_key == key) return 0;
  // a shape seen before: its table is still there (looked up before anything is rebuilt)
  for (const auto& t : m->node_tabs)
    if (t.key == key) { m->d_node_tab = t.d; m->node_tab_key = key; m->node_slot_h = t.slot; m->node_log_h = t.lg; return 0; }
  std::vector<uint32_t> slot, lgs;
  uint32_t extra = G;
  for (uint32_t r = 0; r < G; r++) {
    uint64_t first[64];
    uint32_t lg[64];
    uint64_t c0, c1;
    shard_chunk_range(elem_bytes(c), nch, G, r, &c0, &c1);
    const int n = shard_nodes(c0, c1, first, lg);
    if (slots_per_rank && (uint32_t)n > slots_per_rank) return LCPC_ERR_ARG;
    for (int k = 0; k < n; k++) {
      slot.push_back(slots_per_rank ? r * slots_per_rank + (uint32_t)k : (k == 0 ? r : extra++));
      lgs.push_back(lg[k]);
    }
  }
  const uint32_t nn = (uint32_t)slot.size();
  // A table that a finish step still in flight (any stream) may be reading is never freed or overwritten: a new shape gets a NEW
  // table (a few dozen bytes); an object sees one or two shapes.  One that keeps meeting new ones holds at most MAX_TABS: beyond
  // that the oldest goes, after the last enqueued finish of this object has completed (ev_done is recorded behind every one).
  // The blocking copy below writes memory nothing on the device refers to yet.
  constexpr size_t MAX_TABS = 8;
  if (m->node_tabs.size() >= MAX_TABS) {
    if (m->ev_done) HIPCHK(m, hipEventSynchronize(m->ev_done));
    dev_free(m->node_tabs.front().d);
    m->node_tabs.erase(m->node_tabs.begin());
  }
  uint32_t* d = nullptr;
  int rc = dev_alloc(&m->err, &d, (size_t)nn * 8);
  if (rc) return rc;
  hipError_t e = hipMemcpy(d, slot.data(), nn * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + nn, lgs.data(), nn * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) { dev_free(d); return fail_hip(&m->err, e, "node table upload"); }
  m->node_tabs.push_back({key, d, slot, lgs});
  m->d_node_tab = d; m->node_tab_key = key; m->node_slot_h = slot; m->node_log_h = lgs;
  return 0;
}

// gathered node CVs (gathered[slot][n_cols][32 B], clobbered) -> leaf digests hashes[0, n_cols)
static int shard_finish_cols(lcpc_commit_t* m, uint8_t* gathered, uint32_t slots_per_rank, hipStream_t st) {
  const uint64_t w = m->enc->n_cols;
  int rc = shard_node_table(m, slots_per_rank);
  if (rc) return rc;
  const uint32_t n_nodes = (uint32_t)m->node_slot_h.size();
  uint32_t* out = m->d_hashes;
  if (m->n_chunks == 1 || n_nodes == 1) {   // single-chunk message, or the whole message as one rank's one node (shard_hash_cols):
                                            // that "node" already carries ROOT
    HIPCHK(m, hipMemcpyAsync(out, gathered + (size_t)m->node_slot_h[0] * w * 32, (size_t)w * 32, hipMemcpyDeviceToDevice, st));
  } else {
    HIPCHK(m, launch_leaf_finish_nodes(reinterpret_cast<uint32_t*>(gathered), m->d_node_tab, m->d_node_tab + n_nodes, n_nodes, w, out, true, st));
    m->launches[1]++;
  }
  return 0;
}

// the Merkle tree over the finished leaf digests; the commitment is complete behind this
static int shard_merkle_phase(lcpc_commit_t* m, hipStream_t st, uint8_t* root) {
  int rc = merkle_top(m, st);
  if (rc) return rc;
  if (m->timing) {
    HIPCHK(m, hipEventRecord(m->ev[3], st));
    HIPCHK(m, hipEventSynchronize(m->ev[3]));
    (void)hipEventElapsedTime(&m->last.encode_ms, m->ev[0], m->ev[1]);
    (void)hipEventElapsedTime(&m->last.hash_ms, m->ev[1], m->ev[2]);
    (void)hipEventElapsedTime(&m->last.merkle_ms, m->ev[2], m->ev[3]);   // includes whatever of the exchange is exposed
    (void)hipEventElapsedTime(&m->last.total_ms, m->ev[0], m->ev[3]);
    m->last.encode_launches = m->launches[0]; m->last.hash_launches = m->launches[1]; m->last.merkle_launches = m->launches[2];
  }
  m->committed = true;
  m->shard_encoded = false;
  // a prove on this commitment runs on its own stream: it waits for this point of the commit's stream
  if (!m->ev_done) HIPCHK(m, hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming));
  HIPCHK(m, hipEventRecord(m->ev_done, st));
  if (root) return fetch_root(m, st, root);
  return 0;
}

}  // namespace lcpc

extern "C" {

int lcpc_shard_layout(const lcpc_ctx* c, uint64_t n_rows_total, uint64_t* rb, uint64_t* re, uint64_t* cb, uint64_t* ce, uint64_t* nch) {
  if (!c || n_rows_total == 0) return LCPC_ERR_ARG;
  uint64_t a, b, x, y, z;
  shard_layout_of(c, c->prm.shard_rank, n_rows_total, &a, &b, &x, &y, &z);
  if (rb) *rb = a;
  if (re) *re = b;
  if (cb) *cb = x;
  if (ce) *ce = y;
  if (nch) *nch = z;
  return 0;
}

int lcpc_shard_nodes_field(uint32_t field, uint64_t n_chunks, uint32_t G, uint32_t g, uint32_t* n_nodes, uint64_t* first, uint32_t* lg) {
  const FieldDesc* f = field_desc((int)field);
  if (!f || !n_nodes || !first || !lg || n_chunks == 0) return LCPC_ERR_ARG;
  if (G <= 1) { G = 1; g = 0; }
  if (g >= G) return LCPC_ERR_ARG;
  uint64_t c0, c1;
  shard_chunk_range((uint64_t)8 * f->L, n_chunks, G, g, &c0, &c1);
  *n_nodes = (uint32_t)shard_nodes(c0, c1, first, lg);
  return 0;
}
int lcpc_shard_nodes(uint64_t n_chunks, uint32_t G, uint32_t g, uint32_t* n_nodes, uint64_t* first, uint32_t* lg) {
  return lcpc_shard_nodes_field(LCPC_FT255, n_chunks, G, g, n_nodes, first, lg);    // (any field whose elements divide 1024)
}

// ---- split phases (a caller-side collective) ----------------------------------------------------------------
// A failure in any step after the encode leaves the commit un-started (shard_encoded false): the finish step of a commit whose
// hash step failed must not build a tree over stale digests.
int lcpc_commit_shard_device(lcpc_commit_t* m, const uint64_t* coeffs_local, uint64_t n_rows_total, void* stream, uint32_t flags,
                             uint8_t* nodes_dev) {
  if (!m || n_rows_total == 0 || !nodes_dev) return LCPC_ERR_ARG;
  LCPC_TRY
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(m->enc->prm.device));
  hipStream_t st = (hipStream_t)stream;
  int rc = shard_encode_phase(m, coeffs_local, n_rows_total, st, flags);
  if (!rc) rc = shard_hash_cols(m, st, nodes_dev);
  if (rc) { m->shard_encoded = false; return rc; }
  if (m->timing) HIPCHK(m, hipEventRecord(m->ev[2], st));
  return 0;
  LCPC_CATCH(m)
}

int lcpc_commit_finish_device(lcpc_commit_t* m, uint8_t* gathered, uint64_t n_rows_total, uint32_t slots_per_rank, void* stream, uint8_t* root) {
  if (!m || !gathered) return LCPC_ERR_ARG;
  LCPC_TRY
  std::lock_guard<std::mutex> g(m->mu);
  if (n_rows_total != m->n_rows) return LCPC_ERR_ARG;
  if (!m->shard_encoded) return LCPC_ERR_STATE;             // (read under the lock: the encode step may run on another host thread)
  HIPCHK(m, hipSetDevice(m->enc->prm.device));
  // the node table first: a slots_per_rank too small for this shape is an argument error that changes nothing -- the caller may
  // call again with a corrected value
  int rc = shard_node_table(m, slots_per_rank);
  if (rc == LCPC_ERR_ARG) return rc;
  if (!rc) rc = shard_finish_cols(m, gathered, slots_per_rank, (hipStream_t)stream);
  if (!rc) rc = shard_merkle_phase(m, (hipStream_t)stream, root);
  if (rc) m->shard_encoded = false;                         // a device step failed: this commit cannot be finished any more
  return rc;
  LCPC_CATCH(m)
}

// ---- native exchange (RCCL) -------------------------------------------------------------------------------
int lcpc_comm_unique_id(uint8_t id[128]) {
  if (!id) return LCPC_ERR_ARG;
  if (!rccl().ok) return LCPC_ERR_NO_RCCL;
  NcclUniqueId u;
  if (rccl().GetUniqueId(&u) != 0) return LCPC_ERR_XCHG;
  memcpy(id, u.internal, 128);
  return 0;
}

int lcpc_comm_init(lcpc_ctx* c, const uint8_t id[128], uint32_t rank, uint32_t world) {
  if (!c || !id || world == 0 || rank >= world) return LCPC_ERR_ARG;
  const uint32_t G = c->prm.shard_count > 1 ? c->prm.shard_count : 1;
  if (world != G || rank != (G > 1 ? c->prm.shard_rank : 0)) return LCPC_ERR_ARG;
  if (!rccl().ok) return LCPC_ERR_NO_RCCL;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  comm_release(c);
  NcclUniqueId u;
  memcpy(u.internal, id, 128);
  void* comm = nullptr;
  const int rc = rccl().CommInitRank(&comm, (int)world, u, (int)rank);
  if (rc != 0) return fail_nccl(&c->err, rc, "ncclCommInitRank");
  c->comm = comm;
  return 0;
}

int lcpc_comm_destroy(lcpc_ctx* c) {
  if (!c) return LCPC_ERR_ARG;
  std::lock_guard<std::mutex> g(c->mu);
  (void)hipSetDevice(c->prm.device);
  comm_release(c);
  return 0;
}

// What crosses the wire in the native exchange: node 0 of every rank (one all-gather of one slot each; slot = one chaining value
// per column) plus the few second / third nodes some ranks own (one broadcast each: at the headline only the last rank has one),
// instead of padding every rank to the largest node count.  Layout of d_gather, contiguous:
// [ this rank's nodes: my_slots ][ gathered: G slots of node 0, then the extra nodes in rank order ]
struct XchgPlan {
  uint32_t G = 1, me = 0, my_slots = 1, extras = 0;
  uint32_t n_nodes_of[256];
  uint64_t slot_bytes = 0;
  uint64_t tot_slots() const { return (uint64_t)my_slots + G + extras; }
  uint64_t bytes_in() const { return (uint64_t)(G - 1 + extras - (my_slots - 1)) * slot_bytes; }   // what this rank RECEIVES from others
};
static int xchg_plan(const lcpc_ctx* c, uint64_t n_rows_total, XchgPlan* p) {
  p->G = c->prm.shard_count > 1 ? c->prm.shard_count : 1;
  p->me = p->G > 1 ? c->prm.shard_rank : 0;
  if (p->G > 256) return LCPC_ERR_ARG;
  const uint64_t nch = leaf_chunks(c, n_rows_total);
  p->extras = 0; p->my_slots = 1;
  for (uint32_t r = 0; r < p->G; r++) {
    uint64_t first[64];
    uint32_t lg[64];
    uint64_t c0, c1;
    shard_chunk_range(elem_bytes(c), nch, p->G, r, &c0, &c1);
    p->n_nodes_of[r] = (uint32_t)shard_nodes(c0, c1, first, lg);
    if (p->n_nodes_of[r] > 1) p->extras += p->n_nodes_of[r] - 1;
    if (r == p->me && p->n_nodes_of[r] > 1) p->my_slots = p->n_nodes_of[r];
  }
  p->slot_bytes = c->n_cols * 32;
  return 0;
}
// the collectives of one exchange, enqueued on sx (the ONE exchange step of the path)
static int xchg_enqueue(lcpc_commit_t* m, const XchgPlan& p, hipStream_t sx) {
  lcpc_ctx* c = m->enc;
  uint8_t* send = m->d_gather;
  uint8_t* recv = send + p.slot_bytes * p.my_slots;
  // Collectives of one communicator must be submitted in the same order on every rank.  This lock only keeps the exchanges of
  // two commitments of ONE process from interleaving; it cannot order submissions ACROSS ranks -- one encoder's sharded
  // commits / proves must be issued in the same program order on every rank (one driving thread per communicator).
  std::lock_guard<std::mutex> xg(c->xchg_mu);
  HIPCHK(m, xchg_order_before(c, sx));
  int nrc = rccl().GroupStart();
  if (nrc == 0) nrc = rccl().AllGather(send, recv, (size_t)p.slot_bytes, NCCL_UINT8, c->comm, sx);
  uint32_t x = p.G;
  for (uint32_t r = 0; r < p.G && nrc == 0; r++)
    for (uint32_t k = 1; k < p.n_nodes_of[r] && nrc == 0; k++, x++) {
      uint8_t* dst = recv + p.slot_bytes * x;
      nrc = rccl().Broadcast(r == p.me ? send + p.slot_bytes * k : dst, dst, (size_t)p.slot_bytes, NCCL_UINT8, (int)r, c->comm, sx);
    }
  const int erc = rccl().GroupEnd();
  if (nrc == 0) nrc = erc;
  if (nrc != 0) return fail_nccl(&m->err, nrc, "ncclAllGather / ncclBroadcast");
  HIPCHK(m, xchg_order_after(c, sx));
  return 0;
}

// One commit on a row shard with the exchange inside: encode and the local column hash on `stream`; the collectives (node 0 of
// every rank by ncclAllGather, the few second / third nodes by one ncclBroadcast each, grouped), the leaf digests and the Merkle
// tree follow
//   * by default on `stream` itself, everything in sequence;
//   * with LCPC_COMMIT_ASYNC_TAIL in `flags` on the commitment's exchange stream WITHOUT `stream` waiting for them: `stream` is
//     free again after the column hash, so the next commit (another lcpc_commit_t of the same encoder) encodes while this one's
//     node values are on the wire.  The commitment is complete behind its event, which every reader and a refill wait for.
int lcpc_commit_sharded_device(lcpc_commit_t* m, const uint64_t* coeffs_local, uint64_t n_rows_total, void* stream, uint32_t flags, uint8_t* root) {
  if (!m || n_rows_total == 0) return LCPC_ERR_ARG;
  lcpc_ctx* c = m->enc;
  if (!c->comm) return LCPC_ERR_STATE;             // lcpc_comm_init first
  LCPC_TRY
  std::lock_guard<std::mutex> g(m->mu);
  HIPCHK(m, hipSetDevice(c->prm.device));
  hipStream_t st = (hipStream_t)stream;
  XchgPlan xp;
  int rc = xchg_plan(c, n_rows_total, &xp);
  if (rc) return rc;
  const uint64_t tot_slots = xp.tot_slots();
  // (the refill order of shard_encode_phase comes first: the buffer below may still be read by the previous fill's exchange)
  if ((rc = order_after_commit(m, st))) return rc;
  if (tot_slots * c->n_cols * 32 > m->gather_cap && m->ev_done) HIPCHK(m, hipEventSynchronize(m->ev_done));
  if ((rc = ensure_dev(&m->err, &m->d_gather, &m->gather_cap, tot_slots * c->n_cols * 32))) return rc;
  const bool async_tail = (flags & LCPC_COMMIT_ASYNC_TAIL) != 0;
  if ((rc = shard_encode_phase(m, coeffs_local, n_rows_total, st, flags))) return rc;
  // from here on a failure leaves the object un-committed AND un-started (no split-phase finish may pick it up)
  struct Unstart { lcpc_commit_t* m; bool armed = true; ~Unstart() { if (armed) m->shard_encoded = false; } } unstart{m};
  if ((rc = shard_node_table(m, 0))) return rc;
  hipStream_t sx = st;
  if (async_tail) {
    if (!m->s_xchg) HIPCHK(m, hipStreamCreateWithFlags(&m->s_xchg, hipStreamNonBlocking));
    if (!m->ev_hashed) HIPCHK(m, hipEventCreateWithFlags(&m->ev_hashed, hipEventDisableTiming));
    sx = m->s_xchg;
  }
  uint8_t* send = m->d_gather;
  uint8_t* recv = send + xp.slot_bytes * xp.my_slots;
  if ((rc = shard_hash_cols(m, st, send))) return rc;
  if (sx != st) {
    HIPCHK(m, hipEventRecord(m->ev_hashed, st));
    HIPCHK(m, hipStreamWaitEvent(sx, m->ev_hashed, 0));
  }
  if (m->timing) HIPCHK(m, hipEventRecord(m->ev[2], st));
  if ((rc = xchg_enqueue(m, xp, sx))) return rc;
  if (m->timing) HIPCHK(m, hipEventRecord(m->ev[5], sx));      // the wire is done
  if ((rc = shard_finish_cols(m, recv, 0, sx))) return rc;
  if (m->timing) HIPCHK(m, hipEventRecord(m->ev[4], sx));
  if ((rc = shard_merkle_phase(m, sx, root))) return rc;      // (async tail: `st` is not held up, it is free for the next commit's encode)
  unstart.armed = false;
  if (m->timing) {
    (void)hipEventElapsedTime(&m->last.exchange_exposed_ms, m->ev[2], m->ev[4]);
    (void)hipEventElapsedTime(&m->last.exchange_wire_ms, m->ev[2], m->ev[5]);
  }
  return 0;
  LCPC_CATCH(m)
}

// measurement hook: the exchange of the last lcpc_commit_sharded_device of this object ALONE, once more, on `stream` -- the same
// collectives on the same buffers (the node values this rank sent are still in place; what arrives overwrites the gathered
// area, which the finished commit no longer reads).  Enqueues only.  A collective: every rank calls it, in the same order.
int lcpc_shard_exchange_probe(lcpc_commit_t* m, void* stream, uint64_t* bytes_in) {
  if (!m) return LCPC_ERR_ARG;
  lcpc_ctx* c = m->enc;
  if (!c->comm) return LCPC_ERR_STATE;
  LCPC_TRY
  std::lock_guard<std::mutex> g(m->mu);
  if (!m->committed || !m->d_gather) return LCPC_ERR_STATE;
  HIPCHK(m, hipSetDevice(c->prm.device));
  XchgPlan xp;
  int rc = xchg_plan(c, m->n_rows, &xp);
  if (rc) return rc;
  if (xp.tot_slots() * xp.slot_bytes > m->gather_cap) return LCPC_ERR_STATE;
  if (bytes_in) *bytes_in = xp.bytes_in();
  hipStream_t st = (hipStream_t)stream;
  if ((rc = order_after_commit(m, st))) return rc;           // behind the commit (and its async tail) that owns the buffer
  return xchg_enqueue(m, xp, st);
  LCPC_CATCH(m)
}

// NCCL_VERSION_CODE of the loaded communicator library (ncclGetVersion), 0 if it has none; LCPC_ERR_NO_RCCL without a library
int lcpc_comm_rccl_version(int* version) {
  if (!version) return LCPC_ERR_ARG;
  if (!rccl().ok) return LCPC_ERR_NO_RCCL;
  *version = 0;
  if (rccl().GetVersion) (void)rccl().GetVersion(version);
  return 0;
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

int lcpc_prove_sharded(lcpc_commit_t* m, const uint64_t* outer, uint64_t n_outer, lcpc_transcript* trw, uint8_t* send_dev, uint8_t* recv_dev,
                       uint64_t max_bytes, lcpc_allgather_fn fn, void* user, uint8_t** proof, uint64_t* proof_len,
                       uint64_t* cols_opened) {
  if (!m || !send_dev || !recv_dev || !fn) return LCPC_ERR_ARG;
  if (m->enc->prm.shard_count <= 1) return LCPC_ERR_STATE;
  if (m->committed && max_bytes < lcpc_prove_sharded_bytes(m->enc, m->n_rows)) return LCPC_ERR_ARG;
  LCPC_TRY
  const ShardXchg x{send_dev, recv_dev, max_bytes, fn, user};
  return prove_impl(m, outer, n_outer, trw, proof, proof_len, cols_opened, &x);
  LCPC_CATCH(m)
}

int lcpc_prove_sharded_rccl(lcpc_commit_t* m, const uint64_t* outer, uint64_t n_outer, lcpc_transcript* trw, uint8_t** proof,
                            uint64_t* proof_len, uint64_t* cols_opened) {
  if (!m) return LCPC_ERR_ARG;
  lcpc_ctx* c = m->enc;
  if (!c->comm || !m->committed) return LCPC_ERR_STATE;
  LCPC_TRY
  const uint64_t nb = lcpc_prove_sharded_bytes(c, m->n_rows);
  {
    std::lock_guard<std::mutex> g(m->mu);
    HIPCHK(m, hipSetDevice(c->prm.device));
    if (nb > m->xchg_cap || !m->d_xsend) {
      dev_free(m->d_xsend); dev_free(m->d_xrecv);
      m->d_xsend = m->d_xrecv = nullptr; m->xchg_cap = 0;
      int rc = dev_alloc(&m->err, &m->d_xsend, (size_t)nb);
      if (!rc) rc = dev_alloc(&m->err, &m->d_xrecv, (size_t)nb * std::max<uint32_t>(1, c->prm.shard_count));
      if (rc) return rc;
      m->xchg_cap = nb;
    }
  }
  {
    std::lock_guard<std::mutex> g(m->mu);
    hipStream_t st;
    int rc = prove_stream(m, &st);              // the callback enqueues on m->s_prove: it must exist before the first exchange
    if (rc) return rc;
  }
  ShardXchg x{m->d_xsend, m->d_xrecv, m->xchg_cap, rccl_allgather_cb, m};
  x.stream_ordered = true;
  return prove_impl(m, outer, n_outer, trw, proof, proof_len, cols_opened, &x);
  LCPC_CATCH(m)
}

}  // extern "C"
