// lcpc_amd/csrc/internal.h -- shared internals of the C ABI implementation (include/lcpc_hip.h).
//
//   ctx.cpp     lcpc_ctx     = an LcEncoding implementor (ligero lib.rs:31-186, brakedown lib.rs:41-176): twiddle tables /
//                              expander matrices on the device, dims, batched encode.  Immutable after creation, shared
//                              by any number of commitments (the reference's `&E`, lcpc-2d lib.rs:74-104).
//   commit.cpp  lcpc_commit  = an LcCommit<D, E> (lcpc-2d lib.rs:172-184): comm / coeffs / hashes in HBM, created by
//                              commit(), consumed by prove / open_column / collapse_columns.
//   prove.cpp   transcript wrappers, prove (lib.rs:1004-1093), verify (lib.rs:832-1000), bincode (lib.rs:550-609)
//   shard.cpp   row-sharded commit / prove across GPUs and the RCCL exchange
#pragma once
#include "../../include/lcpc_hip.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "encoding.h"
#include "host_crypto.h"
#include "host_field.h"
#include "host_par.h"
#include "kernels.h"

namespace lcpc {

struct DevCsr {
  uint64_t n_in = 0, n_out = 0;
  uint32_t *rowptr = nullptr, *colidx = nullptr, *vals = nullptr;
  uint32_t* vals29 = nullptr;     // Ft255: values in the 29-bit-limb / 2^261 form (lazy29_mac)
};
struct Pass { uint32_t t0, s, log_tj; int log_tile; };

// device working buffers of one Brakedown encode: owned by a commitment (where the position-major copy IS the
// commitment matrix) or, for lcpc_encode_rows, by the encoder context
struct EncodeWs {
  uint32_t* d_tmp = nullptr;       // last precode output, n_rows x m_last
  uint64_t tmp_cap = 0;            // bytes
  uint32_t* d_t = nullptr;         // position-major working copy T[pos][row] of the rows being encoded
  uint64_t t_cap = 0;              // bytes
  uint32_t* d_mid = nullptr;       // Ligero, Ft255 two-pass plans: the 29-bit-limb intermediate between the passes (ntt_l9s.hip),
  uint64_t mid_cap = 0;            // rows_per_batch x n_cols x 36 bytes (bytes)
  bool mid_failed = false;         // the allocation failed once: stay on the packed intermediate (comm itself)
};
// rows whose limb intermediate fits LCPC_NTT_MID_MAX_MB (default 6144 MiB: the whole headline commitment in one batch; 0 = off)
uint64_t ntt_mid_rows(const lcpc_ctx* c, uint64_t n_rows);

// Brakedown: from this many rows on the rows are encoded on a position-major copy (lane = row); below, row-major with
// lanes over outputs and terms
constexpr uint64_t SDIG_T_MIN_ROWS = 24;

// proof buffers (prove.cpp): lcpc_free hands them back; one is kept for the next proof
void* proof_buf_alloc(size_t n);
void proof_buf_free(void* p);

}  // namespace lcpc

struct lcpc_transcript {
  lcpc::Transcript t;
  lcpc_transcript(const uint8_t* l, size_t n) : t(l, n) {}
};

struct lcpc_ctx {
  lcpc_params prm{};
  // A/B switches, read from the environment ONCE, when the context is created (never on a launch path: getenv next to a
  // setenv of another thread is undefined behaviour, and a context must not change plans under a running commit)
  bool sw_ntt_general = false;     // LCPC_NTT_GENERAL: every Ligero row on the general kernel (K1) instead of the shape-specialised plans
  int64_t sw_ntt_mid_max_mb = -1;  // LCPC_NTT_MID_MAX_MB: -1 = the default rule of ntt_mid_rows
  bool sw_debug_timing = false;    // LCPC_DEBUG_TIMING: phase times of construction / prove / verify on stderr
#ifdef LCPC_TEST_HOOKS             // forced allocation failures, compiled only into lib/liblcpc_hip_testhooks.so (the tests' second build of ctx.cpp)
  bool sw_test_fail_3pass = false; // LCPC_TEST_FAIL=3pass: the three-pass plan's tables "do not fit" -> the general kernel's plan
  bool sw_test_fail_mid = false;   // LCPC_TEST_FAIL=mid: the K1s limb-intermediate allocation fails -> packed intermediate
#endif
  const lcpc::FieldDesc* f = nullptr;
  int L = 0, NL = 0;
  uint64_t n_per_row = 0, n_cols = 0, np2 = 0;
  uint32_t path_len = 0;
  // Ligero
  unsigned log_n = 0;
  uint32_t* d_roots = nullptr;
  uint32_t* d_roots29 = nullptr;   // Ft255: twiddles in radix-2^29 / R'=2^261 Montgomery form (field_dev.h fe_mul_r29)
  uint32_t* d_qp29 = nullptr;      // Ft255: q*p as 29-bit limbs (l9::clamp); null = packed-form NTT kernel
  uint32_t* d_roots29c = nullptr;  // Ft255 lazy-limb kernel: w^i * 2^5, the table that converts to canonical on the fly
                                   // (Brakedown: the position-major commitment of a commit -- ws.d_t, >= SDIG_T_MIN_ROWS rows -- always holds
                                   // canonical values: converted once in the input transpose, kept by every (linear) level)
  bool comm_canon = false;         // d_comm of a commit holds canonical values (x * R^-1), not Montgomery form: the column
                                   // hash reads them as they are; every read-out (get_comm, open_columns) converts back
  std::vector<lcpc::Pass> passes;
  uint32_t* d_pack[3] = {nullptr, nullptr, nullptr};   // Ft255 two- / three-pass plans: lane-order twiddle packs of the specialised kernel (ntt_l9s.hip)
  lcpc::NttPackInfo pack_info[3]{};
  bool l9s = false;
  uint32_t* d_wq_w = nullptr;      // Ft255: the shifted multiples of the primitive 4th root w^(n/4) (the same element for every n), 96 words
  bool l9s3 = false;               // 2^21 .. 2^26 columns: first-pass kernel over the whole rows, then the 2^20-point two-pass plan per block
  uint32_t* d_roots29s = nullptr;  // l9s3: the 2^20-point twiddle tables (every 2^(log_n - 20)-th entry of d_roots29 / d_roots29c)
  uint32_t* d_roots29cs = nullptr;
  // Ft63 / Ft127 / Ft191 two-pass plans: the lazy-limb kernel of ntt_lns.hip (packs in d_pack / pack_info as well)
  bool lns = false;
  bool lns3 = false;               // the same fields at 2^21 .. 2^26 columns: three passes (sub-sampled tables in d_rootsls / d_rootslcs)
  uint32_t* d_rootsls = nullptr;
  uint32_t* d_rootslcs = nullptr;
  uint32_t* d_rootsl = nullptr;    // w^i * R' mod p as N limbs of W bits (field_ln.h), ntt_lns_stride words per entry
  uint32_t* d_rootslc = nullptr;   // w^i * R' R^-1 mod p: the table that converts to canonical on the fly (canonical-output commits)
  uint32_t* d_qpl = nullptr;       // (i - 24) * p, i < 64, same form (ln::clamp_*)
  // Brakedown
  lcpc::SdigSpec spec{};
  std::vector<lcpc::LevelDims> pre_dims, post_dims;
  std::vector<lcpc::DevCsr> d_pre, d_post;
  uint32_t* d_r2 = nullptr;
  // lcpc_encode_rows (the verifier's row encodes): scratch under `mu`
  lcpc::EncodeWs ws;
  uint32_t* d_scratch = nullptr;
  uint64_t scratch_cap = 0;
  // RCCL communicator of a sharded encoder (lcpc_comm_init); opaque ncclComm_t
  void* comm = nullptr;
  std::mutex xchg_mu;              // serialises the submission of collectives on `comm` (several commitments, several host threads)
  hipEvent_t ev_xchg = nullptr;    // recorded behind the last collective enqueued on `comm`; the next one's stream waits for it: the
                                   // collectives of one communicator run in submission order whatever streams they are enqueued on
                                   // (two commitments' exchange streams, a prove stream) -- under xchg_mu
  std::atomic<int> refs{1};        // the handle itself + one per live lcpc_commit
  std::string err;
  std::mutex mu;
  // lcpc_verify's host working set (encoded rows as they come back, to_repr of the polynomials): pinned, kept between
  // calls (a fresh 150 MB of pageable memory costs ~20 ms in first-touch faults and munmap at Brakedown 2^27)
  std::mutex verify_mu;            // held for a whole lcpc_verify call
  uint8_t* h_varena = nullptr;
  size_t h_varena_cap = 0;
  // lcpc_commit from PAGEABLE host memory (commit.cpp upload_host): a ring of pinned bounce buffers the host pool fills while
  // the previous slices cross the bus.  stage_mu is held per upload call (one row batch of one commit); the buffers are kept between commits
  // (up to 4 x 64 MiB of pinned host memory per encoder that has taken a pageable source; LCPC_HOST_STAGE=1 pins the first two at lcpc_ctx_create).
  std::mutex stage_mu;
  static constexpr unsigned N_STAGE = 4;
  uint8_t* h_stage[N_STAGE] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_stage[N_STAGE] = {nullptr, nullptr, nullptr, nullptr};   // the H2D copy that last read h_stage[k]
  size_t stage_cap = 0;            // bytes per buffer
  unsigned stage_next = 0;
  int32_t sw_host_stage = -1;      // LCPC_HOST_STAGE: 0 = never stage (the runtime's own pageable path), 1 = always stage, -1 (unset) = by pointer attributes
};

struct lcpc_commit_s {
  lcpc_ctx* enc = nullptr;
  bool committed = false;
  uint64_t n_rows = 0;             // rows of the whole commitment
  uint64_t row_begin = 0, n_rows_local = 0;
  uint64_t chunk_begin = 0, chunk_end = 0, n_chunks = 0;
  uint32_t *d_coeffs = nullptr, *d_comm = nullptr, *d_hashes = nullptr, *d_cvs = nullptr;
  const uint32_t* coeffs_view = nullptr;   // LcCommit.coeffs as prove/collapse read it: d_coeffs, or the caller's buffer
                                           // when the commit was made with LCPC_COMMIT_BORROW_COEFFS
  uint64_t cap_coeff_rows = 0, cap_comm_rows = 0, cap_cvs = 0;
  lcpc::EncodeWs ws;
  bool comm_t = false;             // Brakedown commit with >= sdig_t_min_rows() local rows: the commitment matrix lives in ws.d_t (position-major,
                                   // element (row, col) at (col * n_rows_local + row)); hash / open read it there, d_comm is only
                                   // filled on demand (lcpc_get_comm) -- no back-transpose on the commit path
  bool comm_rows_valid = false;    // d_comm holds the row-major copy of the commitment in ws.d_t
  uint32_t* d_node_tab = nullptr;  // sharded finish: node_slot[0..n) then node_log[0..n) -- the current shape's entry of node_tabs
  uint64_t node_tab_key = 0;
  std::vector<uint32_t> node_slot_h, node_log_h;
  struct NodeTab { uint64_t key; uint32_t* d; std::vector<uint32_t> slot, lg; };
  std::vector<NodeTab> node_tabs;  // one device table per shape seen, freed with the object (never while a finish step may read it)
  uint8_t* d_gather = nullptr;     // native sharded commit (lcpc_commit_sharded_device): this rank's nodes + the all-gather output
  uint64_t gather_cap = 0;
  uint8_t *d_xsend = nullptr, *d_xrecv = nullptr;   // native sharded prove: exchange buffers
  uint64_t xchg_cap = 0;
  // scratch for prove / collapse / open
  uint32_t* d_scratch = nullptr;
  uint64_t scratch_cap = 0;
  uint32_t* d_t29 = nullptr;       // collapse: tensors in the 29-bit-limb form
  uint64_t t29_cap = 0;
  uint32_t* h_root = nullptr;      // pinned, device-mapped: the Merkle kernel that produces the root writes it here as well
  uint32_t* d_root_alias = nullptr;  // ... through this device address (null: no mapping, the root is copied out)
  uint8_t* h_pin = nullptr;        // pinned host arena of prove (tensors, polynomials, their canonical forms)
  uint64_t h_pin_cap = 0;
  // timing
  bool timing = false;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // start | encoded | hashed | done; [4]: the commit's stream
                                   // has the leaf digests (sharded commit: behind the exchange); [5]: the exchange's collectives are done
  hipStream_t s_prove = nullptr;   // sharded prove: its device steps and the native exchange, ordered behind the commit by ev_done
  hipEvent_t ev_done = nullptr;    // recorded on the commit's stream when a sharded commit has been enqueued completely
  // native sharded commit (shard.cpp): the exchange stream of an async tail and its hand-over event
  hipStream_t s_xchg = nullptr;
  hipEvent_t ev_hashed = nullptr;  // recorded on the caller's stream behind the local column hash; s_xchg waits for it
  bool shard_encoded = false;      // split phases: the encode step of a sharded commit has been enqueued, hash / finish / merkle may follow
  hipStream_t s_copy = nullptr, s_comp = nullptr;   // lcpc_commit (host pointer): H2D of row batch b+1 overlaps the NTTs of batch b
  hipEvent_t ev_batch[16] = {nullptr};
  std::mutex prove_mu;             // held for a whole prove on this commitment: the pinned arena, the slice events and the scratch layout are
                                   // per object (LcCommit::prove takes &self: concurrent proves on ONE commitment queue up, on different ones run side by side)
  hipEvent_t ev_slice[2] = {nullptr, nullptr};   // prove: arrival of the two column ranges of p_random on the host (collapse_host_sliced)
  lcpc_timings last{};
  uint32_t launches[3] = {0, 0, 0};
  std::string err;
  std::mutex mu;
};

namespace lcpc {

// ---- error plumbing ---------------------------------------------------------------------------------
inline int fail_hip(std::string* err, hipError_t e, const char* what) {
  if (err) *err = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? LCPC_ERR_NOMEM : LCPC_ERR_HIP;
}
// `c` is an lcpc_ctx* or lcpc_commit_t* (both have .err)
#define HIPCHK(c, call)                                                   \
  do {                                                                    \
    hipError_t e__ = (call);                                              \
    if (e__ != hipSuccess) return lcpc::fail_hip((c) ? &(c)->err : nullptr, e__, #call); \
  } while (0)

// nothing may unwind through the C ABI (include/lcpc_hip.h): every extern "C" body that can allocate runs inside this
#define LCPC_TRY try {
#define LCPC_CATCH(c)                                                     \
  } catch (const std::bad_alloc&) {                                       \
    if (c) (c)->err = "host allocation failed";                          \
    return LCPC_ERR_NOMEM;                                                \
  } catch (const std::exception& ex__) {                                  \
    if (c) (c)->err = ex__.what();                                       \
    return LCPC_ERR_STATE;                                                \
  } catch (...) {                                                         \
    return LCPC_ERR_STATE;                                                \
  }

template <typename T> int dev_alloc(std::string* err, T** p, size_t bytes) {
  *p = nullptr;
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), bytes);
  if (e != hipSuccess) return fail_hip(err, e, "hipMalloc");
  return 0;
}
inline void dev_free(void* p) { if (p) (void)hipFree(p); }
// grow-only device buffer; the requested size is rounded up to 256 bytes so that offsets computed from the capacity
// (collapse partials at the end of scratch) stay 16-byte aligned for the uint4 element accesses
template <typename T> int ensure_dev(std::string* err, T** p, uint64_t* cap, uint64_t bytes) {
  bytes = (bytes + 255) & ~(uint64_t)255;
  if (bytes > *cap || !*p) {
    dev_free(*p);
    *p = nullptr; *cap = 0;
    int rc = dev_alloc(err, p, (size_t)bytes);
    if (rc) return rc;
    *cap = bytes;
  }
  return 0;
}

inline size_t elem_bytes(const lcpc_ctx* c) { return (size_t)8 * c->L; }
// leaf message = 32 + F * n_rows bytes -> BLAKE3 chunks of 1 KiB
inline uint64_t leaf_chunks(const lcpc_ctx* c, uint64_t n_rows) { return (32 + elem_bytes(c) * n_rows + 1023) / 1024; }

// labels of the transcript (macros.rs:31-34)
extern const uint8_t LBL_DT[7], LBL_PR[7], LBL_PE[7], LBL_CO[7];


// ---- ctx.cpp ----------------------------------------------------------------------------------------
void ctx_ref(lcpc_ctx* c);
void ctx_unref(lcpc_ctx* c);
// encode n_rows rows: src (src_stride elements per row, first n_valid valid, flat elements >= n_src_total zero) -> dst
// (n_cols per row).  err: where HIP error text goes; launches: encode launch counter or null.
struct EncodeJob {
  const uint32_t* src = nullptr;
  uint64_t src_stride = 0, n_valid = 0;
  uint32_t* dst = nullptr;
  uint64_t n_rows = 0;
  uint64_t n_src_total = ~(uint64_t)0;
  uint32_t* copy_dst = nullptr;    // padded LcCommit.coeffs copy written while the source streams through (or null)
  bool canon_out = false;          // dst receives canonical values instead of Montgomery form (commit paths of a comm_canon context)
  bool keep_t = false;             // Brakedown: leave the result position-major in ws->d_t (the commit path); *kept_t reports it
  bool* kept_t = nullptr;
};
int encode_rows_device(const lcpc_ctx* c, EncodeWs* ws, const EncodeJob& j, hipStream_t st, std::string* err, uint32_t* launches);

int encode_msgs_host(lcpc_ctx* c, const uint64_t* const* msgs, uint64_t n_rows, uint64_t* out);

// ---- commit.cpp -------------------------------------------------------------------------------------
int ensure_scratch(lcpc_commit_t* m, uint64_t bytes);
int ensure_cvs(lcpc_commit_t* m, uint64_t n_chunks);
int ensure_commit_buffers(lcpc_commit_t* m, uint64_t n_rows_local, bool own_coeffs);
int merkle_top(lcpc_commit_t* m, hipStream_t st, uint32_t levels_done = 0);       // zero padding leaves + tree above the leaf digests (above level `levels_done`)
int order_after_commit(lcpc_commit_t* m, hipStream_t st);          // st waits for the commit that filled m (event; cheap)
int fetch_root(lcpc_commit_t* m, hipStream_t st, uint8_t* root);   // root of the commit just enqueued on st -> host (synchronises)
int finish_timing(lcpc_commit_t* m, hipStream_t st);
int collapse_run(lcpc_commit_t* m, const uint32_t* d_tensors, uint32_t n_tensors, hipStream_t st, uint32_t* d_polys);
size_t collapse_scratch_bytes(const lcpc_commit_t* m, uint32_t n_tensors);
int collapse_host(lcpc_commit_t* m, const uint64_t* tensors, uint32_t n_tensors, uint64_t* polys, uint64_t* polys_canon);
int collapse_host_sliced(lcpc_commit_t* m, const uint64_t* tensor, uint64_t* polys, uint64_t* polys_canon, uint64_t* cut_out);
int collapse_wait_slice(lcpc_commit_t* m, int s);
// open_column values / paths into device buffers (either may be null)
int open_columns_host(lcpc_commit_t* m, const uint64_t* cols, uint32_t n, uint64_t* col_vals, size_t vals_pitch, uint8_t* paths);
int ensure_pinned(lcpc_commit_t* m, uint64_t bytes);
int open_columns_device(lcpc_commit_t* m, const uint64_t* d_cols, uint32_t n, uint32_t* d_vals, uint32_t* d_paths, hipStream_t st);

// ---- shard.cpp --------------------------------------------------------------------------------------
void shard_layout_of(const lcpc_ctx* c, uint64_t g, uint64_t n_rows, uint64_t* rb, uint64_t* re, uint64_t* cb, uint64_t* ce, uint64_t* nch);
int shard_nodes(uint64_t c0, uint64_t c1, uint64_t* first, uint32_t* lg);
void shard_chunk_range(uint64_t F, uint64_t n_chunks, uint64_t G, uint64_t g, uint64_t* c0, uint64_t* c1);
void comm_release(lcpc_ctx* c);
// the exchange of a row-sharded prove: every rank contributes `bytes` from send_dev, receives all ranks'
// blocks in rank order in recv_dev
struct ShardXchg {
  uint8_t *send_dev, *recv_dev;
  uint64_t max_bytes;
  lcpc_allgather_fn fn;
  void* user;
  bool stream_ordered = false;     // fn only ENQUEUES the all-gather on the commitment's prove stream (the native RCCL exchange):
                                   // no host synchronisation around it; a caller-supplied fn (torch.distributed, MPI) is host-driven
};
// every device step of a sharded prove runs on the commitment's own stream (prove_stream), ordered behind the commit by an event;
// polys_canon (optional): to_repr of the polynomials, converted on the device; vals_pitch: bytes between the values of
// consecutive opened columns in `vals` (0 = packed)
int prove_stream(lcpc_commit_t* m, hipStream_t* st);
int collapse_sharded(lcpc_commit_t* m, const ShardXchg& x, const uint64_t* tensors_full, uint32_t nt, uint64_t* polys, uint64_t* polys_canon);
int open_sharded(lcpc_commit_t* m, const ShardXchg& x, const uint64_t* cols, uint32_t n, uint64_t* vals, size_t vals_pitch, uint8_t* paths);

// ---- prove.cpp --------------------------------------------------------------------------------------
int prove_impl(lcpc_commit_t* m, const uint64_t* outer, uint64_t n_outer, lcpc_transcript* trw, uint8_t** proof, uint64_t* proof_len,
               uint64_t* cols_opened, const ShardXchg* xchg);

}  // namespace lcpc
