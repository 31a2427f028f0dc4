/*
 * include/lcpc_hip.h -- C ABI of the MI355X-native lcpc-2d commit / prove path.
 *
 * Drop-in boundary for conroi/lcpc (reference @ /root/reference; file:line below cite that tree).
 * The reference's plugin API for this path is the LcEncoding trait (lcpc-2d/src/lib.rs:74-104)
 * plus LcCommit::{commit, prove, get_root} (lib.rs:270-312) and LcEvalProof::verify (lib.rs:518-527).
 * Per-row `encode` cannot be the FFI seam (it is called once per row from inside a Rayon closure,
 * lib.rs:648-653), so the seam is one level up: whole-matrix operations on a device-resident
 * commitment, with a batched single-row `encode` kept for the verifier.
 *
 * Conventions
 *  - every function returns 0 on success or a negative lcpc_status; nothing throws across the ABI;
 *  - field elements cross exactly as ff_derive stores them: L little-endian uint64_t limbs in
 *    Montgomery form (R = 2^(64 L)), so a Rust `&[Ft255]` is passed as `*const u64` unchanged
 *    (lcpc-test-fields/src/lib.rs:18-58); digests are 32 raw bytes;
 *  - two handle types, as in the reference: `lcpc_ctx` is an LcEncoding implementor (`&E`: twiddle tables /
 *    expander matrices on the device, immutable after creation, shared by any number of commitments and
 *    usable from several host threads at once, like a `Sync` encoder), and `lcpc_commit_t` is an
 *    LcCommit<D, E> (lib.rs:172-184): comm / coeffs / hashes of ONE commitment, resident in HBM, created by
 *    commit() and consumed by prove / open_column / collapse_columns.  Many commitments may be live under
 *    one encoder (lib.rs:299-311); each is used by one host thread at a time;
 *  - the caller owns every host buffer; device memory belongs to the handle that allocated it;
 *  - `*_device` entry points take HIP device pointers + a hipStream_t (passed as void*), so a host
 *    runtime (torch, or a Rust hip-sys binding) can keep inputs resident in HBM.
 */
#ifndef LCPC_HIP_H
#define LCPC_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 5: the four column-range phases of ABI 3 (lcpc_commit_shard_encode_device, lcpc_commit_shard_hash_device,
 * lcpc_commit_finish_cols_device, lcpc_commit_finish_merkle_device) are gone -- slicing the exchange was measured neutral to
 * negative (profiles/r04_shard_slices.jsonl) and nothing called them; lcpc_timings.staged_slices / .exchange_wire_ms; lcpc_random_coeffs_device, lcpc_shard_exchange_probe, lcpc_comm_rccl_version.
 * 4: lcpc_shard_nodes_field (row sharding for Ft191, whose elements straddle BLAKE3 chunks), LCPC_COMMIT_ASYNC_TAIL.
 * 3: the column-range phases of the sharded commit (lcpc_commit_shard_encode_device, lcpc_commit_shard_hash_device,
 * lcpc_commit_finish_cols_device, lcpc_commit_finish_merkle_device), lcpc_timings.exchange_exposed_ms; the LcCommit bincode
 * entry points of round 3.  A caller checks lcpc_abi_version() == LCPC_ABI_VERSION before anything else. */
#define LCPC_ABI_VERSION 5

/* fields of lcpc-test-fields/src/lib.rs:13-59 */
enum { LCPC_FT63 = 0, LCPC_FT127 = 1, LCPC_FT191 = 2, LCPC_FT255 = 3 };
/* encodings: lcpc-ligero-pc/src/lib.rs:31-37 (LigeroEncodingRho), lcpc-brakedown-pc/src/lib.rs:41-47 (SdigEncodingS) */
enum { LCPC_ENC_LIGERO = 0, LCPC_ENC_SDIG = 1 };
/* D: Digest -- every reference test/bench uses blake3::Hasher */
enum { LCPC_HASH_BLAKE3 = 0 };

typedef enum {
  LCPC_OK = 0,
  /* ProverError, lcpc-2d/src/lib.rs:111-131 */
  LCPC_ERR_TOO_BIG = -1,
  LCPC_ERR_ENCODE = -2,
  LCPC_ERR_COMMIT = -3,
  LCPC_ERR_COLUMN_NUMBER = -4,
  LCPC_ERR_OUTER_TENSOR = -5,
  /* the reference's assert!()s on caller-shaped input (lib.rs:630-632, ligero lib.rs:139) */
  LCPC_ERR_DIMS = -6,
  LCPC_ERR_ARG = -7,
  LCPC_ERR_STATE = -8,      /* e.g. prove before commit */
  /* device / runtime */
  LCPC_ERR_HIP = -16,
  LCPC_ERR_NOMEM = -17,
  LCPC_ERR_NO_DEVICE = -18,
  LCPC_ERR_XCHG = -19,      /* the all-gather of a sharded commit / prove failed (callback or RCCL) */
  LCPC_ERR_NO_RCCL = -20,   /* librccl could not be loaded (native exchange only) */
  /* VerifierError, lib.rs:137-166 */
  LCPC_VERR_NUM_COL_OPENS = -32,
  LCPC_VERR_COLUMN_PATH = -33,
  LCPC_VERR_COLUMN_EVAL = -34,
  LCPC_VERR_COLUMN_DEGREE = -35,
  LCPC_VERR_OUTER_TENSOR = -36,
  LCPC_VERR_INNER_TENSOR = -37,
  LCPC_VERR_ENCODING_DIMS = -38,
  LCPC_VERR_ENCODE = -39,
  LCPC_VERR_MALFORMED = -40 /* bincode payload does not parse (serde error in the reference) */
} lcpc_status;

typedef struct lcpc_ctx lcpc_ctx;            /* an LcEncoding implementor (`enc`) */
typedef struct lcpc_commit_s lcpc_commit_t;    /* an LcCommit<D, E> */
typedef struct lcpc_transcript lcpc_transcript;

typedef struct {
  uint32_t field;        /* LCPC_FT* */
  uint32_t encoding;     /* LCPC_ENC_* */
  uint32_t hash;         /* LCPC_HASH_BLAKE3 */
  uint32_t rho_num, rho_den;  /* Ligero rate Rn/Rd (default alias 1/2: ligero lib.rs:189) */
  uint32_t sdig_code;    /* 1..6 = SdigCode1..6 (codespec.rs:169-232); default 3 (brakedown lib.rs:19) */
  uint64_t seed;         /* Brakedown matgen seed (brakedown lib.rs:103) */
  uint64_t n_coeffs;     /* `len`: dims come from the restated `new(len)` optimisers ... */
  uint64_t n_per_row;    /* ... unless n_per_row and n_cols are both nonzero: `new_from_dims` */
  uint64_t n_cols;
  int32_t  device;       /* HIP device ordinal */
  /* row sharding across GPUs (one ctx/process per GPU): this ctx holds shard `shard_rank` of
   * `shard_count`; 0/0 or 0/1 = unsharded.  Shards are aligned to BLAKE3 chunk boundaries of the
   * leaf message so that each GPU reduces its rows to chunk chaining values. */
  uint32_t shard_rank, shard_count;
} lcpc_params;

/* ---- construction: LigeroEncoding::new / new_from_dims (ligero lib.rs:121-148),
 *      SdigEncoding::new / new_from_dims (brakedown lib.rs:103-137).  Twiddles / expander matrices
 *      are built and uploaded here, outside any timed commit (as in rough_bench, ligero tests.rs:86-90). */
int  lcpc_ctx_create(const lcpc_params *params, lcpc_ctx **out);
/* commitments created from the context keep its tables alive until they are destroyed themselves */
void lcpc_ctx_destroy(lcpc_ctx *ctx);
const char *lcpc_strerror(int status);
const char *lcpc_last_error(const lcpc_ctx *ctx);     /* detail string for LCPC_ERR_HIP etc. */
int  lcpc_abi_version(void);

/* ---- LcEncoding trait (lcpc-2d/src/lib.rs:74-104) ---- */
/* get_dims(len) (ligero lib.rs:166-169, brakedown lib.rs:155-158) */
int  lcpc_get_dims(const lcpc_ctx *ctx, uint64_t len, uint64_t *n_rows, uint64_t *n_per_row, uint64_t *n_cols);
/* dims_ok (ligero lib.rs:171-177, brakedown lib.rs:160-167): returns 1 / 0 */
int  lcpc_dims_ok(const lcpc_ctx *ctx, uint64_t n_per_row, uint64_t n_cols);
uint64_t lcpc_get_n_col_opens(const lcpc_ctx *ctx);       /* ligero lib.rs:179-181 */
uint64_t lcpc_get_n_degree_tests(const lcpc_ctx *ctx);    /* ligero lib.rs:183-185 */
uint32_t lcpc_field_limbs(const lcpc_ctx *ctx);           /* L */
/* static `_get_dims(len)` without a context (ligero lib.rs:70-112; brakedown lib.rs:69-110) */
int  lcpc_static_get_dims(const lcpc_params *params, uint64_t *n_rows, uint64_t *n_per_row, uint64_t *n_cols);
/* new_ml(n_vars) (ligero lib.rs:128-135; brakedown lib.rs:114-123): the dims the reference picks for a multilinear
 * polynomial with 2^n_vars monomials; build the context with new_from_dims semantics (n_per_row, n_cols) from them,
 * as the reference does.  LCPC_ERR_DIMS where the reference's assert!s fire (ligero: non-power-of-two split). */
int  lcpc_static_get_dims_ml(const lcpc_params *params, uint32_t n_vars, uint64_t *n_rows, uint64_t *n_per_row, uint64_t *n_cols);
/* encode (ligero lib.rs:162-164, brakedown lib.rs:150-153), batched: `rows` holds n_rows rows of
 * n_cols elements each, first n_per_row = message, rest zero on entry; encoded in place. */
int  lcpc_encode_rows(lcpc_ctx *ctx, uint64_t *rows_host, uint64_t n_rows);
/* lcpc_test_fields::random_coeffs (lcpc-test-fields/src/lib.rs:75-97) with a FIXED generator, on the device: n elements of the
 * encoder's field by ff's `Field::random` rule (L x next_u64 as limbs, top limb masked to NUM_BITS, accepted iff < p; the limbs
 * are the Montgomery representation) drawn from ChaCha20Rng::from_seed(seed) after set_stream(stream_id) -- element for element
 * the vector the same rule gives on a host (SURVEY.md 8d "Synthetic inputs": seed 32 x 0x00, stream 0).  out_dev: n * L u64 on
 * the encoder's device.  Synchronises `stream` (a utility for tests and benches, not a commit-path call). */
int  lcpc_random_coeffs_device(lcpc_ctx *ctx, const uint8_t seed[32], uint64_t stream_id, uint64_t n, uint64_t *out_dev, void *stream);

/* ---- LcCommit (lcpc-2d/src/lib.rs:172-184, 270-312) ---- */
/* An empty LcCommit bound to `enc` (no device memory yet); every commit entry point below fills it.  Filling it
 * again replaces the commitment and reuses the buffers (a benchmark loop commits into one object; the reference
 * would drop and reallocate its Vecs).  The object holds a reference on `enc`.
 * Refilling across streams: a fill that only enqueues (lcpc_commit_device and the sharded entry points with a NULL
 * `root`) records an event; the next fill -- on any stream, or from host memory -- and every reader of the library
 * (prove, collapse, open, the getters) is ordered behind that event by the library.  What the library cannot see is
 * the caller's own device-side reader: work enqueued with lcpc_collapse_device on stream A must have completed, or be
 * ordered by the caller, before the object is refilled on another stream B. */
int  lcpc_commit_create(lcpc_ctx *enc, lcpc_commit_t **out);
void lcpc_commit_destroy(lcpc_commit_t *cm);
const char *lcpc_commit_last_error(const lcpc_commit_t *cm);
/* commit(coeffs, enc) (lib.rs:299-301 -> 622-671): pad, encode every row, hash columns, Merkleize.
 * comm / coeffs / hashes stay on the device.  `root` (32 bytes) may be NULL.  Returns after the work is complete. */
int  lcpc_commit(lcpc_commit_t *cm, const uint64_t *coeffs_host, uint64_t n_coeffs, uint8_t *root);
/* same with the coefficients already resident in HBM (device pointer), work enqueued on `stream`;
 * if `root` is non-NULL the call synchronises the stream and copies the root out.
 * flags & LCPC_COMMIT_BORROW_COEFFS: LcCommit.coeffs (lib.rs:636-645) is not copied -- the commitment keeps reading the
 * caller's buffer (prove / collapse), which must stay valid and unchanged until the commitment is replaced or
 * destroyed.  Honoured when n_coeffs fills whole rows (n_coeffs == n_rows * n_per_row); a ragged vector is copied
 * (the padded tail has to exist somewhere). */
enum {
  LCPC_COMMIT_BORROW_COEFFS = 1,
  /* lcpc_commit_sharded_device only: the exchange, the leaf digests and the Merkle tree run on the commitment's own stream and
   * `stream` does not wait for them -- it is free again after the local column hash, so the next commit on `stream` (into
   * ANOTHER lcpc_commit_t; a refill of the same one waits for it) encodes while this one's node values are on the wire.  The
   * commitment is complete behind its event, which every reader of the library waits for; `root` != NULL still synchronises. */
  LCPC_COMMIT_ASYNC_TAIL = 2
};
int  lcpc_commit_device(lcpc_commit_t *cm, const uint64_t *coeffs_dev, uint64_t n_coeffs, void *stream, uint32_t flags,
                        uint8_t *root);
/* test hook for lcpc-2d/src/tests.rs:435-466 `random_comm` + `merkleize` (tests.rs:136-149): install a
 * caller-supplied comm (n_rows x n_cols) and coeffs (n_rows x n_per_row, may be NULL), then Merkleize. */
int  lcpc_commit_from_parts(lcpc_commit_t *cm, const uint64_t *comm_host, const uint64_t *coeffs_host,
                            uint64_t n_rows, uint8_t *root);
/* ---- serde of LcCommit itself (lcpc-2d/src/lib.rs:186-268: `Serialize` / `Deserialize for LcCommit` through
 * WrappedLcCommit { comm, coeffs, n_rows, n_cols, n_per_row, hashes }) in bincode 1.3's default layout -- the only way to
 * hand a COMMITMENT to or from the reference (its fields are private; GPU commit -> reference prove, or a reference
 * commitment -> GPU prove):
 *   u64 len(comm) | comm: len x L u64 limbs (Montgomery form, row-major) | u64 len(coeffs) | coeffs | u64 n_rows | u64 n_cols |
 *   u64 n_per_row | u64 len(hashes) | len x (u64 32 | 32 digest bytes)
 * Streaming: 6 GiB at the headline never sits in one host buffer.  `write` receives consecutive pieces (<= 64 MiB each),
 * `read` must fill exactly `len` bytes; either returns 0 to go on, anything else aborts the call (LCPC_ERR_ARG).
 * lcpc_commit_from_bincode checks what check_comm (lib.rs:673-688) checks -- lengths against the dims and the encoder's
 * dims_ok -- refuses limbs >= p (LCPC_ERR_COMMIT), and rebuilds the Merkle tree from `comm` on the device: digests that
 * differ from the stream's `hashes` are refused too (the reference would carry them along and prove against a root nobody
 * can verify).  Unsharded commitments only (LCPC_ERR_STATE otherwise). */
typedef int (*lcpc_write_fn)(void *user, const uint8_t *data, uint64_t len);
typedef int (*lcpc_read_fn)(void *user, uint8_t *data, uint64_t len);
uint64_t lcpc_commit_bincode_size(const lcpc_commit_t *cm);             /* 0: nothing committed */
int  lcpc_commit_bincode_write(lcpc_commit_t *cm, lcpc_write_fn write, void *user);
int  lcpc_commit_from_bincode(lcpc_commit_t *cm, lcpc_read_fn read, void *user, uint8_t *root);
int  lcpc_get_root(lcpc_commit_t *cm, uint8_t root[32]);                /* get_root lib.rs:276-281 */
int  lcpc_commit_dims(const lcpc_commit_t *cm, uint64_t *n_rows, uint64_t *n_per_row, uint64_t *n_cols,
                      uint64_t *n_hashes);                               /* get_n_rows/.. lib.rs:283-296 */
int  lcpc_get_hashes(lcpc_commit_t *cm, uint8_t *hashes);               /* LcCommit.hashes: (2*np2-1)*32 bytes */
int  lcpc_get_comm(lcpc_commit_t *cm, uint64_t row0, uint64_t n_rows, uint64_t *out);    /* LcCommit.comm rows (Montgomery form,
                                                                                             whatever the device keeps) */
int  lcpc_get_coeffs(lcpc_commit_t *cm, uint64_t row0, uint64_t n_rows, uint64_t *out);  /* LcCommit.coeffs rows */

/* collapse_columns (lib.rs:1095-1123; test alias eval_outer lib.rs:1176-1202) for n_tensors tensors
 * of n_rows elements each, fused into one pass over coeffs: polys[t][j] = sum_r coeffs[r][j]*tensors[t][r]. */
int  lcpc_collapse(lcpc_commit_t *cm, const uint64_t *tensors_host, uint32_t n_tensors, uint64_t *polys_host);
/* open_column (lib.rs:788-825) for n columns: col_vals[n][n_rows][L], paths[n][path_len][32],
 * path_len = ceil(log2 n_cols).  LCPC_ERR_COLUMN_NUMBER if any column >= n_cols. */
int  lcpc_open_columns(lcpc_commit_t *cm, const uint64_t *cols, uint32_t n, uint64_t *col_vals, uint8_t *paths);

/* ---- merlin::Transcript (the `tr: &mut Transcript` argument of prove/verify, lib.rs:304-311, 518-527) ---- */
lcpc_transcript *lcpc_transcript_new(const uint8_t *label, size_t len);
lcpc_transcript *lcpc_transcript_clone(const lcpc_transcript *);
void lcpc_transcript_append_message(lcpc_transcript *, const uint8_t *label, size_t llen, const uint8_t *msg, size_t mlen);
/* n calls of append_message(label, msgs + i * mlen) in one: the shape of prove / verify's per-coefficient absorbs
 * (lib.rs:1045-1047); same transcript state as the loop, with the sponge kept in vector registers in between */
void lcpc_transcript_append_messages(lcpc_transcript *, const uint8_t *label, size_t llen, const uint8_t *msgs, size_t mlen, size_t n);
void lcpc_transcript_challenge_bytes(lcpc_transcript *, const uint8_t *label, size_t llen, uint8_t *out, size_t n);
void lcpc_transcript_free(lcpc_transcript *);

/* ---- prove / verify ---- */
/* LcCommit::prove (lib.rs:304-311 -> 1004-1093); the encoder is the one the commitment was made with.  The proof is returned in the reference's bincode 1.3
 * wire layout (lib.rs:550-609): the only way to hand an LcEvalProof to the reference (its fields are
 * private).  `*proof` is allocated by the library; release it with lcpc_free
 * (which may keep one released buffer for the next proof instead of returning it to the OS).  cols_opened (n_col_opens entries) may be NULL. */
int  lcpc_prove(lcpc_commit_t *cm, const uint64_t *outer_tensor, uint64_t n_outer, lcpc_transcript *tr,
                uint8_t **proof, uint64_t *proof_len, uint64_t *cols_opened);
/* LcEvalProof::verify (lib.rs:518-527 -> 832-952) on a bincode proof; `ctx` plays the role of `enc`
 * (it need not hold a commitment).  eval_out: L limbs.  Bytes after the last column are ignored, as by
 * bincode::deserialize (bincode 1.3's top-level functions allow trailing bytes). */
int  lcpc_verify(lcpc_ctx *ctx, const uint8_t root[32], const uint64_t *outer_tensor, uint64_t n_outer,
                 const uint64_t *inner_tensor, uint64_t n_inner, const uint8_t *proof, uint64_t proof_len,
                 lcpc_transcript *tr, uint64_t *eval_out);
/* bincode of LcRoot (lib.rs:373-384): 8-byte length + 32 bytes -> out[40] */
void lcpc_root_bincode(const uint8_t root[32], uint8_t out[40]);
void lcpc_free(void *p);

/* ---- row-sharded commit across GPUs: one encoder ctx (shard_rank / shard_count set) and one process per
 *      GPU.  Rows are split into blocks aligned to the BLAKE3 chunk boundaries of the leaf message, so a rank reduces its
 *      rows to chaining values alone; the ONE exchange step of the path is an all-gather of those.  Two ways to run it:
 *      (a) native: lcpc_comm_init() gives the encoder an RCCL communicator and lcpc_commit_sharded_device() /
 *          lcpc_prove_sharded_rccl() do shard -> ncclAllGather -> finish on one stream (xGMI, no host round trip);
 *      (b) split phases + a caller-supplied exchange (torch.distributed, MPI, a test harness): lcpc_commit_shard_device,
 *          the caller's all-gather, lcpc_commit_finish_device; lcpc_prove_sharded with an lcpc_allgather_fn. ---- */
/* rows [row_begin,row_end) and leaf-message chunks [chunk_begin,chunk_end) owned by this shard for a
 * commitment of n_rows_total rows; n_chunks_total = BLAKE3 chunks per leaf message. */
int  lcpc_shard_layout(const lcpc_ctx *ctx, uint64_t n_rows_total, uint64_t *row_begin, uint64_t *row_end,
                       uint64_t *chunk_begin, uint64_t *chunk_end, uint64_t *n_chunks_total);
/* The chunks of a shard are pre-merged on the GPU into aligned BLAKE3 subtree nodes (a node = 2^log_size
 * consecutive chunks starting at a multiple of its size), which is what crosses the wire.  Pure function of
 * (field, n_chunks_total, shard_count, shard_rank): node k of that shard starts at chunk first_chunk[k].  A shard begins where
 * a chunk boundary of the leaf message is also a row boundary: for Ft63 / Ft127 / Ft255 (element size divides 1024) that is
 * every chunk and shard g owns chunks [n g / G, n (g + 1) / G); for Ft191 (24-byte elements) it is every third chunk
 * (rows = 84 mod 128) and the even split is moved down to the nearest one.  lcpc_shard_nodes is the first case. */
int  lcpc_shard_nodes(uint64_t n_chunks_total, uint32_t shard_count, uint32_t shard_rank, uint32_t *n_nodes,
                      uint64_t *first_chunk /* [64] */, uint32_t *log_size /* [64] */);
int  lcpc_shard_nodes_field(uint32_t field, uint64_t n_chunks_total, uint32_t shard_count, uint32_t shard_rank, uint32_t *n_nodes,
                            uint64_t *first_chunk /* [64] */, uint32_t *log_size /* [64] */);
/* (a) native exchange.  One rank calls lcpc_comm_unique_id (ncclGetUniqueId) and distributes the 128 bytes by any
 * means; every rank then calls lcpc_comm_init on its sharded encoder (ncclCommInitRank on the encoder's device;
 * rank / world must equal shard_rank / shard_count; world == 1 is allowed and still goes through RCCL).
 * librccl is loaded at run time (dlopen; a copy already in the process wins, LCPC_RCCL_LIB=<path> names one explicitly):
 * LCPC_ERR_NO_RCCL if it is absent. */
int  lcpc_comm_unique_id(uint8_t id[128]);
int  lcpc_comm_init(lcpc_ctx *ctx, const uint8_t id[128], uint32_t rank, uint32_t world);
int  lcpc_comm_destroy(lcpc_ctx *ctx);
/* whole sharded commit: encode the local rows (coeffs_local_dev = this rank's rows, row-major) and hash their columns down to
 * node chaining values on `stream`, ONE exchange of those (ncclAllGather + grouped ncclBroadcasts for the few ranks that own a
 * second node), leaf digests and the Merkle tree (replicated) -- by default all on `stream`, in sequence.  LCPC_COMMIT_ASYNC_TAIL
 * in `flags` takes the wire off the critical path (above: overlap with the NEXT commit's encode; what a prover committing several
 * polynomials wants); the result is bit-identical.
 * No host synchronisation unless `root` is non-NULL.  flags: LCPC_COMMIT_BORROW_COEFFS as for lcpc_commit_device.
 * Collectives on one communicator must be issued in the same order on every rank: drive the sharded commits / proves of
 * one encoder from ONE host thread per rank, in the same program order everywhere (the library only keeps two commitments
 * of one process from interleaving their exchanges). */
int  lcpc_commit_sharded_device(lcpc_commit_t *cm, const uint64_t *coeffs_local_dev, uint64_t n_rows_total, void *stream,
                                uint32_t flags, uint8_t *root);
/* LcCommit::prove on a sharded commitment with the three all-gathers on RCCL (see lcpc_prove_sharded). */
int  lcpc_prove_sharded_rccl(lcpc_commit_t *cm, const uint64_t *outer_tensor, uint64_t n_outer, lcpc_transcript *tr,
                             uint8_t **proof, uint64_t *proof_len, uint64_t *cols_opened);
/* (b) split phases: a caller with its own collective (torch.distributed, MPI, a test harness). */
/* phase 1: encode the local rows (coeffs_dev = local rows only, row-major, padded) and reduce them to one
 * BLAKE3 chaining value per (local node, column): nodes_dev[k * n_cols + col][32 B], k < n_nodes. */
int  lcpc_commit_shard_device(lcpc_commit_t *cm, const uint64_t *coeffs_local_dev, uint64_t n_rows_total,
                              void *stream, uint32_t flags, uint8_t *nodes_dev);
/* phase 2 (after the all-gather): gathered_dev is the raw all-gather output, rank g's nodes at
 * [(g * slots_per_rank + k) * n_cols + col][32 B] (slots_per_rank >= max nodes per rank; unused slots ignored)
 * -> leaf digests, Merkle tree, root.  The buffer is clobbered.
 * slots_per_rank == 0 selects the COMPACT layout the native exchange uses: slot g = node 0 of rank g (g < shard_count),
 * then the nodes k >= 1 of all ranks in rank order from slot shard_count on. */
int  lcpc_commit_finish_device(lcpc_commit_t *cm, uint8_t *gathered_dev, uint64_t n_rows_total, uint32_t slots_per_rank,
                               void *stream, uint8_t *root);
/* collapse on the local rows only (tensor entries for the local rows); partial results are summed
 * mod p by lcpc_field_sum_device after an all-gather. */
int  lcpc_collapse_device(lcpc_commit_t *cm, const uint64_t *tensors_dev, uint32_t n_tensors, void *stream,
                          uint64_t *polys_dev);
int  lcpc_field_sum_device(lcpc_ctx *ctx, const uint64_t *parts_dev, uint32_t n_parts, uint64_t n_elems,
                           void *stream, uint64_t *out_dev);
/* LcCommit::prove (lib.rs:1004-1093) on a row-sharded commitment (after the sharded commit on every rank).
 * collapse_columns splits by rows: each rank sums its rows, the partial
 * polynomials are all-gathered and added mod p; open_column gathers each rank's rows of the requested columns and
 * all-gathers them; transcript, challenges and bincode are computed identically on every rank, so every rank returns
 * the same proof bytes (== the unsharded proof).  The exchange is the caller's: `allgather(user, bytes)` must
 * all-gather the first `bytes` bytes of send_dev from every rank into recv_dev (rank g's block at g * bytes) -- e.g.
 * torch.distributed.all_gather_into_tensor -- and return 0 once recv_dev is complete and
 * visible to the null stream.  All ranks must call with the same outer_tensor (all n_rows_total entries) and the
 * same transcript state.  send_dev: max_bytes, recv_dev: shard_count * max_bytes device bytes,
 * max_bytes >= lcpc_prove_sharded_bytes().  3 exchanges for n_degree_tests = 1. */
typedef int (*lcpc_allgather_fn)(void *user, uint64_t bytes);
uint64_t lcpc_prove_sharded_bytes(const lcpc_ctx *ctx, uint64_t n_rows_total);
int  lcpc_prove_sharded(lcpc_commit_t *cm, const uint64_t *outer_tensor, uint64_t n_outer, lcpc_transcript *tr,
                        uint8_t *send_dev, uint8_t *recv_dev, uint64_t max_bytes, lcpc_allgather_fn allgather, void *user,
                        uint8_t **proof, uint64_t *proof_len, uint64_t *cols_opened);

/* ---- measurement hooks (bench.py): HIP-event time of each kernel group of the last commit, ms ---- */
typedef struct {
  float encode_ms, hash_ms, merkle_ms, total_ms;
  uint32_t encode_launches, hash_launches, merkle_launches;
  /* lcpc_commit_sharded_device only: how long the commit's stream stood still between the end of its last hash launch and
   * the arrival of the last slice's leaf digests from the exchange stream (the part of the exchange that is NOT hidden;
   * contained in merkle_ms).  0 elsewhere. */
  float exchange_exposed_ms;
  /* lcpc_commit (host pointer) only: H2D copies of the last call that were staged through the library's pinned bounce ring because
   * the source was pageable memory (0: the source was pinned / registered and was copied from directly).  Set with or without
   * lcpc_set_timing. */
  uint32_t staged_slices;
  /* lcpc_commit_sharded_device only: from the end of the local column hash to the end of the exchange's collectives on the
   * stream that carries them (the wire alone; exchange_exposed_ms - exchange_wire_ms = the leaf digests). */
  float exchange_wire_ms;
} lcpc_timings;
/* the exchange of this object's last lcpc_commit_sharded_device ALONE, once more, on `stream` (the same collectives on the same
 * buffers; the finished commitment is not touched): bench.py times 20 of them to MEASURE the wire instead of assuming it.
 * bytes_in: what this rank receives from the others per exchange.  Enqueues only.  A collective: every rank, the same order. */
int  lcpc_shard_exchange_probe(lcpc_commit_t *cm, void *stream, uint64_t *bytes_in);
/* ncclGetVersion of the communicator library the native exchange uses (0: it exports none); LCPC_ERR_NO_RCCL without one */
int  lcpc_comm_rccl_version(int *version);
int  lcpc_set_timing(lcpc_commit_t *cm, int enable);
int  lcpc_get_timings(lcpc_commit_t *cm, lcpc_timings *out);

#ifdef __cplusplus
}
#endif
#endif /* LCPC_HIP_H */
