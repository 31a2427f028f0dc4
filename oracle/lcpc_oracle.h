/*
 * oracle/lcpc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C-ABI of the CPU restatement of conroi/lcpc's lcpc-2d commit / prove / verify
 * path (see lcpc_oracle.c for the per-function reference citations and the
 * parity status).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (lcpc_amd/) never does.
 *
 * Field elements cross this boundary exactly as ff_derive stores them:
 * L little-endian uint64_t limbs in Montgomery form (R = 2^(64 L)).
 */
#ifndef LCPC_ORACLE_H
#define LCPC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { LO_FT63 = 0, LO_FT127 = 1, LO_FT191 = 2, LO_FT255 = 3 };
enum { LO_ENC_LIGERO = 0, LO_ENC_SDIG = 1 };

/* error codes: mirror ProverError (lcpc-2d/src/lib.rs:111-131) and
 * VerifierError (lib.rs:137-166) */
enum {
  LO_OK = 0,
  LO_ERR_TOO_BIG = -1, LO_ERR_ENCODE = -2, LO_ERR_COMMIT = -3, LO_ERR_COLUMN_NUMBER = -4,
  LO_ERR_OUTER_TENSOR = -5,
  LO_VERR_NUM_COL_OPENS = -32, LO_VERR_COLUMN_PATH = -33, LO_VERR_COLUMN_EVAL = -34,
  LO_VERR_COLUMN_DEGREE = -35, LO_VERR_OUTER_TENSOR = -36, LO_VERR_INNER_TENSOR = -37,
  LO_VERR_ENCODING_DIMS = -38, LO_VERR_ENCODE = -39, LO_VERR_MALFORMED = -40,
  LO_ERR_ARG = -64
};

/* ---- field ---- */
int  lo_field_limbs(int fid);
int  lo_field_info(int fid, uint64_t *modulus, uint64_t *r, uint64_t *r2, uint64_t *inv,
                   uint64_t *root_of_unity_mont, uint32_t *two_adicity, uint32_t *num_bits);
void lo_f_mul(int fid, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void lo_f_add(int fid, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void lo_f_sub(int fid, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void lo_f_to_repr(int fid, const uint64_t *a, uint8_t *out, size_t n);      /* canonical LE bytes */
void lo_f_from_canon(int fid, const uint64_t *canon, uint64_t *out, size_t n); /* canon limbs -> Montgomery */
void lo_f_from_u64(int fid, const uint64_t *v, uint64_t *out, size_t n);

/* ---- NTT (fffft fft_io_pc) ---- */
int  lo_roots_table(int fid, unsigned log_n, uint64_t *out /* max(1,n/2) * L */);
int  lo_fft_io(int fid, uint64_t *x, unsigned log_n);

/* ---- hashes / rng ---- */
void lo_blake3(const uint8_t *in, size_t len, uint8_t out[32]);
void lo_keccak_f1600(uint8_t state[200]);
typedef struct lo_rng lo_rng;
lo_rng  *lo_rng_from_seed(const uint8_t seed[32]);
lo_rng  *lo_rng_seed_from_u64(uint64_t s);
void     lo_rng_set_stream(lo_rng *, uint64_t stream);
uint32_t lo_rng_next_u32(lo_rng *);
uint64_t lo_rng_next_u64(lo_rng *);
uint64_t lo_rng_uniform(lo_rng *, uint64_t high);
void     lo_rng_field_random(lo_rng *, int fid, uint64_t *out, size_t n);
void     lo_rng_free(lo_rng *);

/* ---- merlin transcript ---- */
typedef struct lo_transcript lo_transcript;
lo_transcript *lo_tr_new(const uint8_t *label, size_t len);
lo_transcript *lo_tr_clone(const lo_transcript *);
void lo_tr_append_message(lo_transcript *, const uint8_t *label, size_t llen, const uint8_t *msg, size_t mlen);
void lo_tr_challenge_bytes(lo_transcript *, const uint8_t *label, size_t llen, uint8_t *out, size_t n);
void lo_tr_free(lo_transcript *);

/* ---- encodings (LcEncoding implementors) ---- */
typedef struct lo_enc lo_enc;
int     lo_ligero_get_dims(int fid, uint64_t len, unsigned rho_num, unsigned rho_den,
                           uint64_t *n_rows, uint64_t *n_per_row, uint64_t *n_cols);
lo_enc *lo_ligero_new(int fid, uint64_t len, unsigned rho_num, unsigned rho_den);
lo_enc *lo_ligero_new_from_dims(int fid, uint64_t n_per_row, uint64_t n_cols, unsigned rho_num, unsigned rho_den);
int     lo_sdig_get_dims(int fid, uint64_t len, int code,
                         uint64_t *n_rows, uint64_t *n_per_row, uint64_t *n_cols);
int     lo_ligero_get_dims_ml(int fid, unsigned n_vars, unsigned rho_num, unsigned rho_den,
                              uint64_t *n_rows, uint64_t *n_per_row, uint64_t *n_cols);   /* new_ml, ligero lib.rs:128-135 */
int     lo_sdig_get_dims_ml(int fid, unsigned n_vars, int code,
                            uint64_t *n_rows, uint64_t *n_per_row, uint64_t *n_cols);     /* new_ml, brakedown lib.rs:114-123 */
lo_enc *lo_sdig_new(int fid, uint64_t len, uint64_t seed, int code);
lo_enc *lo_sdig_new_from_dims(int fid, uint64_t n_per_row, uint64_t n_cols, uint64_t seed, int code);
void    lo_enc_free(lo_enc *);
void    lo_enc_get_dims(const lo_enc *, uint64_t len, uint64_t *n_rows, uint64_t *n_per_row, uint64_t *n_cols);
int     lo_enc_dims_ok(const lo_enc *, uint64_t n_per_row, uint64_t n_cols);
uint64_t lo_enc_n_col_opens(const lo_enc *);
uint64_t lo_enc_n_degree_tests(const lo_enc *);
int     lo_enc_encode(const lo_enc *, uint64_t *row /* n_cols elements, in place */);
/* expander matrices (CSC, as sprs stores them): which = 0 precode, 1 postcode */
int     lo_sdig_n_levels(const lo_enc *);
int     lo_sdig_matrix(const lo_enc *, int level, int which, uint64_t *rows, uint64_t *cols, uint64_t *nnz,
                       const uint64_t **colptr, const uint64_t **rowidx, const uint64_t **vals);

/* ---- commit ---- */
typedef struct lo_commit lo_commit;
int  lo_commit_new(const lo_enc *, const uint64_t *coeffs, uint64_t n_coeffs, int n_threads, lo_commit **out);
/* build an LcCommit from a caller-supplied comm matrix (lcpc-2d/src/tests.rs:435-466 random_comm) */
int  lo_commit_from_parts(const lo_enc *, const uint64_t *comm, const uint64_t *coeffs,
                          uint64_t n_rows, lo_commit **out);
void lo_commit_free(lo_commit *);
void lo_commit_dims(const lo_commit *, uint64_t *n_rows, uint64_t *n_per_row, uint64_t *n_cols, uint64_t *n_hashes);
const uint64_t *lo_commit_comm(const lo_commit *);
const uint64_t *lo_commit_coeffs(const lo_commit *);
const uint8_t  *lo_commit_hashes(const lo_commit *);
void lo_commit_root(const lo_commit *, uint8_t out[32]);
void lo_merkleize(lo_commit *, int n_threads);       /* lib.rs:690-704 */
void lo_merkleize_ser(lo_commit *);                  /* lib.rs:1127-1158 */
int  lo_collapse_columns(const lo_commit *, const uint64_t *tensor, uint64_t n_tensor, uint64_t *poly, int n_threads);
int  lo_open_column(const lo_commit *, uint64_t column, uint64_t *col_out, uint8_t *path_out);
void lo_hash_column(int fid, const uint64_t *col, uint64_t n_rows, uint8_t out[32]);
/* row-sharded hashing (DESIGN.md multi-GPU): BLAKE3 chaining values of leaf-message chunks
 * [chunk_begin, chunk_end) for every column, from the rows [row_base, row_base + n_rows_local) of comm;
 * cvs[(chunk - chunk_begin) * n_cols + col][32].  With a single chunk the "CV" is the final digest. */
int  lo_leaf_chunk_cvs(int fid, const uint64_t *comm_local, uint64_t n_cols, uint64_t row_base, uint64_t n_rows_local,
                       uint64_t n_rows_total, uint64_t chunk_begin, uint64_t chunk_end, uint8_t *cvs);
/* fold all chunk CVs of every column into leaf digests (BLAKE3 parent rule), then the Merkle tree */
int  lo_finish_from_cvs(const uint8_t *all_cvs, uint64_t n_chunks, uint64_t n_cols, uint8_t *hashes /* (2*np2-1)*32 */);
/* the same two with the columns spread over n_threads, and the encode loop of commit alone (lib.rs:648-653): together a
 * STREAMING commit -- encode a chunk-aligned block of rows, reduce it to chunk CVs, drop it -- whose host memory is a few
 * row blocks instead of the whole commitment (tests/oracle_lib.py commit_streaming: the 2^28 whole-tree check) */
int  lo_leaf_chunk_cvs_mt(int fid, const uint64_t *comm_local, uint64_t n_cols, uint64_t row_base, uint64_t n_rows_local,
                          uint64_t n_rows_total, uint64_t chunk_begin, uint64_t chunk_end, uint8_t *cvs, int n_threads);
int  lo_finish_from_cvs_mt(const uint8_t *all_cvs, uint64_t n_chunks, uint64_t n_cols, uint8_t *hashes, int n_threads);
int  lo_encode_rows(const lo_enc *, const uint64_t *coeffs /* n_rows x n_per_row */, uint64_t n_rows, int n_threads,
                    uint64_t *comm /* n_rows x n_cols */);

/* ---- prove / verify (bincode 1.3 wire format) ---- */
int  lo_prove(const lo_commit *, const lo_enc *, const uint64_t *outer_tensor, uint64_t n_outer,
              lo_transcript *, uint8_t **proof, uint64_t *proof_len, uint64_t *cols_opened /* may be NULL */);
int  lo_verify(const lo_enc *, const uint8_t root[32], const uint64_t *outer_tensor, uint64_t n_outer,
               const uint64_t *inner_tensor, uint64_t n_inner, const uint8_t *proof, uint64_t proof_len,
               lo_transcript *, uint64_t *eval_out);
void lo_free(void *);

#ifdef __cplusplus
}
#endif
#endif
