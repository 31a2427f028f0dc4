/* examples/abi_demo.c -- the drop-in boundary used from plain C (what a cgo / Rust-FFI / JNI binding would call).
 *
 *   gcc -O2 -Iinclude examples/abi_demo.c -Llcpc_amd/lib -llcpc_hip -Wl,-rpath,$PWD/lcpc_amd/lib -o examples/abi_demo
 *   ./examples/abi_demo            # prints the Merkle root of a fixed ft63 commitment + a verified evaluation
 *
 * Commits c_i = i + 1 (i < 1024) over Ft63 with LigeroEncoding::new(1024), proves and verifies an evaluation, then
 * streams the commitment out in the reference's serde layout and back into a second object, all through include/lcpc_hip.h.  tests/test_gpu_abi_demo.py checks the printed root against the golden
 * fixture "ligero_ft63_2e10_iota" (tests/golden/commit_cases.json). */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lcpc_hip.h"

/* memory sink / source for the streamed serde of LcCommit (lcpc_commit_bincode_write / lcpc_commit_from_bincode) */
typedef struct { uint8_t *p; uint64_t len, cap, pos; } membuf;
static int sink(void *user, const uint8_t *data, uint64_t len) {
  membuf *b = (membuf *)user;
  if (b->len + len > b->cap) return 1;
  memcpy(b->p + b->len, data, len);
  b->len += len;
  return 0;
}
static int source(void *user, uint8_t *data, uint64_t len) {
  membuf *b = (membuf *)user;
  if (b->pos + len > b->len) return 1;
  memcpy(data, b->p + b->pos, len);
  b->pos += len;
  return 0;
}

#define CHECK(call) do { int rc__ = (call); if (rc__) { fprintf(stderr, "%s -> %d (%s)\n", #call, rc__, lcpc_strerror(rc__)); return 1; } } while (0)

/* Ft63: p = 0x46d0760000000001, Montgomery form of v is v * 2^64 mod p (one u64 limb) */
static uint64_t to_mont63(uint64_t v) {
  const unsigned __int128 p = 0x46d0760000000001ull;
  return (uint64_t)((((unsigned __int128)v) << 64) % p);
}

int main(void) {
  lcpc_params prm;
  memset(&prm, 0, sizeof prm);
  prm.field = LCPC_FT63; prm.encoding = LCPC_ENC_LIGERO; prm.hash = LCPC_HASH_BLAKE3;
  prm.rho_num = 1; prm.rho_den = 2; prm.n_coeffs = 1024; prm.device = 0; prm.shard_count = 1;
  lcpc_ctx *ctx = NULL;
  CHECK(lcpc_ctx_create(&prm, &ctx));
  uint64_t n_rows, n_per_row, n_cols;
  CHECK(lcpc_get_dims(ctx, 1024, &n_rows, &n_per_row, &n_cols));
  printf("dims %" PRIu64 " x %" PRIu64 " -> %" PRIu64 ", col_opens %" PRIu64 ", degree_tests %" PRIu64 "\n", n_rows, n_per_row, n_cols,
         lcpc_get_n_col_opens(ctx), lcpc_get_n_degree_tests(ctx));
  uint64_t coeffs[1024];
  for (int i = 0; i < 1024; i++) coeffs[i] = to_mont63((uint64_t)i + 1);
  uint8_t root[32];
  lcpc_commit_t *cm = NULL;                     /* LcCommit::commit(&coeffs, &enc) */
  CHECK(lcpc_commit_create(ctx, &cm));
  CHECK(lcpc_commit(cm, coeffs, 1024, root));
  printf("root ");
  for (int i = 0; i < 32; i++) printf("%02x", root[i]);
  printf("\n");

  /* evaluate at x = 0x1234567: outer = (x^n_per_row)^r, inner = x^j (lcpc-ligero-pc/src/tests.rs:226-252) */
  const unsigned __int128 p = 0x46d0760000000001ull;
  const uint64_t x = 0x1234567;
  uint64_t *inner = malloc(n_per_row * 8), *outer = malloc(n_rows * 8), cur = 1, xr;
  for (uint64_t j = 0; j < n_per_row; j++) { inner[j] = to_mont63(cur); cur = (uint64_t)(((unsigned __int128)cur * x) % p); }
  xr = cur;
  cur = 1;
  for (uint64_t r = 0; r < n_rows; r++) { outer[r] = to_mont63(cur); cur = (uint64_t)(((unsigned __int128)cur * xr) % p); }
  uint8_t ncols_be[8];
  uint64_t nco = lcpc_get_n_col_opens(ctx);
  for (int i = 0; i < 8; i++) ncols_be[i] = (uint8_t)(nco >> (56 - 8 * i));
  lcpc_transcript *tp = lcpc_transcript_new((const uint8_t *)"test transcript", 15);
  lcpc_transcript_append_message(tp, (const uint8_t *)"polycommit", 10, root, 32);
  lcpc_transcript_append_message(tp, (const uint8_t *)"ncols", 5, ncols_be, 8);
  lcpc_transcript *tv = lcpc_transcript_clone(tp);
  uint8_t *proof = NULL;
  uint64_t proof_len = 0;
  CHECK(lcpc_prove(cm, outer, n_rows, tp, &proof, &proof_len, NULL));
  uint64_t eval_mont = 0;
  CHECK(lcpc_verify(ctx, root, outer, n_rows, inner, n_per_row, proof, proof_len, tv, &eval_mont));
  /* the true evaluation sum (i+1) x^i, by Horner, and the verifier's answer brought out of Montgomery form */
  unsigned __int128 acc = 0;
  for (int i = 1023; i >= 0; i--) acc = (acc * x + (uint64_t)(i + 1)) % p;
  unsigned __int128 rinv = 1;   /* 2^-64 mod p by repeated halving */
  for (int i = 0; i < 64; i++) rinv = (rinv & 1) ? (rinv + p) >> 1 : rinv >> 1;
  const uint64_t eval = (uint64_t)(((unsigned __int128)eval_mont * rinv) % p);
  printf("proof_bytes %" PRIu64 "\neval %016" PRIx64 " expected %016" PRIx64 " %s\n", proof_len, eval, (uint64_t)acc,
         eval == (uint64_t)acc ? "OK" : "MISMATCH");
  /* the whole commitment through the reference's serde layout (lcpc-2d/src/lib.rs:186-268) and back into a second object:
   * same root, and the same proof bytes from an identical transcript */
  membuf mb;
  mb.cap = lcpc_commit_bincode_size(cm); mb.len = 0; mb.pos = 0;
  mb.p = malloc(mb.cap);
  CHECK(lcpc_commit_bincode_write(cm, sink, &mb));
  lcpc_commit_t *cm2 = NULL;
  uint8_t root2[32];
  CHECK(lcpc_commit_create(ctx, &cm2));
  CHECK(lcpc_commit_from_bincode(cm2, source, &mb, root2));
  lcpc_transcript *tp2 = lcpc_transcript_new((const uint8_t *)"test transcript", 15);
  lcpc_transcript_append_message(tp2, (const uint8_t *)"polycommit", 10, root, 32);
  lcpc_transcript_append_message(tp2, (const uint8_t *)"ncols", 5, ncols_be, 8);
  uint8_t *proof2 = NULL;
  uint64_t proof2_len = 0;
  CHECK(lcpc_prove(cm2, outer, n_rows, tp2, &proof2, &proof2_len, NULL));
  const int serde_ok = mb.len == mb.cap && mb.pos == mb.len && memcmp(root, root2, 32) == 0 && proof2_len == proof_len &&
                       memcmp(proof, proof2, proof_len) == 0;
  printf("commit_bincode_bytes %" PRIu64 " %s\n", mb.len, serde_ok ? "OK" : "MISMATCH");
  lcpc_free(proof2);
  lcpc_transcript_free(tp2);
  lcpc_commit_destroy(cm2);
  free(mb.p);
  lcpc_free(proof);
  lcpc_transcript_free(tp);
  lcpc_transcript_free(tv);
  free(inner);
  free(outer);
  lcpc_commit_destroy(cm);
  lcpc_ctx_destroy(ctx);
  return eval == (uint64_t)acc && serde_ok ? 0 : 2;
}
