// lcpc_amd/csrc/prove.cpp -- LcCommit::prove (lcpc-2d/src/lib.rs:1004-1093), LcEvalProof::verify (lib.rs:832-1000)
// and the bincode wire layout (lib.rs:550-609) of /root/reference, plus the C wrappers of merlin::Transcript.
//
// The heavy steps run on the GPU (collapse_columns, open_column, the verifier's row encodes); the Fiat-Shamir
// transcript is serial by construction and runs on the host (host_crypto.cpp).
#include "internal.h"
#include <chrono>
#include <memory>

using namespace lcpc;

namespace lcpc {

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// tr.append_message(label, to_repr(poly[i])) for every coefficient (lib.rs:1045-1047, 1066-1068): to_repr
// (Montgomery -> canonical little-endian, lib.rs:47-57) is independent per element and done in parallel; only the
// STROBE absorb itself is serial
static void absorb_poly(Transcript& tr, const uint8_t* label, const FieldDesc& f, const uint64_t* poly, uint64_t n) {
  const int L = f.L;
  std::vector<uint64_t> canon(n * L);
  parallel_for(n, 4096, [&](uint64_t b, uint64_t e) { for (uint64_t i = b; i < e; i++) h_canon(f, &canon[i * L], poly + i * L); });
  tr.append_messages(label, 6, reinterpret_cast<const uint8_t*>(canon.data()), 8 * L, n);
}

int prove_impl(lcpc_commit_t* m, const uint64_t* outer, uint64_t n_outer, lcpc_transcript* trw, uint8_t** proof, uint64_t* proof_len,
               uint64_t* cols_opened, const ShardXchg* xchg) {
  if (!m || !outer || !trw || !proof || !proof_len) return LCPC_ERR_ARG;
  if (!m->committed) return LCPC_ERR_STATE;
  lcpc_ctx* c = m->enc;
  if (c->prm.shard_count > 1 && !xchg) return LCPC_ERR_STATE;   // sharded commitments prove through lcpc_prove_sharded*
  auto collapse = [&](const uint64_t* tensors, uint32_t nt, uint64_t* polys) -> int {
    return xchg ? collapse_sharded(m, *xchg, tensors, nt, polys) : lcpc_collapse(m, tensors, nt, polys);
  };
  if (!lcpc_dims_ok(c, c->n_per_row, c->n_cols)) return LCPC_ERR_COMMIT;      // check_comm lib.rs:1015
  if (n_outer != m->n_rows) return LCPC_ERR_OUTER_TENSOR;                     // lib.rs:1016-1018
  const FieldDesc& f = *c->f;
  const int L = f.L;
  Transcript& tr = trw->t;
  const uint64_t n_deg = lcpc_get_n_degree_tests(c), n_open = lcpc_get_n_col_opens(c);
  const uint64_t np = c->n_per_row, nr = m->n_rows;
  const bool dbg = getenv("LCPC_DEBUG_TIMING") != nullptr;
  double tp[8] = {now_ms(), 0, 0, 0, 0, 0, 0, 0};
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
    double t0 = now_ms();
    int rc = collapse(tensors.data(), nt, polys.data());
    if (rc) return rc;
    t_collapse += now_ms() - t0;
    p_random[i].assign(polys.begin(), polys.begin() + np * L);
    if (nt == 2) { memcpy(p_eval.data(), &polys[np * L], np * L * 8); have_eval = true; }
    t0 = now_ms();
    absorb_poly(tr, LBL_PR, f, p_random[i].data(), np);
    t_absorb += now_ms() - t0;
  }
  if (!have_eval) {                                                           // lib.rs:1053-1064
    int rc = collapse(outer, 1, p_eval.data());
    if (rc) return rc;
  }
  tp[1] = now_ms();
  absorb_poly(tr, LBL_PE, f, p_eval.data(), np);                              // lib.rs:1066-1068
  tp[2] = now_ms();
  uint8_t key[32];
  tr.challenge_bytes(LBL_CO, 6, key, 32);                                     // lib.rs:1071-1080
  ChaCha20Rng rng(key);
  std::vector<uint64_t> cols(n_open);
  for (auto& x : cols) x = rng.uniform(c->n_cols);
  if (cols_opened) memcpy(cols_opened, cols.data(), n_open * 8);
  // (uninitialised buffers: a Brakedown proof opens 6593 columns, tens of MB that are overwritten anyway)
  std::unique_ptr<uint64_t[]> vals(new uint64_t[(size_t)n_open * nr * L + 1]);
  std::unique_ptr<uint8_t[]> paths(new uint8_t[(size_t)n_open * c->path_len * 32 + 32]);
  tp[3] = now_ms();
  int rc = xchg ? open_sharded(m, *xchg, cols.data(), (uint32_t)n_open, vals.get(), paths.get())
                : lcpc_open_columns(m, cols.data(), (uint32_t)n_open, vals.get(), paths.get());   // lib.rs:1081-1084
  if (rc) return rc;
  tp[4] = now_ms();
  // bincode 1.3 of WrappedLcEvalProof (lib.rs:550-560): n_cols, p_eval, p_random_vec, columns -- the size is known
  // up front, so the proof is written once, straight into the buffer the caller receives
  const size_t pbytes = np * L * 8;
  const size_t col_bytes = 8 + nr * L * 8 + 8 + (size_t)c->path_len * 40;
  const size_t head = 8 + (8 + pbytes) + 8 + n_deg * (8 + pbytes) + 8;
  const size_t total = head + n_open * col_bytes;
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
  if ((size_t)(w - out) != head) { free(out); return LCPC_ERR_STATE; }
  parallel_for(n_open, 64, [&](uint64_t b, uint64_t e) {
    for (uint64_t k = b; k < e; k++) {
      uint8_t* q = out + head + k * col_bytes;
      auto q64 = [&](uint64_t v) { memcpy(q, &v, 8); q += 8; };
      q64(nr); memcpy(q, &vals[k * nr * L], nr * L * 8); q += nr * L * 8;
      q64(c->path_len);
      for (uint32_t l = 0; l < c->path_len; l++) { q64(32); memcpy(q, &paths[((size_t)k * c->path_len + l) * 32], 32); q += 32; }
    }
  });
  *proof = out; *proof_len = total;
  if (dbg)
    fprintf(stderr, "[lcpc_prove] collapse %.2f ms, absorb p_random %.2f, absorb p_eval %.2f, challenges+alloc %.2f, open %.2f, bincode %.2f, total %.2f\n",
            t_collapse, t_absorb, tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], now_ms() - tp[4], now_ms() - tp[0]);
  return 0;
}

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
// every element of an untrusted vector must be a reduced Montgomery representative (< p): the device arithmetic
// (lazy-limb NTT, lazy dot products) is only proven for reduced inputs.  The reference's derived Deserialize
// (lcpc-test-fields/src/lib.rs:18-58) takes the raw limbs unchecked and its CPU arithmetic stays correct mod p for
// them; such a proof is refused here (LCPC_VERR_MALFORMED) -- an honest prover never produces one.
bool all_reduced(const FieldDesc& f, const uint64_t* v, uint64_t n) {
  std::atomic<bool> ok{true};
  parallel_for(n, 1 << 15, [&](uint64_t b, uint64_t e) {
    bool good = true;
    for (uint64_t i = b; i < e && good; i++) good = !h_ge_p(f, v + i * f.L);
    if (!good) ok.store(false);
  });
  return ok.load();
}
struct JoinGuard {          // a joinable std::thread must never be destroyed (std::terminate): join on every exit path
  std::thread& t;
  ~JoinGuard() { if (t.joinable()) t.join(); }
};
}  // namespace

}  // namespace lcpc

extern "C" {

// ---- transcript ------------------------------------------------------------------------------------------
lcpc_transcript* lcpc_transcript_new(const uint8_t* label, size_t len) { return new (std::nothrow) lcpc_transcript(label, len); }
lcpc_transcript* lcpc_transcript_clone(const lcpc_transcript* t) { return t ? new (std::nothrow) lcpc_transcript(*t) : nullptr; }
void lcpc_transcript_append_message(lcpc_transcript* t, const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen) {
  if (t) t->t.append_message(label, llen, msg, mlen);
}
void lcpc_transcript_challenge_bytes(lcpc_transcript* t, const uint8_t* label, size_t llen, uint8_t* out, size_t n) {
  if (t) t->t.challenge_bytes(label, llen, out, n);
}
void lcpc_transcript_free(lcpc_transcript* t) { delete t; }

int lcpc_prove(lcpc_commit_t* m, const uint64_t* outer, uint64_t n_outer, lcpc_transcript* trw, uint8_t** proof, uint64_t* proof_len,
               uint64_t* cols_opened) {
  LCPC_TRY
  return prove_impl(m, outer, n_outer, trw, proof, proof_len, cols_opened, nullptr);
  LCPC_CATCH(m)
}

// ---- verify (lib.rs:832-1000) ----------------------------------------------------------------------------
int lcpc_verify(lcpc_ctx* c, const uint8_t root[32], const uint64_t* outer, uint64_t n_outer, const uint64_t* inner, uint64_t n_inner,
                const uint8_t* proof, uint64_t proof_len, lcpc_transcript* trw, uint64_t* eval_out) {
  if (!c || !root || !outer || !inner || !proof || !trw || !eval_out) return LCPC_ERR_ARG;
  LCPC_TRY
  const FieldDesc& f = *c->f;
  const int L = f.L;
  const uint64_t F = 8 * L;
  Transcript& tr = trw->t;
  const bool dbg = getenv("LCPC_DEBUG_TIMING") != nullptr;
  double tv[6] = {now_ms(), 0, 0, 0, 0, 0};
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
  // untrusted limbs: nothing >= p reaches the host or device arithmetic
  if (!all_reduced(f, p_eval.data(), n_per_row)) return LCPC_VERR_MALFORMED;
  for (uint64_t i = 0; i < n_deg_pf; i++) if (!all_reduced(f, p_random[i].data(), p_random[i].size() / L)) return LCPC_VERR_MALFORMED;
  {
    std::atomic<bool> ok{true};
    parallel_for(n_columns, 16, [&](uint64_t b, uint64_t e) {
      for (uint64_t i = b; i < e; i++)
        for (uint64_t k = 0; k < n_rows; k++)
          if (h_ge_p(f, cols[i].data() + k * L)) { ok.store(false); return; }
    });
    if (!ok.load()) return LCPC_VERR_MALFORMED;
  }
  // step 2 first: the 1 + n_deg row encodes (lib.rs:886, 918) depend only on the proof, not on the transcript, so they
  // run on the GPU (own thread: upload, kernels, download) while this thread does step 1, the serial transcript work
  std::vector<uint64_t> enc((n_deg + 1) * n_cols * L, 0);
  for (uint64_t i = 0; i < n_deg; i++) memcpy(&enc[i * n_cols * L], p_random[i].data(), n_per_row * F);
  memcpy(&enc[n_deg * n_cols * L], p_eval.data(), n_per_row * F);
  std::vector<std::vector<uint64_t>> rand_tensors(n_deg, std::vector<uint64_t>(n_rows * L));
  std::vector<uint64_t> cols_to_open(n_columns);
  std::vector<int> status(n_columns, 0);
  int enc_rc = 0;
  double t_enc = 0;
  tv[1] = now_ms();
  std::thread enc_thread;
  JoinGuard join{enc_thread};
  enc_thread = std::thread([&] { const double t0 = now_ms(); enc_rc = lcpc_encode_rows(c, enc.data(), n_deg + 1); t_enc = now_ms() - t0; });
  // step 1: random tensors, transcript (lib.rs:868-920)
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
  tv[2] = now_ms();
  enc_thread.join();
  tv[3] = now_ms();
  if (enc_rc) return enc_rc == LCPC_ERR_ENCODE ? LCPC_VERR_ENCODE : enc_rc;
  // step 3: per-column checks (lib.rs:923-944), in parallel over columns like the reference's par_iter;
  // the error reported is that of the first failing column, with the reference's precedence degree > eval > path
  for (auto& x : cols_to_open) x = rng.uniform(n_cols);
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
  tv[4] = now_ms();
  for (uint64_t i = 0; i < n_columns; i++)
    if (status[i]) return status[i];
  if (dbg)
    fprintf(stderr, "[lcpc_verify] parse %.2f ms, transcript %.2f (row encodes on the GPU meanwhile: %.2f), wait for encodes %.2f, column checks %.2f\n",
            tv[1] - tv[0], tv[2] - tv[1], t_enc, tv[3] - tv[2], tv[4] - tv[3]);
  // <inner_tensor, p_eval> (lib.rs:947-951): partial sums over blocks of 4096 terms, added in block order
  const uint64_t n_blk = (n_per_row + 4095) / 4096;
  std::vector<uint64_t> part(n_blk * MAXL, 0);
  parallel_for(n_blk, 2, [&](uint64_t b, uint64_t e) {
    for (uint64_t blk = b; blk < e; blk++) {
      uint64_t a[MAXL] = {0, 0, 0, 0}, t[MAXL];
      const uint64_t k1 = std::min<uint64_t>(n_per_row, (blk + 1) * 4096);
      for (uint64_t k = blk * 4096; k < k1; k++) { h_mul(f, t, inner + k * L, p_eval.data() + k * L); h_add(f, a, a, t); }
      memcpy(&part[blk * MAXL], a, F);
    }
  });
  uint64_t acc[MAXL] = {0, 0, 0, 0};
  for (uint64_t blk = 0; blk < n_blk; blk++) h_add(f, acc, acc, &part[blk * MAXL]);
  memcpy(eval_out, acc, F);
  return 0;
  LCPC_CATCH(c)
}

}  // extern "C"
