// lcpc_amd/csrc/prove.cpp -- LcCommit::prove (lcpc-2d/src/lib.rs:1004-1093), LcEvalProof::verify (lib.rs:832-1000)
// and the bincode wire layout (lib.rs:550-609) of /root/reference, plus the C wrappers of merlin::Transcript.
//
// The heavy steps run on the GPU (collapse_columns, open_column, the verifier's row encodes); the Fiat-Shamir
// transcript is serial by construction and runs on the host (host_crypto.cpp).
#include "internal.h"
#include <chrono>
#include <memory>
#include <mutex>
#include <unordered_map>

using namespace lcpc;

namespace lcpc {

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// tr.append_message(label, to_repr(poly[i])) for every coefficient (lib.rs:1045-1047, 1066-1068).  to_repr (Montgomery ->
// canonical little-endian, lib.rs:47-57) is independent per element: prove (sharded or not) gets it from the device with the
// polynomial, verify converts in parallel on the host; only the STROBE absorb itself is serial.
static void absorb_canon(Transcript& tr, const uint8_t* label, const FieldDesc& f, const uint64_t* canon, uint64_t n) {
  tr.append_messages(label, 6, reinterpret_cast<const uint8_t*>(canon), 8 * f.L, n);
}

namespace {
struct JoinGuard {          // a joinable std::thread must never be destroyed (std::terminate): join on every exit path
  std::thread& t;
  ~JoinGuard() { if (t.joinable()) t.join(); }
};
}  // namespace

// Proof buffers go back to the library through lcpc_free.  One released buffer is kept for the next proof: a prove loop
// (the reference's prove_verify_size_bench, tests.rs:102-170) then reuses mapped pages instead of paying munmap + first-touch
// faults on every iteration (9 ms per 50 MB proof on a 256-thread host: TLB shootdowns).
namespace {
std::mutex g_pool_mu;
std::unordered_map<void*, size_t> g_pool_live;      // buffers handed out by proof_buf_alloc -> capacity
void* g_pool_p = nullptr;
size_t g_pool_cap = 0;
constexpr size_t POOL_MAX = (size_t)512 << 20;
}  // namespace
void* proof_buf_alloc(size_t n) {
  {
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (g_pool_p && g_pool_cap >= n && g_pool_cap / 2 <= n) {
      void* p = g_pool_p;
      g_pool_live[p] = g_pool_cap;
      g_pool_p = nullptr; g_pool_cap = 0;
      return p;
    }
  }
  void* p = malloc(n);
  if (p) {
    std::lock_guard<std::mutex> g(g_pool_mu);
    g_pool_live[p] = n;
  }
  return p;
}
void proof_buf_free(void* p) {
  if (!p) return;
  void* drop = p;
  {
    std::lock_guard<std::mutex> g(g_pool_mu);
    auto it = g_pool_live.find(p);
    if (it != g_pool_live.end()) {
      const size_t cap = it->second;
      g_pool_live.erase(it);
      if (cap <= POOL_MAX && cap > g_pool_cap) {      // keep the larger of the two, release the other
        drop = g_pool_p;
        g_pool_p = p; g_pool_cap = cap;
      }
    }
  }
  free(drop);
}

int prove_impl(lcpc_commit_t* m, const uint64_t* outer, uint64_t n_outer, lcpc_transcript* trw, uint8_t** proof, uint64_t* proof_len,
               uint64_t* cols_opened, const ShardXchg* xchg) {
  if (!m || !outer || !trw || !proof || !proof_len) return LCPC_ERR_ARG;
  std::lock_guard<std::mutex> prove_lock(m->prove_mu);
  if (!m->committed) return LCPC_ERR_STATE;
  lcpc_ctx* c = m->enc;
  if (c->prm.shard_count > 1 && !xchg) return LCPC_ERR_STATE;   // sharded commitments prove through lcpc_prove_sharded*
  const FieldDesc& f = *c->f;
  const int L = f.L;
  // polynomials in Montgomery form (what the proof carries) and as canonical values (what the transcript absorbs)
  auto collapse = [&](const uint64_t* tensors, uint32_t nt, uint64_t* polys, uint64_t* canon) -> int {
    if (!xchg) return collapse_host(m, tensors, nt, polys, canon);
    return collapse_sharded(m, *xchg, tensors, nt, polys, canon);
  };
  if (!lcpc_dims_ok(c, c->n_per_row, c->n_cols)) return LCPC_ERR_COMMIT;      // check_comm lib.rs:1015
  if (n_outer != m->n_rows) return LCPC_ERR_OUTER_TENSOR;                     // lib.rs:1016-1018
  Transcript& tr = trw->t;
  const uint64_t n_deg = lcpc_get_n_degree_tests(c), n_open = lcpc_get_n_col_opens(c);
  const uint64_t np = c->n_per_row, nr = m->n_rows;
  const bool dbg = c->sw_debug_timing;
  double tp[8] = {now_ms(), 0, 0, 0, 0, 0, 0, 0};
  double t_collapse = 0, t_absorb = 0;
  // bincode 1.3 of WrappedLcEvalProof (lib.rs:550-560): n_cols, p_eval, p_random_vec, columns.  The size is known up
  // front, so the proof is written once, straight into the buffer the caller receives: the polynomials by a helper
  // thread while this one runs the (serial) transcript, the opened columns by the device-to-host copy itself.
  const size_t pbytes = np * L * 8;
  const size_t col_bytes = 8 + nr * L * 8 + 8 + (size_t)c->path_len * 40;
  const size_t off_eval = 8 + 8, off_rand0 = off_eval + pbytes + 8 + 8;      // first element of p_eval / of p_random_vec[0]
  const size_t head = 8 + (8 + pbytes) + 8 + n_deg * (8 + pbytes) + 8;
  const size_t total = head + n_open * col_bytes;
  struct Buf { uint8_t* p = nullptr; ~Buf() { proof_buf_free(p); } } out;
  out.p = static_cast<uint8_t*>(proof_buf_alloc(total ? total : 1));
  if (!out.p) return LCPC_ERR_NOMEM;
  // pinned arena that stays with the commitment: [tensors 2 nr][polys 2 np][canon 2 np] elements
  const size_t a_t = 2 * nr * L, a_p = 2 * np * L;
  {
    std::lock_guard<std::mutex> g(m->mu);
    HIPCHK(m, hipSetDevice(c->prm.device));      // the pinned arena belongs to the encoder's device, whatever this thread used last
    int rc = ensure_pinned(m, (a_t + 2 * a_p) * 8);
    if (rc) return rc;
  }
  uint64_t* tensors = reinterpret_cast<uint64_t*>(m->h_pin);
  uint64_t* polys = tensors + a_t;
  uint64_t* canon = polys + a_p;
  std::vector<uint64_t> p_eval_canon;                                         // only when the eval tensor could not be fused
  auto put64 = [&](size_t off, uint64_t v) { memcpy(out.p + off, &v, 8); };
  std::thread filler;
  JoinGuard join{filler};
  bool have_eval = false;
  int eval_rc = 0;                                                            // of the p_eval collapse when it runs on the helper thread
  for (uint64_t i = 0; i < n_deg; i++) {                                      // lib.rs:1024-1050
    uint8_t key[32];
    tr.challenge_bytes(LBL_DT, 6, key, 32);
    ChaCha20Rng rng(key);
    for (uint64_t r = 0; r < nr; r++) rng.field_random(f, &tensors[r * L]);
    // The eval tensor is independent of the transcript.  Sharded: it is fused into the first pass over coeffs (one
    // exchange for both polynomials).  Unsharded: the transcript only waits for p_random; the helper thread collapses
    // p_eval while this one absorbs p_random (a second pass over coeffs on an otherwise idle GPU: -0.3 ms on the wait).
    uint32_t nt = 1;
    const bool eval_here = i == 0, eval_beside = eval_here && xchg == nullptr;
    if (eval_here) {
      memcpy(&tensors[nr * L], outer, nr * L * 8);
      if (!eval_beside) nt = 2;
    }
    double t0 = now_ms();
    if (filler.joinable()) filler.join();                                     // the arena is about to be overwritten
    if (eval_rc) return eval_rc;
    // unsharded, one tensor, a polynomial long enough to matter: p_random arrives in two column ranges and the absorb of the first
    // (serial STROBE, ~50 ns per coefficient) hides the computation of the second (commit.cpp collapse_host_sliced)
    const bool sliced = xchg == nullptr && nt == 1 && np >= 32768;
    uint64_t cut = np;
    int rc = sliced ? collapse_host_sliced(m, tensors, polys, canon, &cut) : collapse(tensors, nt, polys, canon);
    if (rc) return rc;
    if (sliced && (rc = collapse_wait_slice(m, 0))) return rc;
    t_collapse += now_ms() - t0;
    if (eval_here) have_eval = true;
    if (nt == 2 && n_deg > 1) p_eval_canon.assign(canon + np * L, canon + 2 * np * L);
    // the helper copies this round's polynomial(s) into the proof (and thereby faults the fresh pages in) meanwhile
    filler = std::thread([=, &out, &eval_rc, &p_eval_canon, &collapse] {
      if (eval_beside) {
        eval_rc = collapse(tensors + nr * L, 1, polys + np * L, canon + np * L);   // (queues up behind p_random's second range)
        if (eval_rc) return;
        if (n_deg > 1) p_eval_canon.assign(canon + np * L, canon + 2 * np * L);
      }
      if (sliced && (eval_rc = collapse_wait_slice(m, 1))) return;              // all of p_random is on the host before it is copied
      memcpy(out.p + off_rand0 + i * (8 + pbytes), polys, pbytes);
      if (eval_here) memcpy(out.p + off_eval, polys + np * L, pbytes);
      if (i + 1 == n_deg) memset(out.p + head, 0, total - head);              // touch the column area before the copies land in it
    });
    t0 = now_ms();
    absorb_canon(tr, LBL_PR, f, canon, cut);
    if (sliced) {
      if ((rc = collapse_wait_slice(m, 1))) return rc;
      absorb_canon(tr, LBL_PR, f, canon + cut * L, np - cut);
    }
    t_absorb += now_ms() - t0;
  }
  tp[1] = now_ms();
  if (!have_eval) {                                                           // lib.rs:1053-1064 (n_degree_tests == 0 cannot happen: >= 1)
    if (filler.joinable()) filler.join();
    int rc = collapse(outer, 1, polys, canon);
    if (rc) return rc;
    memcpy(out.p + off_eval, polys, pbytes);
    absorb_canon(tr, LBL_PE, f, canon, np);
  } else {
    if (n_deg == 1 && filler.joinable()) filler.join();                      // p_eval may still be on its way (eval_beside)
    if (eval_rc) return eval_rc;
    absorb_canon(tr, LBL_PE, f, n_deg > 1 ? p_eval_canon.data() : canon + np * L, np);   // lib.rs:1066-1068
  }
  tp[2] = now_ms();
  uint8_t key[32];
  tr.challenge_bytes(LBL_CO, 6, key, 32);                                     // lib.rs:1071-1080
  ChaCha20Rng rng(key);
  std::vector<uint64_t> cols(n_open);
  for (auto& x : cols) x = rng.uniform(c->n_cols);
  if (cols_opened) memcpy(cols_opened, cols.data(), n_open * 8);
  std::unique_ptr<uint8_t[]> paths(new uint8_t[(size_t)n_open * c->path_len * 32 + 32]);
  if (filler.joinable()) filler.join();
  tp[3] = now_ms();
  int rc;
  uint64_t* vals0 = reinterpret_cast<uint64_t*>(out.p + head + 8);           // values of column 0; column k at + k * col_bytes
  if (xchg) {
    rc = open_sharded(m, *xchg, cols.data(), (uint32_t)n_open, vals0, col_bytes, paths.get());   // straight into the bincode slots
  } else {
    rc = open_columns_host(m, cols.data(), (uint32_t)n_open, vals0, col_bytes, paths.get());      // lib.rs:1081-1084
  }
  if (rc) return rc;
  tp[4] = now_ms();
  put64(0, c->n_cols);
  put64(8, np);
  put64(off_eval + pbytes, n_deg);
  for (uint64_t i = 0; i < n_deg; i++) put64(off_rand0 - 8 + i * (8 + pbytes), np);
  put64(head - 8, n_open);
  parallel_for(n_open, 256, [&](uint64_t b, uint64_t e) {
    for (uint64_t k = b; k < e; k++) {
      uint8_t* q = out.p + head + k * col_bytes;
      auto q64 = [&](uint64_t v) { memcpy(q, &v, 8); q += 8; };
      q64(nr);
      q += nr * L * 8;                                                        // the values are already there
      q64(c->path_len);
      for (uint32_t l = 0; l < c->path_len; l++) { q64(32); memcpy(q, &paths[((size_t)k * c->path_len + l) * 32], 32); q += 32; }
    }
  });
  *proof = out.p; *proof_len = total;
  out.p = nullptr;
  if (dbg)
    fprintf(stderr, "[lcpc_prove] collapse %.2f ms, absorb p_random %.2f, absorb p_eval %.2f, challenges %.2f, open %.2f, bincode %.2f, total %.2f\n",
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
}  // namespace

}  // namespace lcpc

extern "C" {

// ---- transcript ------------------------------------------------------------------------------------------
lcpc_transcript* lcpc_transcript_new(const uint8_t* label, size_t len) { return new (std::nothrow) lcpc_transcript(label, len); }
lcpc_transcript* lcpc_transcript_clone(const lcpc_transcript* t) { return t ? new (std::nothrow) lcpc_transcript(*t) : nullptr; }
void lcpc_transcript_append_message(lcpc_transcript* t, const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen) {
  if (t) t->t.append_message(label, llen, msg, mlen);
}
void lcpc_transcript_append_messages(lcpc_transcript* t, const uint8_t* label, size_t llen, const uint8_t* msgs, size_t mlen, size_t n) {
  if (t) t->t.append_messages(label, llen, msgs, mlen, n);
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
  const bool dbg = c->sw_debug_timing;
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
  // (bytes after the last column are ignored, as by bincode::deserialize, whose legacy options allow trailing bytes)
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
  // untrusted limbs: nothing >= p reaches the device arithmetic (the polynomials are checked before they are sent; the
  // columns, which only meet host arithmetic, are checked by the side thread below)
  if (!all_reduced(f, p_eval.data(), n_per_row)) return LCPC_VERR_MALFORMED;
  for (uint64_t i = 0; i < n_deg_pf; i++) if (!all_reduced(f, p_random[i].data(), p_random[i].size() / L)) return LCPC_VERR_MALFORMED;
  // The transcript (step 1, lib.rs:868-920) is serial and takes almost all of the time; everything that does not need its
  // outcome runs beside it:
  //   enc thread   step 2, the 1 + n_deg row encodes (lib.rs:886, 918), on the GPU: the messages go up without their zero
  //                padding, straight out of the proof buffer, the encoded rows come back whole;
  //   side thread  to_repr of the polynomials the transcript absorbs (published one by one), the limb check of the
  //                columns, and the part of step 3 (lib.rs:923-944) that does not depend on WHICH columns were drawn:
  //                the tensor . column dot products and the leaf hash of every opened column, and <inner, p_eval>.
  const size_t enc_words = (size_t)(n_deg + 1) * n_cols * L, canon_words = (size_t)(n_deg + 1) * n_per_row * L;
  std::lock_guard<std::mutex> arena_lock(c->verify_mu);
  {
    const size_t need = (enc_words + canon_words) * 8 + 256;
    if (c->h_varena_cap < need) {
      if (hipSetDevice(c->prm.device) != hipSuccess) return LCPC_ERR_HIP;
      if (c->h_varena) (void)hipHostFree(c->h_varena);
      c->h_varena = nullptr; c->h_varena_cap = 0;
      void* hp = nullptr;
      if (hipHostMalloc(&hp, need + need / 4, hipHostMallocDefault) != hipSuccess) return LCPC_ERR_NOMEM;
      c->h_varena = static_cast<uint8_t*>(hp); c->h_varena_cap = need + need / 4;
    }
  }
  uint64_t* const enc = reinterpret_cast<uint64_t*>(c->h_varena);
  uint64_t* const vcanon = enc + ((enc_words + 31) & ~(size_t)31);
  std::vector<const uint64_t*> enc_msgs(n_deg + 1);
  for (uint64_t i = 0; i < n_deg; i++) enc_msgs[i] = p_random[i].data();
  enc_msgs[n_deg] = p_eval.data();
  std::vector<std::vector<uint64_t>> rand_tensors(n_deg, std::vector<uint64_t>(n_rows * L));
  std::vector<uint64_t> cols_to_open(n_columns);
  std::vector<uint64_t> dots((n_deg + 1) * n_columns * MAXL);                 // dots[d][i] = <tensor_d, column i>
  std::vector<uint8_t> leaf(n_columns * 32);
  uint64_t eval_acc[MAXL] = {0, 0, 0, 0};
  // to_repr of the polynomials is published in chunks: the transcript starts absorbing a polynomial as soon as its first
  // chunk is there instead of waiting for all of it.  1024 elements per chunk (50 us of conversion on one pool thread): with 8192 a
  // polynomial of <= 8192 coefficients was ONE chunk converted by one thread (0.4 ms) in front of its absorb -- verify at 2^19
  // 2.75 -> 1.05 ms, at 2^23 4.6 -> 3.7, at 2^26 14.2 -> 13.8 (the absorbs alone: 13.5)
  constexpr uint64_t CANON_CHUNK = 1024;
  const uint64_t n_cchunks = (n_per_row + CANON_CHUNK - 1) / CANON_CHUNK;
  std::unique_ptr<std::atomic<uint8_t>[]> canon_done(new std::atomic<uint8_t>[(n_deg + 1) * n_cchunks + 1]);
  for (uint64_t i = 0; i < (n_deg + 1) * n_cchunks; i++) canon_done[i].store(0, std::memory_order_relaxed);
  std::atomic<bool> tensors_ready{false}, cols_bad{false}, side_failed{false};
  int enc_rc = 0;
  double t_enc = 0, t_side = 0;
  tv[1] = now_ms();
  std::thread enc_thread, side_thread;
  JoinGuard join_enc{enc_thread}, join_side{side_thread};
  struct Release { std::atomic<bool>& flag; ~Release() { flag.store(true); } } release{tensors_ready};   // (runs before the joins on every exit path)
  enc_thread = std::thread([&] { const double t0 = now_ms(); enc_rc = encode_msgs_host(c, enc_msgs.data(), n_deg + 1, enc); t_enc = now_ms() - t0; });
  side_thread = std::thread([&] {
    try {
      const double t0 = now_ms();
      for (uint64_t pi_ = 0; pi_ <= n_deg; pi_++) {
        const uint64_t* src = pi_ < n_deg ? p_random[pi_].data() : p_eval.data();
        uint64_t* dst = &vcanon[pi_ * n_per_row * L];
        parallel_for(n_cchunks, 1, [&](uint64_t b, uint64_t e) {          // chunks are claimed in increasing order
          for (uint64_t ch = b; ch < e; ch++) {
            const uint64_t i1 = std::min(n_per_row, (ch + 1) * CANON_CHUNK);
            for (uint64_t i = ch * CANON_CHUNK; i < i1; i++) h_canon(f, dst + i * L, src + i * L);
            canon_done[pi_ * n_cchunks + ch].store(1, std::memory_order_release);
          }
        });
      }
      parallel_for(n_columns, 16, [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; i++)
          for (uint64_t k = 0; k < n_rows; k++)
            if (h_ge_p(f, cols[i].data() + k * L)) { cols_bad.store(true); return; }
      });
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
      for (uint64_t blk = 0; blk < n_blk; blk++) h_add(f, eval_acc, eval_acc, &part[blk * MAXL]);
      while (!tensors_ready.load(std::memory_order_acquire)) std::this_thread::yield();
      if (cols_bad.load()) return;
      parallel_for(n_columns, 4, [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; i++) {
          for (uint64_t d = 0; d <= n_deg; d++) {
            const uint64_t* tensor = d < n_deg ? rand_tensors[d].data() : outer;
            uint64_t acc[MAXL] = {0, 0, 0, 0}, t[MAXL];
            for (uint64_t k = 0; k < n_rows; k++) { h_mul(f, t, tensor + k * L, cols[i].data() + k * L); h_add(f, acc, acc, t); }
            memcpy(&dots[(d * n_columns + i) * MAXL], acc, F);
          }
          hash_column_host(f, cols[i].data(), n_rows, &leaf[i * 32]);
        }
      }, n_columns * n_rows > ((uint64_t)1 << 19) ? 64u : 15u);       // Brakedown opens 6593 columns: ~0.1 s of work single-threaded
      t_side = now_ms() - t0;
    } catch (...) {
      side_failed.store(true);
      for (uint64_t i = 0; i < (n_deg + 1) * n_cchunks; i++) canon_done[i].store(1);   // never leave the transcript thread waiting
    }
  });
  // step 1: random tensors, transcript
  auto absorb_poly = [&](uint64_t pi_, const uint8_t* label) {             // absorb_canon, chunk by chunk as to_repr delivers
    for (uint64_t ch = 0; ch < n_cchunks; ch++) {
      while (!canon_done[pi_ * n_cchunks + ch].load(std::memory_order_acquire)) std::this_thread::yield();
      const uint64_t i0 = ch * CANON_CHUNK, i1 = std::min(n_per_row, i0 + CANON_CHUNK);
      absorb_canon(tr, label, f, &vcanon[(pi_ * n_per_row + i0) * L], i1 - i0);
    }
  };
  for (uint64_t i = 0; i < n_deg; i++) {
    uint8_t key[32];
    tr.challenge_bytes(LBL_DT, 6, key, 32);
    ChaCha20Rng rng(key);
    for (uint64_t k = 0; k < n_rows; k++) rng.field_random(f, &rand_tensors[i][k * L]);
    if (i + 1 == n_deg) tensors_ready.store(true, std::memory_order_release);
    absorb_poly(i, LBL_PR);
  }
  tensors_ready.store(true, std::memory_order_release);
  absorb_poly(n_deg, LBL_PE);
  uint8_t key[32];
  tr.challenge_bytes(LBL_CO, 6, key, 32);
  ChaCha20Rng rng(key);
  for (auto& x : cols_to_open) x = rng.uniform(n_cols);
  tv[2] = now_ms();
  enc_thread.join();
  side_thread.join();
  tv[3] = now_ms();
  if (side_failed.load()) return LCPC_ERR_NOMEM;
  if (cols_bad.load()) return LCPC_VERR_MALFORMED;
  if (enc_rc) return enc_rc == LCPC_ERR_ENCODE ? LCPC_VERR_ENCODE : enc_rc;
  // step 3, the part that needs the drawn column numbers: compare with the encoded rows, fold the Merkle paths.  The error
  // reported is that of the first failing column, with the reference's precedence degree > eval > path
  // (columns are independent: folded on all cores -- Brakedown opens 6593 of them, 13-21 compressions each)
  std::atomic<uint64_t> first_bad(n_columns);
  std::unique_ptr<uint8_t[]> col_status(new uint8_t[n_columns ? n_columns : 1]);
  parallel_for(n_columns, 32, [&](uint64_t b, uint64_t e) {
    for (uint64_t i = b; i < e; i++) {
      if (i > first_bad.load(std::memory_order_relaxed)) { col_status[i] = 0; continue; }    // an earlier column already failed
      const uint64_t cn = cols_to_open[i];
      bool rnd = true, evl = true;
      for (uint64_t d = 0; d <= n_deg; d++) {
        const bool ok = h_eq(f, &dots[(d * n_columns + i) * MAXL], &enc[(d * n_cols + cn) * L]);   // verify_column_value lib.rs:985-1000
        if (d < n_deg) rnd = rnd && ok; else evl = ok;
      }
      uint8_t h[32], blk[64];                                                    // verify_column_path lib.rs:955-982
      memcpy(h, &leaf[i * 32], 32);
      uint64_t cc = cn;
      for (uint64_t k = 0; k < paths[i].n; k++) {
        const uint8_t* pk = paths[i].p + 40 * k + 8;
        if (cc % 2 == 0) { memcpy(blk, h, 32); memcpy(blk + 32, pk, 32); } else { memcpy(blk, pk, 32); memcpy(blk + 32, h, 32); }
        blake3_host(blk, 64, h);
        cc >>= 1;
      }
      const bool pth = memcmp(h, root, 32) == 0;
      const int stt = !rnd ? LCPC_VERR_COLUMN_DEGREE : (!evl ? LCPC_VERR_COLUMN_EVAL : (!pth ? LCPC_VERR_COLUMN_PATH : 0));
      col_status[i] = (uint8_t)(-stt);
      if (stt) {
        uint64_t cur = first_bad.load();
        while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
      }
    }
  });
  const int first_status = first_bad.load() < n_columns ? -(int)col_status[first_bad.load()] : 0;
  tv[4] = now_ms();
  if (first_status) return first_status;
  if (dbg)
    fprintf(stderr, "[lcpc_verify] parse + limb checks %.2f ms, transcript %.2f (beside it: row encodes on the GPU %.2f, side thread %.2f), wait for "
                    "the helpers %.2f, column compares + paths %.2f\n",
            tv[1] - tv[0], tv[2] - tv[1], t_enc, t_side, tv[3] - tv[2], tv[4] - tv[3]);
  memcpy(eval_out, eval_acc, F);
  return 0;
  LCPC_CATCH(c)
}

}  // extern "C"
