// lcpc_amd/csrc/ctx.cpp -- lcpc_ctx: the two LcEncoding implementors on the device.
//
// LigeroEncodingRho::{new, new_from_dims, encode, get_dims, dims_ok, ...} (lcpc-ligero-pc/src/lib.rs:45-185) and
// SdigEncodingS::{new, new_from_dims, encode, ...} (lcpc-brakedown-pc/src/lib.rs:54-176, encode.rs:36-110) of
// /root/reference.  A context is immutable after creation (the reference shares `&E` across Rayon workers,
// lcpc-2d/src/lib.rs:74-104); commitments made with it live in their own objects (commit.cpp).
// There is no CPU fallback: without a usable HIP device lcpc_ctx_create fails with LCPC_ERR_NO_DEVICE.
#include "internal.h"
#include <chrono>

using namespace lcpc;

namespace lcpc {


const uint8_t LBL_DT[7] = "$l//DT", LBL_PR[7] = "$l//PR", LBL_PE[7] = "$l//PE", LBL_CO[7] = "$l//CO";  // macros.rs:31-34

void ctx_ref(lcpc_ctx* c) { c->refs.fetch_add(1); }

static void ctx_free(lcpc_ctx* c) {
  (void)hipSetDevice(c->prm.device);
  comm_release(c);
  dev_free(c->d_wq_w);
  dev_free(c->d_pack[0]); dev_free(c->d_pack[1]); dev_free(c->d_pack[2]); dev_free(c->d_roots29s); dev_free(c->d_roots29cs); dev_free(c->d_rootsls); dev_free(c->d_rootslcs);
  dev_free(c->d_rootsl); dev_free(c->d_rootslc); dev_free(c->d_qpl); dev_free(c->d_roots); dev_free(c->d_roots29); dev_free(c->d_roots29c); dev_free(c->d_qp29); dev_free(c->d_r2);
  dev_free(c->ws.d_tmp); dev_free(c->ws.d_t); dev_free(c->ws.d_mid); dev_free(c->d_scratch);
  if (c->h_varena) (void)hipHostFree(c->h_varena);
  for (unsigned k = 0; k < lcpc_ctx::N_STAGE; k++) {
    if (c->h_stage[k]) (void)hipHostFree(c->h_stage[k]);
    if (c->ev_stage[k]) (void)hipEventDestroy(c->ev_stage[k]);
  }
  for (auto* v : {&c->d_pre, &c->d_post})
    for (auto& d : *v) { dev_free(d.rowptr); dev_free(d.colidx); dev_free(d.vals); dev_free(d.vals29); }
  delete c;
}
void ctx_unref(lcpc_ctx* c) {
  if (c->refs.fetch_sub(1) == 1) ctx_free(c);
}

// Rows per batch of the 29-bit-limb intermediate between the two K1s passes (0 = keep the packed intermediate in comm).
// Measured (tools/ab_mid_shapes.py, profiles/r03_mid_shapes.jsonl): the limb format saves ~3 % of the VALU instructions
// (no clamp / pack at the first pass's store, no unpack at the last pass's load) and 1.5 % of the time while the first
// pass stores runs of >= 32 elements (n_cols <= 2^15); with shorter runs its three planes become partial-line writes
// (64 + 64 + 16 bytes per run at 2^18 columns) and it LOSES 1-2 %.  Hence: on by default only for n_cols <= 2^15;
// LCPC_NTT_MID_MAX_MB=<MiB> forces it for every shape within that budget (A/B, tests), =0 turns it off.
uint64_t ntt_mid_rows(const lcpc_ctx* c, uint64_t n_rows) {
  const bool ev = c->sw_ntt_mid_max_mb >= 0;                    // (read when the context was created)
  if (!c->l9s || n_rows == 0) return 0;
  if (!ev && c->log_n > 15) return 0;
  const uint64_t max_mb = ev ? (uint64_t)c->sw_ntt_mid_max_mb : 6144;
  if (max_mb == 0) return 0;
  const uint64_t per_row = c->n_cols * 36;
  uint64_t fit = (max_mb << 20) / per_row;
  if (fit == 0) return 0;
  if (fit >= n_rows) return n_rows;
  const uint64_t batches = (n_rows + fit - 1) / fit;       // equal batches
  return (n_rows + batches - 1) / batches;
}

// The N shifted multiples W_j = balanced(w 2^(W j) mod p), j < N, of a field element w (Montgomery form in), as the N^2 words
// t = N k + j = limb k of W_j (limbs 0..N-2 in [0, 2^W), the top limb signed) that field_wmul_gen.h's wmul_u* take as scalar operands
// (Ft255: N = 9, W = 29; field_ln.h's forms for the other fields).
static void wmul_table(const FieldDesc& f, const uint64_t* w_mont, int N, int W, uint32_t out[96]) {
  uint64_t v[MAXL];
  h_canon(f, v, w_mont);
  memset(out, 0, 96 * 4);
  for (int j = 0; j < N; j++) {
    // v > (p - 1) / 2 -> v - p, as a 320-bit two's complement number
    uint64_t m[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < f.L; i++) m[i] = v[i];
    bool big = false;
    for (int i = f.L - 1; i >= 0; i--) {
      const uint64_t h = (f.p[i] >> 1) | (i + 1 < f.L ? f.p[i + 1] << 63 : 0);
      if (v[i] != h) { big = v[i] > h; break; }
    }
    if (big) {
      unsigned __int128 br = 0;
      for (int i = 0; i < 5; i++) { const unsigned __int128 d = (unsigned __int128)m[i] - (i < f.L ? f.p[i] : 0) - (uint64_t)br; m[i] = (uint64_t)d; br = (d >> 64) & 1; }
    }
    for (int k = 0; k < N; k++) {
      const int b = W * k, wd = b / 64, sh = b % 64;
      uint64_t x = m[wd] >> sh;
      if (sh && wd + 1 < 5) x |= m[wd + 1] << (64 - sh);
      out[N * k + j] = k < N - 1 ? (uint32_t)(x & (((uint64_t)1 << W) - 1)) : (uint32_t)x;
    }
    for (int s = 0; s < W; s++) h_add(f, v, v, v);          // (h_add works on fully reduced values of either form)
  }
}
// w^(n/4) = ROOT_OF_UNITY^(2^(S - 2)): the primitive 4th root I of the field -- the last two stages of every transform multiply by it,
// and so does every radix-4 butterfly of the limb kernels (w1 = I w0: ntt_l9s.hip) -- as its shifted multiples on the device
static int build_wq_w(lcpc_ctx* c, int N, int W) {
  const FieldDesc* f = c->f;
  uint64_t i4[MAXL];
  memcpy(i4, f->rou, 8 * f->L);
  for (unsigned i = 0; i + 2 < f->S; i++) h_mul(*f, i4, i4, i4);
  uint32_t wt[96];
  wmul_table(*f, i4, N, W, wt);
  int rc = dev_alloc(&c->err, &c->d_wq_w, sizeof wt);
  if (rc) return rc;
  HIPCHK(c, hipMemcpy(c->d_wq_w, wt, sizeof wt, hipMemcpyHostToDevice));
  return 0;
}

// ---- NTT pass plan (DESIGN.md "K1") ----------------------------------------------------------------
static void plan_passes(lcpc_ctx* c) {
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
  if (c->d_qp29 && (k == 19 || k == 20) && !c->sw_ntt_general) {
    // Ft255, 2^19 / 2^20 columns (C4's shape): still two passes on 1024-element tiles for the shape-specialised kernel,
    // whose first pass then moves 64- / 32-byte runs -- affordable because the tiles that share those cache lines run back
    // to back on one XCD (ntt_tile_group below; measured: DESIGN.md section 4 K1s)
    c->passes.push_back({0, k - 10, 20 - k, 10});
    c->passes.push_back({k - 10, 10, 0u, 10});
    return;
  }
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

// first pass of the shape-specialised two-pass kernels: how many neighbouring tiles of a row (log2) run as consecutive
// workgroups of one XCD.  Its strided runs are run_bytes = F << log_tj long; grouping makes the span that one XCD works on at a
// time 1 KiB (2 KiB for runs of <= 32 bytes), which is what measured best (build-time A/B in one process, DESIGN.md K1s: n_cols
// 2^20 -27 % on boxes where the ungrouped order is slow, -4 % elsewhere; 2^19 -3 %; the headline's 128-byte runs -0.5 to -1 %).
static uint32_t ntt_tile_group_of(uint32_t log_n, int L, uint32_t log_tj) {
  const uint32_t tiles_log = log_n - 10;
  if (tiles_log < 3) return 0;
  uint32_t run_log = log_tj;                                              // log2(run bytes)
  for (uint32_t b = 8u * (uint32_t)L; b > 1; b >>= 1) run_log++;
  uint32_t lg = run_log <= 5 ? 11 - run_log : (run_log < 10 ? 10 - run_log : 0);
  return std::min(std::min(lg, 6u), tiles_log - 3);
}
static uint32_t ntt_tile_group(const lcpc_ctx* c, const Pass& first) { return ntt_tile_group_of(c->log_n, c->L, first.log_tj); }

#define ECHK(call)                                                        \
  do {                                                                    \
    hipError_t e__ = (call);                                              \
    if (e__ != hipSuccess) return fail_hip(err, e__, #call);              \
  } while (0)

// ---- encode rows: LcEncoding::encode, batched over rows -----------------------------------------------
int encode_rows_device(const lcpc_ctx* c, EncodeWs* ws, const EncodeJob& j, hipStream_t st, std::string* err, uint32_t* launches) {
  uint32_t dummy = 0;
  uint32_t& nl = launches ? *launches : dummy;
  if (j.kept_t) *j.kept_t = false;
  const uint64_t n_rows = j.n_rows;
  if (n_rows == 0) return 0;
  if (c->prm.encoding == LCPC_ENC_LIGERO && c->l9s) {
    // the shape-specialised two-pass kernel; between the passes the rows live as 29-bit limbs in ws->d_mid (row batches
    // sized by ntt_mid_rows), or -- if that buffer cannot be had -- packed in dst itself
    uint64_t rb = ws->mid_failed ? 0 : ntt_mid_rows(c, n_rows);
    if (rb) {
      std::string scratch_err;
      uint64_t want = rb * c->n_cols * 36;
#ifdef LCPC_TEST_HOOKS
      if (c->sw_test_fail_mid) want = (uint64_t)1 << 62;      // an allocation no device can satisfy, so that the real failure path runs
#endif
      if (ensure_dev(&scratch_err, &ws->d_mid, &ws->mid_cap, want)) {
        (void)hipGetLastError();           // HIP keeps a failed call's error until it is read: it must not surface at the next launch check
        ws->mid_failed = true; ws->mid_cap = 0; rb = 0;
      }
    }
    const uint64_t step = rb ? rb : n_rows;
    for (uint64_t r0 = 0; r0 < n_rows; r0 += step) {
      const uint64_t nr = std::min(step, n_rows - r0);
      for (int i = 0; i < 2; i++) {
        const Pass& p = c->passes[i];
        const bool first = i == 0;
        NttPassArgs a{};
        a.roots29c = j.canon_out ? c->d_roots29c : nullptr;
        // (canonical output: the last pass has a uniform round and converts what is left of block 0 before it -- no Montgomery-form
        // prefix reaches the store; a first pass of 8 to 10 stages has one too, and then the last pass sees canonical values only)
        a.mont_prefix = 0u;
        a.blk0_gone = (!first && j.canon_out && c->passes[0].s >= 8) ? 1u : 0u;
        a.dst = j.dst + r0 * c->n_cols * c->NL;
        a.src = first ? j.src + r0 * j.src_stride * c->NL : a.dst;
        a.roots = c->d_roots; a.roots29 = c->d_roots29; a.qp29 = c->d_qp29; a.wq_w = c->d_wq_w;
        a.src_stride = first ? j.src_stride : c->n_cols;
        a.dst_stride = c->n_cols;
        a.n_valid = first ? j.n_valid : c->n_cols;
        const uint64_t consumed = r0 * j.src_stride;
        a.n_src_total = !first || j.n_src_total == ~(uint64_t)0 ? ~(uint64_t)0 : (j.n_src_total > consumed ? j.n_src_total - consumed : 0);
        a.copy_dst = first && j.copy_dst ? j.copy_dst + r0 * j.src_stride * c->NL : nullptr;
        a.n_rows = nr;
        a.log_n = c->log_n; a.t0 = p.t0; a.s = p.s; a.log_tj = p.log_tj;
        a.mid = rb ? ws->d_mid : nullptr;
        a.tile_group = first ? ntt_tile_group(c, p) : 0u;
        ECHK(launch_ntt_pass_l9s(a, first, c->d_pack[i], c->pack_info[i], st));
        nl++;
      }
    }
    return 0;
  }
  if (c->prm.encoding == LCPC_ENC_LIGERO && c->l9s3) {
    // 2^21 .. 2^26 columns: s0 stages with the first-pass kernel over the whole rows (zero padding, ragged tail and the coeffs
    // copy live there), then every row is 2^s0 independent 2^20-point transforms: the two-pass plan on n_rows << s0 sub-rows, in
    // place, with the sub-sampled tables.  Canonical output: after the first pass only sub-row 0 of each row still holds
    // never-multiplied elements ("block 0"), so only those sub-rows take the converting twiddles / the prefix reduction
    const uint32_t s0 = c->log_n - 20;
    for (int i = 0; i < 3; i++) {
      const Pass& p = c->passes[i];
      const bool first = i == 0, sub = i > 0;
      NttPassArgs a{};
      a.dst = j.dst;
      a.src = first ? j.src : j.dst;
      a.roots = c->d_roots; a.qp29 = c->d_qp29; a.wq_w = c->d_wq_w;
      a.roots29 = sub ? c->d_roots29s : c->d_roots29;
      a.roots29c = j.canon_out ? (sub ? c->d_roots29cs : c->d_roots29c) : nullptr;
      a.canon_row_mask = sub ? (1u << s0) - 1 : 0u;
      a.mont_prefix = 0u;                                     // (as in the two-pass branch: pass 1 -- 10 stages -- has the uniform round)
      a.blk0_gone = (i == 2 && j.canon_out) ? 1u : 0u;
      a.src_stride = first ? j.src_stride : ((uint64_t)1 << 20);
      a.dst_stride = first ? c->n_cols : ((uint64_t)1 << 20);
      a.n_valid = first ? j.n_valid : ((uint64_t)1 << 20);
      a.n_src_total = first ? j.n_src_total : ~(uint64_t)0;
      a.copy_dst = first ? j.copy_dst : nullptr;
      a.n_rows = first ? n_rows : n_rows << s0;
      a.log_n = first ? c->log_n : 20u;
      a.t0 = i == 2 ? 10u : 0u; a.s = p.s; a.log_tj = p.log_tj;
      a.tile_group = i == 0 ? ntt_tile_group_of(c->log_n, c->L, p.log_tj) : (i == 1 ? ntt_tile_group_of(20, c->L, 0) : 0u);
      ECHK(launch_ntt_pass_l9s(a, i < 2, c->d_pack[i], c->pack_info[i], st));
      nl++;
    }
    return 0;
  }
  if (c->prm.encoding == LCPC_ENC_LIGERO && c->lns3) {        // the three-pass plan of the l9s3 branch above, on K1n
    const uint32_t s0 = c->log_n - 20;
    for (int i = 0; i < 3; i++) {
      const Pass& p = c->passes[i];
      const bool first = i == 0, sub = i > 0;
      NttPassArgs a{};
      a.dst = j.dst;
      a.src = first ? j.src : j.dst;
      a.roots = c->d_roots; a.qp29 = c->d_qpl; a.wq_w = c->d_wq_w;
      a.roots29 = sub ? c->d_rootsls : c->d_rootsl;
      a.roots29c = j.canon_out ? (sub ? c->d_rootslcs : c->d_rootslc) : nullptr;
      a.canon_row_mask = sub ? (1u << s0) - 1 : 0u;
      // (fields with a shifted-multiples multiply -- Ft127, Ft191 -- run uniform rounds in every 10-stage pass and convert block 0 before
      // them: pass 1 leaves nothing in Montgomery form; Ft63 keeps the 4-element prefix of the last pass)
      a.mont_prefix = (j.canon_out && i == 2 && c->NL == 2) ? 4u : 0u;
      a.blk0_gone = (i == 2 && j.canon_out && c->NL != 2) ? 1u : 0u;
      a.src_stride = first ? j.src_stride : ((uint64_t)1 << 20);
      a.dst_stride = first ? c->n_cols : ((uint64_t)1 << 20);
      a.n_valid = first ? j.n_valid : ((uint64_t)1 << 20);
      a.n_src_total = first ? j.n_src_total : ~(uint64_t)0;
      a.copy_dst = first ? j.copy_dst : nullptr;
      a.n_rows = first ? n_rows : n_rows << s0;
      a.log_n = first ? c->log_n : 20u;
      a.t0 = i == 2 ? 10u : 0u; a.s = p.s; a.log_tj = p.log_tj;
      a.tile_group = i == 0 ? ntt_tile_group_of(c->log_n, c->L, p.log_tj) : (i == 1 ? ntt_tile_group_of(20, c->L, 0) : 0u);
      ECHK(launch_ntt_pass_lns(c->NL, a, i < 2, c->d_pack[i], c->pack_info[i], st));
      nl++;
    }
    return 0;
  }
  if (c->prm.encoding == LCPC_ENC_LIGERO && c->lns) {
    for (int i = 0; i < 2; i++) {
      const Pass& p = c->passes[i];
      const bool first = i == 0;
      NttPassArgs a{};
      a.src = first ? j.src : j.dst;
      a.dst = j.dst;
      a.roots = c->d_roots; a.roots29 = c->d_rootsl; a.qp29 = c->d_qpl; a.wq_w = c->d_wq_w;
      a.roots29c = j.canon_out ? c->d_rootslc : nullptr;
      // the last pass ends with a radix-4 round (10 stages): Ft63 reduces its 4-element never-multiplied prefix at the store; Ft127 /
      // Ft191 have a uniform round in that pass and convert block 0 before it (ntt_lns.hip), or already in a first pass of >= 8 stages
      a.mont_prefix = (j.canon_out && !first && c->NL == 2) ? 4u : 0u;
      a.blk0_gone = (!first && j.canon_out && c->NL != 2 && c->passes[0].s >= 8) ? 1u : 0u;
      a.src_stride = first ? j.src_stride : c->n_cols;
      a.dst_stride = c->n_cols;
      a.n_valid = first ? j.n_valid : c->n_cols;
      a.n_src_total = first ? j.n_src_total : ~(uint64_t)0;
      a.copy_dst = first ? j.copy_dst : nullptr;
      a.n_rows = n_rows;
      a.log_n = c->log_n; a.t0 = p.t0; a.s = p.s; a.log_tj = p.log_tj;
      a.tile_group = first ? ntt_tile_group(c, p) : 0u;
      ECHK(launch_ntt_pass_lns(c->NL, a, first, c->d_pack[i], c->pack_info[i], st));
      nl++;
    }
    return 0;
  }
  if (c->prm.encoding == LCPC_ENC_LIGERO) {
    bool first = true;
    for (const Pass& p : c->passes) {
      NttPassArgs a;
      a.roots29c = j.canon_out ? c->d_roots29c : nullptr;
      // the trailing stages multiply by 1 only: a final radix-4 round leaves 4 elements per row unconverted, a final
      // radix-2 stage 2 (ntt_pass_l9_kernel)
      a.mont_prefix = (j.canon_out && p.t0 + p.s == c->log_n) ? (c->log_n == 0 ? 1u : (p.s % 2 == 0 ? 4u : 2u)) : 0u;
      a.src = first ? j.src : j.dst;
      a.dst = j.dst;
      a.roots = c->d_roots;
      a.roots29 = c->d_roots29;
      a.qp29 = c->d_qp29;
      a.wq_w = c->d_wq_w;
      a.src_stride = first ? j.src_stride : c->n_cols;
      a.dst_stride = c->n_cols;
      a.n_valid = first ? j.n_valid : c->n_cols;
      a.n_src_total = first ? j.n_src_total : ~(uint64_t)0;
      a.copy_dst = first ? j.copy_dst : nullptr;
      a.n_rows = n_rows;
      a.log_n = c->log_n; a.t0 = p.t0; a.s = p.s; a.log_tj = p.log_tj;
      ECHK(launch_ntt_pass(c->NL, p.log_tile, a, st));
      nl++;
      first = false;
    }
    return 0;
  }
  // Brakedown: systematic part, then precodes down, R-S base case, postcodes up (encode.rs:36-94)
  const size_t t = c->d_pre.size();
  const DevCsr& pl = c->d_pre[t - 1];
  const size_t eb = elem_bytes(c);
  // (capacities in BYTES: ensure_dev rounds to 256, and a capacity kept in elements loses the remainder for 24-byte elements --
  // Ft191 then re-allocated, i.e. hipFree-synchronised, on every call)
  if (int rc = ensure_dev(err, &ws->d_tmp, &ws->tmp_cap, n_rows * pl.n_out * eb)) return rc;
  if (n_rows >= SDIG_T_MIN_ROWS) {
    // fast path: position-major working copy T[pos][row] (lane = row: contiguous gathers, wave-uniform matrix)
    if (int rc = ensure_dev(err, &ws->d_t, &ws->t_cap, n_rows * c->n_cols * eb)) return rc;
    ECHK(launch_transpose_to_t(c->NL, j.src, j.src_stride, j.n_valid, n_rows, ws->d_t, st, j.n_src_total, j.copy_dst, j.canon_out && j.keep_t));
    nl++;
    uint64_t in_start = 0;
    SpmmTArgs a{};
    a.t = ws->d_t; a.n_rows = n_rows;
    auto set_mat = [&](const DevCsr& m) { a.rowptr = m.rowptr; a.colidx = m.colidx; a.vals = m.vals; a.vals29 = m.vals29; a.m = m.n_out; };
    for (size_t i = 0; i + 1 < t; i++) {
      const uint64_t in_end = in_start + c->d_pre[i].n_in;
      a.out_alt = nullptr; a.in_off = in_start; a.out_off = in_end;
      set_mat(c->d_pre[i]);
      ECHK(launch_spmm_t(c->NL, a, st));
      nl++;
      in_start = in_end;
    }
    const uint64_t in_end = in_start + pl.n_in;
    a.out_alt = ws->d_tmp; a.in_off = in_start; a.out_off = 0;
    set_mat(pl);
    ECHK(launch_spmm_t(c->NL, a, st));
    const uint64_t out_end = in_end + c->d_post[t - 1].n_in;
    ECHK(launch_sdig_rs_t(c->NL, ws->d_tmp, (uint32_t)pl.n_out, ws->d_t, in_end, (uint32_t)c->d_post[t - 1].n_in, n_rows, c->d_r2, st));
    nl += 2;
    in_start = in_end + pl.n_out;
    uint64_t out_start = out_end;
    for (size_t ii = t; ii-- > 0;) {
      in_start -= c->d_pre[ii].n_out;
      a.out_alt = nullptr; a.in_off = in_start; a.out_off = out_start;
      set_mat(c->d_post[ii]);
      ECHK(launch_spmm_t(c->NL, a, st));
      nl++;
      out_start += c->d_post[ii].n_out;
    }
    if (j.keep_t) {               // commit: the position-major copy IS the commitment (hash_columns / open_column read it)
      if (j.kept_t) *j.kept_t = true;
      return 0;
    }
    ECHK(launch_transpose_from_t(c->NL, ws->d_t, c->n_cols, n_rows, j.dst, c->n_cols, st));
    nl++;
    return 0;
  }
  // few rows (the verifier's 1 + n_degree_tests single-row encodes): lane = output on the row-major rows
  if (j.src != j.dst || j.src_stride != c->n_cols) {
    ECHK(launch_pad_rows(c->NL, j.src, j.src_stride, j.dst, c->n_cols, j.n_valid, n_rows, st));
    nl++;
  }
  uint64_t in_start = 0;
  SpmvArgs a{};
  a.mat = j.dst; a.stride = c->n_cols; a.n_rows = n_rows;
  for (size_t i = 0; i + 1 < t; i++) {
    const uint64_t in_end = in_start + c->d_pre[i].n_in;
    a.out_alt = nullptr; a.in_off = in_start; a.out_off = in_end;
    a.rowptr = c->d_pre[i].rowptr; a.colidx = c->d_pre[i].colidx; a.vals = c->d_pre[i].vals; a.vals29 = c->d_pre[i].vals29; a.m = c->d_pre[i].n_out;
    ECHK(launch_spmv(c->NL, a, st));
    nl++;
    in_start = in_end;
  }
  const uint64_t in_end = in_start + pl.n_in;
  a.out_alt = ws->d_tmp; a.out_alt_stride = pl.n_out; a.in_off = in_start; a.out_off = 0;
  a.rowptr = pl.rowptr; a.colidx = pl.colidx; a.vals = pl.vals; a.vals29 = pl.vals29; a.m = pl.n_out;
  ECHK(launch_spmv(c->NL, a, st));
  const uint64_t out_end = in_end + c->d_post[t - 1].n_in;
  ECHK(launch_sdig_rs(c->NL, ws->d_tmp, pl.n_out, (uint32_t)pl.n_out, j.dst, c->n_cols, in_end,
                      (uint32_t)c->d_post[t - 1].n_in, n_rows, c->d_r2, st));
  nl += 2;
  in_start = in_end + pl.n_out;
  uint64_t out_start = out_end;
  for (size_t ii = t; ii-- > 0;) {
    in_start -= c->d_pre[ii].n_out;
    a.out_alt = nullptr; a.in_off = in_start; a.out_off = out_start;
    a.rowptr = c->d_post[ii].rowptr; a.colidx = c->d_post[ii].colidx; a.vals = c->d_post[ii].vals; a.vals29 = c->d_post[ii].vals29; a.m = c->d_post[ii].n_out;
    ECHK(launch_spmv(c->NL, a, st));
    nl++;
    out_start += c->d_post[ii].n_out;
  }
  return 0;
}

}  // namespace lcpc

// =====================================================================================================
// C ABI: encoder
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
    case LCPC_ERR_STATE: return "no commitment in this object";
    case LCPC_ERR_HIP: return "HIP runtime error";
    case LCPC_ERR_NOMEM: return "out of memory";
    case LCPC_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case LCPC_ERR_XCHG: return "all-gather of the sharded commit / prove failed";
    case LCPC_ERR_NO_RCCL: return "librccl could not be loaded";
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
  LCPC_TRY
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
  LCPC_CATCH((lcpc_ctx*)nullptr)
}

// new_ml (ligero lib.rs:128-135, brakedown lib.rs:114-123): dims for a multilinear polynomial in n_vars variables
int lcpc_static_get_dims_ml(const lcpc_params* p, uint32_t n_vars, uint64_t* nr, uint64_t* np, uint64_t* nc) {
  if (!p || !nr || !np || !nc || n_vars >= 63) return LCPC_ERR_ARG;
  LCPC_TRY
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
  LCPC_CATCH((lcpc_ctx*)nullptr)
}

static int ctx_build(lcpc_ctx* c, const lcpc_params* p) {
  const FieldDesc* f = c->f;
  std::string* err = &c->err;
  int rc = 0;
  if (p->encoding == LCPC_ENC_LIGERO) {
    if (p->rho_num == 0 || p->rho_num >= p->rho_den) return LCPC_ERR_ARG;
    uint64_t nr, np_, nc;
    if (p->n_per_row && p->n_cols) { np_ = p->n_per_row; nc = p->n_cols; }      // new_from_dims (ligero lib.rs:138-148)
    else if ((rc = lcpc_static_get_dims(p, &nr, &np_, &nc))) return rc;
    if (!(np_ < nc) || (nc & (nc - 1)) || log2_ceil(nc) > f->S) return LCPC_ERR_DIMS;   // _dims_ok + precomp_fft
    if (log2_ceil(nc) > 30) return LCPC_ERR_TOO_BIG;      // device kernels index a row with 32 bits
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
      if ((rc = dev_alloc(err, &d_pw, pw.size() * 8)) || (rc = dev_alloc(err, &d_one, 8 * f->L)) ||
          (rc = dev_alloc(err, &c->d_roots, n_roots * 8 * f->L)) ||
          (f->L == 4 && (rc = dev_alloc(err, &c->d_roots29, n_roots * 48))) ||
          (f->L == 4 && (rc = dev_alloc(err, &c->d_roots29c, n_roots * 48)))) {
        dev_free(d_pw); dev_free(d_one);
        return rc;
      }
      hipError_t he = hipMemcpy(d_pw, pw.data(), pw.size() * 8, hipMemcpyHostToDevice);
      if (he == hipSuccess) he = hipMemcpy(d_one, f->r, 8 * f->L, hipMemcpyHostToDevice);
      if (he == hipSuccess) he = launch_roots(c->NL, d_pw, log_half, d_one, c->d_roots, c->d_roots29, c->d_roots29c, nullptr);
      if (he == hipSuccess) he = hipDeviceSynchronize();
      dev_free(d_pw); dev_free(d_one);
      if (he != hipSuccess) return fail_hip(err, he, "precomp_fft");
    }
    if (f->L == 4) {
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
      if ((rc = dev_alloc(err, &c->d_qp29, tab.size() * 4))) return rc;
      HIPCHK(c, hipMemcpy(c->d_qp29, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
      if ((rc = build_wq_w(c, 9, 29))) return rc;
      c->comm_canon = true;
    }
    if ((rc = dev_alloc(err, &c->d_r2, 8 * f->L))) return rc;
    HIPCHK(c, hipMemcpy(c->d_r2, f->r2, 8 * f->L, hipMemcpyHostToDevice));
    plan_passes(c);
    if (c->d_qp29 && !c->sw_ntt_general && ntt_l9s_supported(c->log_n, (uint32_t)c->passes.size(), c->passes[0].log_tile)) {
      // two passes on 1024-element tiles: the shape-specialised kernel with its lane-order twiddle packs
      for (int i = 0; i < 2; i++) {
        const Pass& ps = c->passes[i];
        const bool first = i == 0;
        NttPassArgs a{};
        a.roots29 = c->d_roots29; a.roots29c = c->d_roots29c; a.log_n = c->log_n; a.t0 = ps.t0; a.s = ps.s; a.log_tj = ps.log_tj;
        c->pack_info[i] = ntt_l9s_pack_info(ps.s, first);
        const uint32_t n_classes = first ? 1u << (c->log_n - 10) : 1u;
        if ((rc = dev_alloc(err, &c->d_pack[i], (size_t)n_classes * c->pack_info[i].class_words * 4))) return rc;
        HIPCHK(c, launch_ntt_l9s_pack(a, first, c->pack_info[i], n_classes, c->d_pack[i], nullptr));
      }
      HIPCHK(c, hipDeviceSynchronize());
      c->l9s = true;
    } else if (c->d_qp29 && !c->sw_ntt_general && ntt_l9s3_supported(c->log_n)) {
      // three passes of the shape-specialised kernel (kernels.h ntt_l9s3_supported): tables for the 2^20-point sub-transforms,
      // pack 0 for the first pass over the whole rows (one class per tile position: 2^(log_n - 10)), packs 1 / 2 = those of a
      // 2^20-column context
      const unsigned k = c->log_n, s0 = k - 20;
      const size_t n_sub = (size_t)1 << 19;
      // the first pack is ~2.3 x one row (4.9 GB at 2^26 columns): if the device cannot hold the plan's tables, the general kernel's
      // three passes (tables already built above) take the rows instead -- slower, not an error
      const std::vector<Pass> general_plan = c->passes;
      auto build = [&]() -> int {
        int r;
        if ((r = dev_alloc(err, &c->d_roots29s, n_sub * 48)) || (r = dev_alloc(err, &c->d_roots29cs, n_sub * 48))) return r;
        HIPCHK(c, launch_ntt_l9s_subtable(c->d_roots29, s0, n_sub, c->d_roots29s, nullptr));
        HIPCHK(c, launch_ntt_l9s_subtable(c->d_roots29c, s0, n_sub, c->d_roots29cs, nullptr));
        c->passes.clear();
        c->passes.push_back({0, s0, 10 - s0, 10});
        c->passes.push_back({s0, 10, 0u, 10});
        c->passes.push_back({s0 + 10, 10, 0u, 10});
        for (int i = 0; i < 3; i++) {
          const Pass& ps = c->passes[i];
          const bool first = i < 2;                            // passes 0 and 1 run the first-pass kernel
          NttPassArgs a{};
          a.roots29 = i == 0 ? c->d_roots29 : c->d_roots29s; a.roots29c = i == 0 ? c->d_roots29c : c->d_roots29cs;
          a.log_n = i == 0 ? k : 20u; a.t0 = i == 2 ? 10u : 0u; a.s = ps.s; a.log_tj = ps.log_tj;
          c->pack_info[i] = ntt_l9s_pack_info(ps.s, first);
          const uint32_t n_classes = i == 0 ? 1u << (k - 10) : (i == 1 ? 1024u : 1u);
#ifdef LCPC_TEST_HOOKS
          if (i == 0 && c->sw_test_fail_3pass) return LCPC_ERR_NOMEM;       // (the fallback below)
#endif
          if ((r = dev_alloc(err, &c->d_pack[i], (size_t)n_classes * c->pack_info[i].class_words * 4))) return r;
          HIPCHK(c, launch_ntt_l9s_pack(a, first, c->pack_info[i], n_classes, c->d_pack[i], nullptr));
        }
        HIPCHK(c, hipDeviceSynchronize());
        return 0;
      };
      const int brc = build();
      if (brc == 0) {
        c->l9s3 = true;
      } else if (brc == LCPC_ERR_NOMEM) {
        for (auto& pk : c->d_pack) { dev_free(pk); pk = nullptr; }
        dev_free(c->d_roots29s); dev_free(c->d_roots29cs);
        c->d_roots29s = c->d_roots29cs = nullptr;
        c->passes = general_plan;
        (void)hipGetLastError();                               // the failed hipMalloc must not surface at the next launch check
        err->clear();
      } else {
        return brc;
      }
    }
    const bool lns3 = !c->d_qp29 && !c->sw_ntt_general && ntt_lns3_supported(c->NL, c->log_n);
    if (!c->d_qp29 && c->passes.size() >= 2 && !c->sw_ntt_general && (ntt_lns_supported(c->NL, c->log_n) || lns3)) {
      // Ft63 / Ft127 / Ft191 rows that need more than one pass: two passes on 1024-element tiles with the lazy-limb
      // kernel (ntt_lns.hip), its twiddle table in limb form (w^i R' mod p), the clamp table and the lane-order packs
      const std::vector<Pass> general_plan = c->passes;      // (if the tables do not fit the device: the general kernel, as for l9s3)
      auto build = [&]() -> int {
        int rc_ = 0;
        const unsigned k = c->log_n;
        c->passes.clear();
        if (lns3) {                                              // three passes: s0 stages over the whole rows, then 10 + 10 per 2^20-element block
          c->passes.push_back({0, k - 20, 30 - k, 10});
          c->passes.push_back({k - 20, 10, 0u, 10});
          c->passes.push_back({k - 10, 10, 0u, 10});
        } else {
          c->passes.push_back({0, k - 10, 20 - k, 10});
          c->passes.push_back({k - 10, 10, 0u, 10});
        }
        const int N = ntt_lns_limbs(c->NL), W = ntt_lns_limb_bits(c->NL), stride = ntt_lns_stride(c->NL);
        uint64_t rp[MAXL] = {1, 0, 0, 0};                        // R' = 2^(N W) mod p, a plain integer
        for (int i = 0; i < N * W; i++) h_add(*f, rp, rp, rp);
        std::vector<uint32_t> tab((size_t)64 * stride, 0);
        for (int i = 0; i < 64; i++) {                           // (i - 24) * p as normalised signed limbs (two's complement top limb)
          const int q = i - 24;
          uint64_t mag[5] = {0, 0, 0, 0, 0};
          unsigned __int128 cy = 0;
          for (int w = 0; w < 5; w++) { cy += (unsigned __int128)(w < f->L ? f->p[w] : 0) * (uint64_t)(q < 0 ? -q : q); mag[w] = (uint64_t)cy; cy >>= 64; }
          if (q < 0) {                                           // two's complement over 320 bits
            unsigned __int128 c2 = 1;
            for (int w = 0; w < 5; w++) { c2 += (unsigned __int128)(~mag[w]); mag[w] = (uint64_t)c2; c2 >>= 64; }
          }
          for (int l = 0; l < N; l++) {
            const int b = W * l, w = b / 64, sh = b % 64;
            uint64_t x = mag[w] >> sh;
            if (sh) x |= mag[w + 1] << (64 - sh);
            tab[(size_t)i * stride + l] = l + 1 < N ? (uint32_t)(x & (((uint64_t)1 << W) - 1)) : (uint32_t)x;   // top limb: sign-extended
          }
        }
        const size_t n_roots = (size_t)1 << (k - 1);
        uint64_t rpc[MAXL];
        h_canon(*f, rpc, rp);                                    // R' R^-1 mod p: the converting table is w^i R' R^-1 = mont_mul(w^i R, R' R^-1)
        uint32_t *d_rp = nullptr, *d_rpc = nullptr;
        if ((rc_ = dev_alloc(err, &d_rp, 8 * f->L))) return rc_;
        if ((rc_ = dev_alloc(err, &d_rpc, 8 * f->L))) { dev_free(d_rp); return rc_; }
        if ((rc_ = dev_alloc(err, &c->d_rootsl, n_roots * stride * 4)) || (rc_ = dev_alloc(err, &c->d_rootslc, n_roots * stride * 4)) ||
            (rc_ = dev_alloc(err, &c->d_qpl, tab.size() * 4))) { dev_free(d_rp); dev_free(d_rpc); return rc_; }
        hipError_t he = hipMemcpy(d_rp, rp, 8 * f->L, hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMemcpy(d_rpc, rpc, 8 * f->L, hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMemcpy(c->d_qpl, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
        if (he == hipSuccess) he = launch_ntt_lns_roots(c->NL, c->d_roots, n_roots, d_rp, c->d_rootsl, nullptr);
        if (he == hipSuccess) he = launch_ntt_lns_roots(c->NL, c->d_roots, n_roots, d_rpc, c->d_rootslc, nullptr);
        if (he == hipSuccess) he = hipDeviceSynchronize();
        dev_free(d_rp); dev_free(d_rpc);
        if (he != hipSuccess) return fail_hip(err, he, "ntt_lns tables");
        if (!c->d_wq_w && (rc_ = build_wq_w(c, N, W))) return rc_;
        if (lns3) {
          const size_t n_sub = (size_t)1 << 19;
          if ((rc_ = dev_alloc(err, &c->d_rootsls, n_sub * stride * 4)) || (rc_ = dev_alloc(err, &c->d_rootslcs, n_sub * stride * 4))) return rc_;
          HIPCHK(c, launch_ntt_lns_subtable(c->NL, c->d_rootsl, k - 20, n_sub, c->d_rootsls, nullptr));
          HIPCHK(c, launch_ntt_lns_subtable(c->NL, c->d_rootslc, k - 20, n_sub, c->d_rootslcs, nullptr));
        }
        for (int i = 0; i < (lns3 ? 3 : 2); i++) {
          const Pass& ps = c->passes[i];
          const bool first = lns3 ? i < 2 : i == 0;              // three-pass plans: passes 0 and 1 run the first-pass kernel
          const bool sub = lns3 && i > 0;
          NttPassArgs a{};
          a.roots29 = sub ? c->d_rootsls : c->d_rootsl; a.roots29c = sub ? c->d_rootslcs : c->d_rootslc;
          a.log_n = sub ? 20u : k; a.t0 = sub ? (i == 2 ? 10u : 0u) : ps.t0; a.s = ps.s; a.log_tj = ps.log_tj;
          c->pack_info[i] = ntt_lns_pack_info(c->NL, ps.s, first);
          const uint32_t n_classes = !first ? 1u : (sub ? 1024u : 1u << (k - 10));
#ifdef LCPC_TEST_HOOKS
          if (lns3 && i == 0 && c->sw_test_fail_3pass) return LCPC_ERR_NOMEM;   // (the fallback below)
#endif
          if ((rc_ = dev_alloc(err, &c->d_pack[i], (size_t)n_classes * c->pack_info[i].class_words * 4))) return rc_;
          HIPCHK(c, launch_ntt_lns_pack(c->NL, a, first, c->pack_info[i], n_classes, c->d_pack[i], nullptr));
        }
        HIPCHK(c, hipDeviceSynchronize());
        return 0;
      };
      const int brc = build();
      if (brc == LCPC_ERR_NOMEM) {
        for (auto& pk : c->d_pack) { dev_free(pk); pk = nullptr; }
        dev_free(c->d_rootsl); dev_free(c->d_rootslc); dev_free(c->d_qpl); dev_free(c->d_rootsls); dev_free(c->d_rootslcs); dev_free(c->d_wq_w);
        c->d_rootsl = c->d_rootslc = c->d_qpl = c->d_rootsls = c->d_rootslcs = c->d_wq_w = nullptr;
        c->passes = general_plan;
        (void)hipGetLastError();
        err->clear();
        return 0;
      }
      if (brc) return brc;
      c->lns = !lns3;
      c->lns3 = lns3;
      c->comm_canon = true;                                    // commits keep comm canonical on the device, as for Ft255
    }
    return 0;
  }
  if (p->encoding != LCPC_ENC_SDIG) return LCPC_ERR_ARG;
  if (!sdig_spec((int)c->prm.sdig_code, &c->spec)) return LCPC_ERR_ARG;
  uint64_t npr = p->n_per_row;
  if (!(p->n_per_row && p->n_cols)) {
    if (!sdig_n_per_row(*f, p->n_coeffs, (int)c->prm.sdig_code, &npr)) return LCPC_ERR_ARG;
  }
  std::vector<CsrMatrix> pre, post;
  const bool dbg = c->sw_debug_timing;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_gen0 = now();
  if (!sdig_generate(*f, c->spec, npr, p->seed, pre, post, c->pre_dims, c->post_dims)) return LCPC_ERR_DIMS;
  const double t_gen1 = now();
  c->n_per_row = npr;
  c->n_cols = sdig_codeword_length(c->pre_dims, c->post_dims);
  if (p->n_per_row && p->n_cols && p->n_cols != c->n_cols) return LCPC_ERR_DIMS;   // new_from_dims assert
  uint32_t* d_rprime = nullptr;          // R' = 2^(N W) mod p as a plain integer (Ft127 / Ft191 limb form)
  if (f->L == 2 || f->L == 3) {
    uint64_t rp[MAXL] = {1, 0, 0, 0};
    for (int i = 0; i < ntt_lns_limbs(c->NL) * ntt_lns_limb_bits(c->NL); i++) h_add(*f, rp, rp, rp);
    if ((rc = dev_alloc(err, &d_rprime, 8 * f->L))) return rc;
    HIPCHK(c, hipMemcpy(d_rprime, rp, 8 * f->L, hipMemcpyHostToDevice));
  }
  struct FreeRp { uint32_t*& p; ~FreeRp() { if (p) { (void)hipDeviceSynchronize(); dev_free(p); p = nullptr; } } } free_rp{d_rprime};
  auto upload = [&](const CsrMatrix& m, DevCsr& d) -> int {
    d.n_in = m.n_in; d.n_out = m.n_out;
    int r;
    if ((r = dev_alloc(err, &d.rowptr, m.rowptr.size() * 4))) return r;
    if ((r = dev_alloc(err, &d.colidx, m.colidx.size() * 4))) return r;
    if ((r = dev_alloc(err, &d.vals, m.vals.size() * 8))) return r;
    HIPCHK(c, hipMemcpy(d.rowptr, m.rowptr.data(), m.rowptr.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d.colidx, m.colidx.data(), m.colidx.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d.vals, m.vals.data(), m.vals.size() * 8, hipMemcpyHostToDevice));
    if (f->L == 4) {      // the 29-bit-limb / 2^261 form of the values (lazy29_mac) is derived on the device from the uploaded copy
      const size_t nnz = m.colidx.size();
      if ((r = dev_alloc(err, &d.vals29, (nnz + 1) * 48))) return r;
      HIPCHK(c, launch_to_r29(d.vals, nnz, d.vals29, nullptr));
      HIPCHK(c, hipMemsetAsync(d.vals29 + nnz * 12, 0, 48, nullptr));      // one entry of slack for the one-ahead prefetch
    } else if ((f->L == 2 || f->L == 3) && d_rprime) {
      // Ft127 / Ft191: values as 5 / 7 limbs of 29 bits in the R'-Montgomery form (ln::lazy_mac), v R' mod p = mont_mul(v R, R')
      const size_t nnz = m.colidx.size();
      const size_t stride = (size_t)ntt_lns_stride(c->NL);
      if ((r = dev_alloc(err, &d.vals29, (nnz + 1) * stride * 4))) return r;
      HIPCHK(c, launch_ntt_lns_roots(c->NL, d.vals, nnz, d_rprime, d.vals29, nullptr));
      HIPCHK(c, hipMemsetAsync(d.vals29 + nnz * stride, 0, stride * 4, nullptr));
    }
    return 0;
  };
  c->d_pre.resize(pre.size());
  c->d_post.resize(post.size());
  for (size_t i = 0; i < pre.size() && !rc; i++) { rc = upload(pre[i], c->d_pre[i]); if (!rc) rc = upload(post[i], c->d_post[i]); }
  if (!rc) rc = dev_alloc(err, &c->d_r2, 8 * f->L);
  if (rc) return rc;
  HIPCHK(c, hipMemcpy(c->d_r2, f->r2, 8 * f->L, hipMemcpyHostToDevice));
  if (dbg) fprintf(stderr, "[SdigEncoding::new] matgen %.1f ms, convert + upload %.1f ms\n", t_gen1 - t_gen0, now() - t_gen1);
  return 0;
}

int lcpc_ctx_create(const lcpc_params* p, lcpc_ctx** out) {
  if (!p || !out) return LCPC_ERR_ARG;
  *out = nullptr;
  lcpc_ctx* c = nullptr;
  LCPC_TRY
  const FieldDesc* f = field_desc((int)p->field);
  if (!f || p->hash != LCPC_HASH_BLAKE3) return LCPC_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device < 0 || p->device >= ndev) return LCPC_ERR_NO_DEVICE;
  if (hipSetDevice(p->device) != hipSuccess) return LCPC_ERR_NO_DEVICE;
  c = new lcpc_ctx();
  c->prm = *p;
  c->f = f; c->L = f->L; c->NL = 2 * f->L;
  if (c->prm.sdig_code == 0) c->prm.sdig_code = 3;
  // the switches: read here, once per context, never on a launch path
  c->sw_ntt_general = getenv("LCPC_NTT_GENERAL") != nullptr;
  if (const char* ev = getenv("LCPC_NTT_MID_MAX_MB")) c->sw_ntt_mid_max_mb = (int64_t)strtoull(ev, nullptr, 10);
  if (const char* ev = getenv("LCPC_HOST_STAGE")) c->sw_host_stage = (int32_t)strtol(ev, nullptr, 10);
  c->sw_debug_timing = getenv("LCPC_DEBUG_TIMING") != nullptr;
#ifdef LCPC_TEST_HOOKS   // only in lib/liblcpc_hip_testhooks.so (Makefile): the product neither reads the variable nor contains the branches
  if (const char* ev = getenv("LCPC_TEST_FAIL")) {           // test hook: "3pass", "mid" (comma-separated) -- forced allocation failures
    c->sw_test_fail_3pass = strstr(ev, "3pass") != nullptr;
    c->sw_test_fail_mid = strstr(ev, "mid") != nullptr;
  }
#endif
  int rc = 0;
  // (row shards begin where a BLAKE3 chunk boundary of the leaf message is also a row boundary: every chunk for Ft63 / Ft127 /
  // Ft255, every third chunk for Ft191 -- shard.cpp shard_chunk_range)
  if (c->prm.shard_count > 1 && c->prm.shard_rank >= c->prm.shard_count) rc = LCPC_ERR_ARG;
  if (!rc) rc = ctx_build(c, p);
  if (rc) { ctx_unref(c); return rc; }
  c->np2 = next_pow2(c->n_cols);
  c->path_len = (uint32_t)log2_ceil(c->n_cols);
  if (c->sw_host_stage > 0) {
    // LCPC_HOST_STAGE=1: every host source goes through the bounce ring -- pin its first two buffers now (~6 ms each), so that the
    // encoder's first lcpc_commit from pageable memory does not pay for them (the other two are made as they are first used; a failure
    // here is not an error: upload_host falls back to the runtime's own path for the slices that find no buffer)
    c->stage_cap = (size_t)64 << 20;
    for (unsigned k = 0; k < 2; k++) {
      if (hipHostMalloc(reinterpret_cast<void**>(&c->h_stage[k]), c->stage_cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->h_stage[k] = nullptr; break; }
      if (hipEventCreateWithFlags(&c->ev_stage[k], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(c->h_stage[k]); c->h_stage[k] = nullptr; c->ev_stage[k] = nullptr; break; }
    }
  }
  *out = c;
  return 0;
  } catch (...) {
    if (c) ctx_unref(c);
    return LCPC_ERR_NOMEM;
  }
}

void lcpc_ctx_destroy(lcpc_ctx* c) {
  if (c) ctx_unref(c);
}

int lcpc_get_dims(const lcpc_ctx* c, uint64_t len, uint64_t* nr, uint64_t* np, uint64_t* nc) {
  if (!c || len == 0) return LCPC_ERR_ARG;
  if (nr) *nr = (len + c->n_per_row - 1) / c->n_per_row;      // ligero lib.rs:166-169
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
  LCPC_TRY
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  const size_t bytes = (size_t)n_rows * c->n_cols * elem_bytes(c);
  int rc = ensure_dev(&c->err, &c->d_scratch, &c->scratch_cap, bytes);
  if (rc) return rc;
  HIPCHK(c, hipMemcpy(c->d_scratch, rows, bytes, hipMemcpyHostToDevice));
  // the trait contract (lib.rs:651-652): entries >= n_per_row are zero on entry; they are read as given here
  EncodeJob j;
  j.src = c->d_scratch; j.src_stride = c->n_cols; j.n_valid = c->n_cols; j.dst = c->d_scratch; j.n_rows = n_rows;
  if ((rc = encode_rows_device(c, &c->ws, j, nullptr, &c->err, nullptr))) return rc;
  HIPCHK(c, hipMemcpy(rows, c->d_scratch, bytes, hipMemcpyDeviceToHost));
  return 0;
  LCPC_CATCH(c)
}

}  // extern "C"

namespace lcpc {
// LcEncoding::encode for messages given WITHOUT their zero padding (msgs[i] = n_per_row elements): the padding is
// produced on the device (fused into the first NTT pass / the Brakedown input copy), only n_per_row elements per row
// cross PCIe.  out: n_rows x n_cols.  The verifier's 1 + n_degree_tests row encodes (lib.rs:886, 918).
int encode_msgs_host(lcpc_ctx* c, const uint64_t* const* msgs, uint64_t n_rows, uint64_t* out) {
  if (n_rows == 0) return 0;
  std::lock_guard<std::mutex> g(c->mu);
  HIPCHK(c, hipSetDevice(c->prm.device));
  const size_t eb = elem_bytes(c);
  const size_t in_b = ((size_t)n_rows * c->n_per_row * eb + 255) & ~(size_t)255, out_b = (size_t)n_rows * c->n_cols * eb;
  int rc = ensure_dev(&c->err, &c->d_scratch, &c->scratch_cap, in_b + out_b);
  if (rc) return rc;
  uint8_t* base = reinterpret_cast<uint8_t*>(c->d_scratch);
  for (uint64_t i = 0; i < n_rows; i++)
    HIPCHK(c, hipMemcpyAsync(base + i * c->n_per_row * eb, msgs[i], c->n_per_row * eb, hipMemcpyHostToDevice, nullptr));
  EncodeJob j;
  j.src = reinterpret_cast<uint32_t*>(base); j.src_stride = c->n_per_row; j.n_valid = c->n_per_row;
  j.dst = reinterpret_cast<uint32_t*>(base + in_b); j.n_rows = n_rows;
  if ((rc = encode_rows_device(c, &c->ws, j, nullptr, &c->err, nullptr))) return rc;
  HIPCHK(c, hipMemcpy(out, base + in_b, out_b, hipMemcpyDeviceToHost));
  return 0;
}
}  // namespace lcpc

extern "C" {

int lcpc_field_sum_device(lcpc_ctx* c, const uint64_t* parts, uint32_t n_parts, uint64_t n_elems, void* stream, uint64_t* out) {
  if (!c || !parts || !out || n_parts == 0) return LCPC_ERR_ARG;
  // stateless apart from the error text: callable from any thread while commitments of this encoder are busy, so a failure
  // must not write c->err without the lock
  hipError_t he = hipSetDevice(c->prm.device);
  if (he == hipSuccess)
    he = launch_field_sum(c->NL, reinterpret_cast<const uint32_t*>(parts), n_parts, n_elems, reinterpret_cast<uint32_t*>(out), (hipStream_t)stream);
  if (he != hipSuccess) {
    std::lock_guard<std::mutex> g(c->mu);
    return fail_hip(&c->err, he, "lcpc_field_sum_device");
  }
  return 0;
}

void lcpc_root_bincode(const uint8_t root[32], uint8_t out[40]) {
  const uint64_t l = 32;
  memcpy(out, &l, 8);
  memcpy(out + 8, root, 32);
}
void lcpc_free(void* p) { lcpc::proof_buf_free(p); }

}  // extern "C"
