// lcpc_amd/csrc/kernels.h -- host-callable launchers for the gfx950 kernels (kernels.hip).
// Element pointers are `const uint32_t*` views of the L x u64 Montgomery limbs (NL = 2L words).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcpc {

struct NttPassArgs {
  const uint32_t* src;     // row-major, src_stride elements per row
  uint32_t* dst;           // row-major, dst_stride elements per row (may alias src when strides match)
  const uint32_t* roots;   // roots[i] = w^i (Montgomery), i < n/2
  const uint32_t* roots29; // Ft255 only: w^i * 2^261 mod p as 9 x 29-bit limbs, 12-word stride (fe_mul_r29)
  const uint32_t* qp29;    // Ft255 only: q*p (q < 32) as 9 x 29-bit limbs, 12-word stride; non-null selects the lazy-limb kernel
  const uint32_t* roots29c;// lazy-limb kernel only, may be null: w^i * 2^5 mod p, same layout.  Non-null = "canonical output":
                           // src is in Montgomery form, dst of the final pass holds canonical values (see ntt_pass_l9_kernel)
  uint32_t mont_prefix;    // canonical output, final pass: elements [0, mont_prefix) of a row never met a multiplier and are
                           // converted at the store
  uint64_t src_stride, dst_stride;
  uint64_t n_valid;        // elements >= n_valid of every src row read as zero (fused zero padding)
  uint64_t n_src_total;    // flat src elements >= n_src_total read as zero (ragged last row)
  uint32_t* copy_dst;      // if non-null (first pass): padded copy of src, same strides (LcCommit.coeffs)
  uint64_t n_rows;
  uint32_t log_n, t0, s, log_tj;   // stages [t0, t0+s) on tiles of 2^s x 2^log_tj elements
  uint32_t* mid = nullptr; // shape-specialised two-pass kernel only, may be null: the 29-bit-limb intermediate between the passes
                           // (n_rows x n_cols x 36 bytes, layout in ntt_l9s.hip); the first pass writes it instead of dst, the
                           // last pass reads it instead of src
  uint32_t canon_row_mask = 0; // canonical output (roots29c != null) applies to the rows r with (r & canon_row_mask) == 0 only: the
                           // three-pass plan runs its last two passes on n_rows << s0 sub-rows of 2^20 elements, of which only every
                           // 2^s0-th still holds never-multiplied elements (ntt_l9s.hip)
  uint32_t blk0_gone = 0;  // shape-specialised kernel, canonical output: an earlier pass converted the last never-multiplied elements
                           // (the round before its uniform round): this pass sees canonical values only
  const uint32_t* wq_w = nullptr;  // shape-specialised last pass (ntt_l9s.hip): the shifted multiples of w^(n/4) (81 words, ctx.cpp wmul_table), the
                           // one wave-uniform twiddle of the transform (stages k-2, k-1); null: the Montgomery-form table entry
  uint32_t tile_group = 0; // shape-specialised first pass (ntt_l9s.hip, ntt_lns.hip): 2^tile_group neighbouring tiles of a row are
                           // consecutive workgroups of one XCD (short strided runs then meet in that L2); needs
                           // tiles_per_row >= 8 << tile_group
};
hipError_t launch_ntt_pass(int nl, int log_tile, const NttPassArgs& a, hipStream_t st);

// ---- shape-specialised Ft255 NTT for two-pass plans on 1024-element tiles (ntt_l9s.hip) ----
// twiddle pack of one pass: per tile class, per round, the table entries in lane order (layout: ntt_l9s.hip)
struct NttPackInfo {
  uint32_t round_off[8];    // word offset of round slot r inside a class block
  uint32_t class_words;     // words per tile class
  uint32_t u_off;           // passes with a uniform round (ntt_l9s.hip Shape::RU): word offset of its 4 x 3 shifted-multiples tables
};
bool ntt_l9s_supported(uint32_t log_n, uint32_t n_passes, int log_tile);
// three passes for 2^21 .. 2^26 columns: s0 = log_n - 20 stages with the first-pass kernel on the whole rows (element stride 2^20),
// then the two-pass plan of a 2^20-point transform on each of the 2^s0 contiguous blocks of every row (DIF: after s0 stages the
// blocks are independent transforms with the root w^(2^s0)), from a sub-sampled twiddle table
bool ntt_l9s3_supported(uint32_t log_n);
// sub[i] = tab[i << shift], 12-word entries, i < n
hipError_t launch_ntt_l9s_subtable(const uint32_t* tab, uint32_t shift, uint64_t n, uint32_t* sub, hipStream_t st);
NttPackInfo ntt_l9s_pack_info(uint32_t s, bool first);
// a: the pass (log_n, t0, s, log_tj, roots29, roots29c); first pass: n_classes = tiles per row, last pass: 1
hipError_t launch_ntt_l9s_pack(const NttPassArgs& a, bool first, const NttPackInfo& pi, uint32_t n_classes, uint32_t* pack, hipStream_t st);
hipError_t launch_ntt_pass_l9s(const NttPassArgs& a, bool first, const uint32_t* pack, const NttPackInfo& pi, hipStream_t st);

// ---- shape-specialised lazy-limb NTT for Ft63 / Ft127 / Ft191, two-pass plans on 1024-element tiles (ntt_lns.hip) ----
// nl = 2 / 4 / 6.  a.roots29 = the limb-form twiddle table (w^i R' mod p, ntt_lns_stride words per entry), a.qp29 = the
// (i - 24) p table (64 rows, same stride)
bool ntt_lns_supported(int nl, uint32_t log_n);
// three passes for 2^21 .. 2^26 columns, built like ntt_l9s3_supported's plan (first-pass kernel over the whole rows, then the
// 2^20-point two-pass plan per block from sub-sampled tables)
bool ntt_lns3_supported(int nl, uint32_t log_n);
hipError_t launch_ntt_lns_subtable(int nl, const uint32_t* tab, uint32_t shift, uint64_t n, uint32_t* sub, hipStream_t st);
int ntt_lns_limbs(int nl);
int ntt_lns_limb_bits(int nl);
int ntt_lns_stride(int nl);
NttPackInfo ntt_lns_pack_info(int nl, uint32_t s, bool first);
hipError_t launch_ntt_lns_roots(int nl, const uint32_t* roots, uint64_t n, const uint32_t* rprime, uint32_t* out, hipStream_t st);
hipError_t launch_ntt_lns_pack(int nl, const NttPassArgs& a, bool first, const NttPackInfo& pi, uint32_t n_classes, uint32_t* pack, hipStream_t st);
hipError_t launch_ntt_pass_lns(int nl, const NttPassArgs& a, bool first, const uint32_t* pack, const NttPackInfo& pi, hipStream_t st);

// device-side precomp_fft: roots[i] = w^i (i < 2^log_half) from pw[j] = w^(2^j); roots29 (Ft255) may be null
hipError_t launch_roots(int nl, const uint32_t* pw, uint32_t log_half, const uint32_t* one, uint32_t* roots, uint32_t* roots29,
                        uint32_t* roots29c, hipStream_t st);

struct LeafArgs {
  const uint32_t* comm;        // local rows, row-major
  uint64_t row_stride;         // elements between consecutive rows of a column (n_cols for the row-major comm)
  uint64_t col_stride;         // elements between consecutive columns of a row (1 for row-major; n_rows for a position-major copy)
  uint64_t n_cols;
  int64_t  row_base;           // global index of local row 0
  uint64_t n_rows_total;
  uint32_t chunk_begin, n_chunks_local, n_chunks_total;
  uint32_t* out;               // n_chunks_total == 1: digests [n_cols][8]; else CVs [n_chunks_local][n_cols][8]
  uint32_t canon_in;           // comm already holds canonical values (Ft255 Ligero commit): no Montgomery reduction here
};
hipError_t launch_leaf_chunks(int nl, const LeafArgs& a, hipStream_t st);
// cvs [n_chunks][n_cols][8] (clobbered: used as the BLAKE3 CV stack) -> digests [n_cols][8]
hipError_t launch_leaf_finish(uint32_t* cvs, uint32_t n_chunks, uint64_t n_cols, uint32_t* digests, hipStream_t st);
// general form: node j = aligned subtree of 2^node_log[j] chunks, CV at cvs[node_slot[j]][col] (tables on the device,
// nullptr = identity / all zero); root = false only pre-merges (no ROOT flag) into one subtree CV per column
hipError_t launch_leaf_finish_nodes(uint32_t* cvs, const uint32_t* node_slot, const uint32_t* node_log, uint32_t n_nodes,
                                    uint64_t n_cols, uint32_t* out, bool root, hipStream_t st);
// whole tree above the leaf layer in as few launches as possible (hashes = LcCommit.hashes, np2 leaves)
// root_out (may be null): 8 more words the root is written to by the launch that produces it (host-mapped memory)
hipError_t launch_merkle_tree(uint32_t* hashes, uint64_t np2, hipStream_t st, uint32_t* root_out);
hipError_t launch_merkle_tree_from(uint32_t* hashes, uint64_t np2, uint32_t levels_done, hipStream_t st, uint32_t* root_out);
// small commitments: leaf digests (<= 2 chunks per leaf message) AND the first six tree levels in one launch (np2 == n_cols,
// n_cols % 64 == 0, n_cols >= 128, n_cols * n_chunks <= 65536); follow with launch_merkle_tree_from(.., 6, ..)
bool leaf_tree_supported(const LeafArgs& a, uint64_t np2);
hipError_t launch_leaf_tree(int nl, const LeafArgs& a, uint32_t* hashes, uint64_t np2, hipStream_t st);

struct CollapseArgs {
  const uint32_t* coeffs;      // local rows x n_per_row
  const uint32_t* tensors;     // [n_tensors][n_rows_local]
  const uint32_t* tensors29;   // Ft255: the same tensors as 9 x 29-bit limbs of t * 2^261 (12-word stride), or null
  uint32_t* out;               // n_splits == 1: polys [n_tensors][n_per_row]; else partial [n_splits][n_tensors][n_per_row]
  uint64_t n_rows, n_per_row;
  uint32_t n_tensors, n_splits;
  // column range [j0, j1) of this launch (j1 == 0: all of them) and the element stride between the output rows: a range's outputs sit
  // at out + ((split * n_tensors + tensor) * out_stride + (j - j0)) elements (prove collapses p_random in two ranges: commit.cpp)
  uint64_t j0 = 0, j1 = 0, out_stride = 0;
};
hipError_t launch_collapse(int nl, const CollapseArgs& a, hipStream_t st);
// Ft255 tensors (Montgomery) -> the 29-bit-limb / 2^261 form used by the lazy dot-product kernels
hipError_t launch_to_r29(const uint32_t* in, uint64_t n, uint32_t* out, hipStream_t st);
// out[e] = sum_p parts[p][e] mod p
hipError_t launch_field_sum(int nl, const uint32_t* parts, uint32_t n_parts, uint64_t n_elems, uint32_t* out, hipStream_t st);

// open_column: vals[k][r] = comm[r][cols[k]]; paths[k][lvl] = sibling digest.  r2 non-null: comm holds canonical
// values, multiply by R^2 on the way out so that vals are Montgomery-form elements like everything else at the ABI
// element (r, c) at comm + (r * row_stride + c * col_stride) elements (row-major comm: n_cols, 1; position-major: 1, n_rows)
hipError_t launch_gather_columns(int nl, const uint32_t* comm, uint64_t n_rows, uint64_t row_stride, uint64_t col_stride,
                                 const uint64_t* cols, uint32_t n, uint32_t* vals, const uint32_t* r2, hipStream_t st);
// elementwise representation change of n elements (in may equal out): to_mont = x * R (r2 = R^2 mod p, Montgomery
// multiply), to_canon = x * R^-1 (Montgomery reduction = PrimeField::to_repr without the byte dump)
hipError_t launch_to_mont(int nl, const uint32_t* in, uint64_t n, const uint32_t* r2, uint32_t* out, hipStream_t st);
hipError_t launch_to_canon(int nl, const uint32_t* in, uint64_t n, uint32_t* out, hipStream_t st);
// sharded open_column: [rank][k][rows of rank] (blocks of block_words) -> [k][all rows]; rb: G + 1 row boundaries on the device
hipError_t launch_assemble_columns(int nl, const uint32_t* recv, uint64_t block_words, const uint64_t* rb, uint32_t G, uint32_t n,
                                   uint64_t n_rows, uint32_t* out, hipStream_t st);
hipError_t launch_gather_paths(const uint32_t* hashes, uint64_t np2, uint32_t path_len, const uint64_t* cols, uint32_t n,
                               uint32_t* paths, hipStream_t st);

// Brakedown: y[row][out_off + o] = sum_k vals[k] * x[row][in_off + colidx[k]], k in [rowptr[o], rowptr[o+1])
struct SpmvArgs {
  uint32_t* mat;                // comm (row-major, stride elements per row); in and out segments are disjoint
  uint32_t* out_alt;            // if non-null: write to out_alt[row][o] (stride out_alt_stride) instead of mat
  uint64_t stride, out_alt_stride;
  uint64_t in_off, out_off;
  const uint32_t* rowptr;       // [m+1]
  const uint32_t* colidx;       // [nnz]
  const uint32_t* vals;         // [nnz][NL]
  const uint32_t* vals29;       // Ft255: [nnz][12], the 29-bit-limb / 2^261 form (may be null)
  uint64_t m, n_rows;
};
hipError_t launch_spmv(int nl, const SpmvArgs& a, hipStream_t st);
// ---- position-major ("transposed") Brakedown path: T[pos][row], row fastest -----------------------
// rows x n_valid block of a row-major matrix -> T (leading dimension n_rows); and back
hipError_t launch_transpose_to_t(int nl, const uint32_t* src, uint64_t src_stride, uint64_t n_valid, uint64_t n_rows,
                                 uint32_t* t, hipStream_t st, uint64_t n_src_total = ~(uint64_t)0, uint32_t* copy_dst = nullptr,
                                 bool canon = false);   // canon: T receives canonical values (x R^-1); copy_dst the values as read
hipError_t launch_transpose_from_t(int nl, const uint32_t* t, uint64_t n_pos, uint64_t n_rows, uint32_t* dst,
                                   uint64_t dst_stride, hipStream_t st);
struct SpmmTArgs {
  uint32_t* t;                  // T[pos][row]
  uint32_t* out_alt;            // if non-null: outputs go to out_alt[o][row] instead of t[out_off + o][row]
  uint64_t n_rows;
  uint64_t in_off, out_off;
  const uint32_t* rowptr;       // [m+1]
  const uint32_t* colidx;       // [nnz]
  const uint32_t* vals;         // [nnz][NL]  Montgomery (R = 2^(32 NL))
  const uint32_t* vals29;       // Ft255: [nnz][12]  value * 2^261 mod p as 9 x 29-bit limbs
  uint64_t m;
};
hipError_t launch_spmm_t(int nl, const SpmmTArgs& a, hipStream_t st);
hipError_t launch_sdig_rs_t(int nl, const uint32_t* in_t, uint32_t n_in, uint32_t* t, uint64_t out_off, uint32_t n_out,
                            uint64_t n_rows, const uint32_t* r2, hipStream_t st);

// Reed-Solomon base case (encode.rs:97-110): out[row][out_off + k] = sum_j in[row][j] (k+1)^j
hipError_t launch_sdig_rs(int nl, const uint32_t* in, uint64_t in_stride, uint32_t n_in, uint32_t* mat, uint64_t stride,
                          uint64_t out_off, uint32_t n_out, uint64_t n_rows, const uint32_t* r2, hipStream_t st);
// copy rows with zero padding: dst[row][0..n_valid) = src[row][..], rest zero (Brakedown row setup)
hipError_t launch_pad_rows(int nl, const uint32_t* src, uint64_t src_stride, uint32_t* dst, uint64_t dst_stride,
                           uint64_t n_valid, uint64_t n_rows, hipStream_t st);

}  // namespace lcpc
