//! Raw declarations of `include/lcpc_hip.h` (ABI version 5): the C ABI of the MI355X-native lcpc-2d commit / prove path.
//!
//! One item per item of the header, same names, same order of arguments; `tests/test_rust_bindings.py` of the repository
//! compares this file with the header (symbols, argument types, struct fields, constants) on every run of the CPU test suite,
//! because the image the library is built in has no Rust toolchain.
//!
//! Conventions of the ABI (header, lines 11-24): every function returns 0 or a negative `lcpc_status`; field elements cross as
//! `ff_derive` stores them (L little-endian `u64` limbs, Montgomery form), so `&[Ft255]` is passed as `*const u64` unchanged;
//! digests are 32 raw bytes; `*_device` entry points take HIP device pointers and a `hipStream_t` as `*mut c_void`.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const LCPC_ABI_VERSION: c_int = 5;

// fields of lcpc-test-fields/src/lib.rs:13-59
pub const LCPC_FT63: u32 = 0;
pub const LCPC_FT127: u32 = 1;
pub const LCPC_FT191: u32 = 2;
pub const LCPC_FT255: u32 = 3;
// encodings: lcpc-ligero-pc/src/lib.rs:31-37 (LigeroEncodingRho), lcpc-brakedown-pc/src/lib.rs:41-47 (SdigEncodingS)
pub const LCPC_ENC_LIGERO: u32 = 0;
pub const LCPC_ENC_SDIG: u32 = 1;
// D: Digest -- every reference test / bench uses blake3::Hasher
pub const LCPC_HASH_BLAKE3: u32 = 0;

// lcpc_status
pub const LCPC_OK: c_int = 0;
pub const LCPC_ERR_TOO_BIG: c_int = -1;
pub const LCPC_ERR_ENCODE: c_int = -2;
pub const LCPC_ERR_COMMIT: c_int = -3;
pub const LCPC_ERR_COLUMN_NUMBER: c_int = -4;
pub const LCPC_ERR_OUTER_TENSOR: c_int = -5;
pub const LCPC_ERR_DIMS: c_int = -6;
pub const LCPC_ERR_ARG: c_int = -7;
pub const LCPC_ERR_STATE: c_int = -8;
pub const LCPC_ERR_HIP: c_int = -16;
pub const LCPC_ERR_NOMEM: c_int = -17;
pub const LCPC_ERR_NO_DEVICE: c_int = -18;
pub const LCPC_ERR_XCHG: c_int = -19;
pub const LCPC_ERR_NO_RCCL: c_int = -20;
pub const LCPC_VERR_NUM_COL_OPENS: c_int = -32;
pub const LCPC_VERR_COLUMN_PATH: c_int = -33;
pub const LCPC_VERR_COLUMN_EVAL: c_int = -34;
pub const LCPC_VERR_COLUMN_DEGREE: c_int = -35;
pub const LCPC_VERR_OUTER_TENSOR: c_int = -36;
pub const LCPC_VERR_INNER_TENSOR: c_int = -37;
pub const LCPC_VERR_ENCODING_DIMS: c_int = -38;
pub const LCPC_VERR_ENCODE: c_int = -39;
pub const LCPC_VERR_MALFORMED: c_int = -40;

pub const LCPC_COMMIT_BORROW_COEFFS: u32 = 1;
pub const LCPC_COMMIT_ASYNC_TAIL: u32 = 2;

/// an LcEncoding implementor (`enc`): opaque
#[repr(C)]
pub struct lcpc_ctx {
    _private: [u8; 0],
}
/// an LcCommit<D, E>: opaque
#[repr(C)]
pub struct lcpc_commit_t {
    _private: [u8; 0],
}
/// a merlin::Transcript: opaque
#[repr(C)]
pub struct lcpc_transcript {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct lcpc_params {
    pub field: u32,
    pub encoding: u32,
    pub hash: u32,
    pub rho_num: u32,
    pub rho_den: u32,
    pub sdig_code: u32,
    pub seed: u64,
    pub n_coeffs: u64,
    pub n_per_row: u64,
    pub n_cols: u64,
    pub device: i32,
    pub shard_rank: u32,
    pub shard_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct lcpc_timings {
    pub encode_ms: f32,
    pub hash_ms: f32,
    pub merkle_ms: f32,
    pub total_ms: f32,
    pub encode_launches: u32,
    pub hash_launches: u32,
    pub merkle_launches: u32,
    pub exchange_exposed_ms: f32,
    pub staged_slices: u32,
    pub exchange_wire_ms: f32,
}

pub type lcpc_write_fn = Option<unsafe extern "C" fn(user: *mut c_void, data: *const u8, len: u64) -> c_int>;
pub type lcpc_read_fn = Option<unsafe extern "C" fn(user: *mut c_void, data: *mut u8, len: u64) -> c_int>;
pub type lcpc_allgather_fn = Option<unsafe extern "C" fn(user: *mut c_void, bytes: u64) -> c_int>;

extern "C" {
    // ---- construction: LigeroEncoding::new / new_from_dims (ligero lib.rs:121-148), SdigEncoding::new / new_from_dims
    //      (brakedown lib.rs:103-137) ----
    pub fn lcpc_ctx_create(params: *const lcpc_params, out: *mut *mut lcpc_ctx) -> c_int;
    pub fn lcpc_ctx_destroy(ctx: *mut lcpc_ctx);
    pub fn lcpc_strerror(status: c_int) -> *const c_char;
    pub fn lcpc_last_error(ctx: *const lcpc_ctx) -> *const c_char;
    pub fn lcpc_abi_version() -> c_int;

    // ---- LcEncoding trait (lcpc-2d/src/lib.rs:74-104) ----
    pub fn lcpc_get_dims(ctx: *const lcpc_ctx, len: u64, n_rows: *mut u64, n_per_row: *mut u64, n_cols: *mut u64) -> c_int;
    pub fn lcpc_dims_ok(ctx: *const lcpc_ctx, n_per_row: u64, n_cols: u64) -> c_int;
    pub fn lcpc_get_n_col_opens(ctx: *const lcpc_ctx) -> u64;
    pub fn lcpc_get_n_degree_tests(ctx: *const lcpc_ctx) -> u64;
    pub fn lcpc_field_limbs(ctx: *const lcpc_ctx) -> u32;
    pub fn lcpc_static_get_dims(params: *const lcpc_params, n_rows: *mut u64, n_per_row: *mut u64, n_cols: *mut u64) -> c_int;
    pub fn lcpc_static_get_dims_ml(
        params: *const lcpc_params,
        n_vars: u32,
        n_rows: *mut u64,
        n_per_row: *mut u64,
        n_cols: *mut u64,
    ) -> c_int;
    pub fn lcpc_random_coeffs_device(
        ctx: *mut lcpc_ctx,
        seed: *const u8,
        stream_id: u64,
        n: u64,
        out_dev: *mut u64,
        stream: *mut c_void,
    ) -> c_int;
    pub fn lcpc_encode_rows(ctx: *mut lcpc_ctx, rows_host: *mut u64, n_rows: u64) -> c_int;

    // ---- LcCommit (lcpc-2d/src/lib.rs:172-184, 270-312) ----
    pub fn lcpc_commit_create(enc: *mut lcpc_ctx, out: *mut *mut lcpc_commit_t) -> c_int;
    pub fn lcpc_commit_destroy(cm: *mut lcpc_commit_t);
    pub fn lcpc_commit_last_error(cm: *const lcpc_commit_t) -> *const c_char;
    pub fn lcpc_commit(cm: *mut lcpc_commit_t, coeffs_host: *const u64, n_coeffs: u64, root: *mut u8) -> c_int;
    pub fn lcpc_commit_device(
        cm: *mut lcpc_commit_t,
        coeffs_dev: *const u64,
        n_coeffs: u64,
        stream: *mut c_void,
        flags: u32,
        root: *mut u8,
    ) -> c_int;
    pub fn lcpc_commit_from_parts(
        cm: *mut lcpc_commit_t,
        comm_host: *const u64,
        coeffs_host: *const u64,
        n_rows: u64,
        root: *mut u8,
    ) -> c_int;
    // serde of LcCommit itself (lib.rs:186-268), bincode 1.3, streamed
    pub fn lcpc_commit_bincode_size(cm: *const lcpc_commit_t) -> u64;
    pub fn lcpc_commit_bincode_write(cm: *mut lcpc_commit_t, write: lcpc_write_fn, user: *mut c_void) -> c_int;
    pub fn lcpc_commit_from_bincode(cm: *mut lcpc_commit_t, read: lcpc_read_fn, user: *mut c_void, root: *mut u8) -> c_int;
    pub fn lcpc_get_root(cm: *mut lcpc_commit_t, root: *mut u8) -> c_int;
    pub fn lcpc_commit_dims(
        cm: *const lcpc_commit_t,
        n_rows: *mut u64,
        n_per_row: *mut u64,
        n_cols: *mut u64,
        n_hashes: *mut u64,
    ) -> c_int;
    pub fn lcpc_get_hashes(cm: *mut lcpc_commit_t, hashes: *mut u8) -> c_int;
    pub fn lcpc_get_comm(cm: *mut lcpc_commit_t, row0: u64, n_rows: u64, out: *mut u64) -> c_int;
    pub fn lcpc_get_coeffs(cm: *mut lcpc_commit_t, row0: u64, n_rows: u64, out: *mut u64) -> c_int;

    // collapse_columns (lib.rs:1095-1123), open_column (lib.rs:788-825)
    pub fn lcpc_collapse(cm: *mut lcpc_commit_t, tensors_host: *const u64, n_tensors: u32, polys_host: *mut u64) -> c_int;
    pub fn lcpc_open_columns(cm: *mut lcpc_commit_t, cols: *const u64, n: u32, col_vals: *mut u64, paths: *mut u8) -> c_int;

    // ---- merlin::Transcript ----
    pub fn lcpc_transcript_new(label: *const u8, len: usize) -> *mut lcpc_transcript;
    pub fn lcpc_transcript_clone(t: *const lcpc_transcript) -> *mut lcpc_transcript;
    pub fn lcpc_transcript_append_message(t: *mut lcpc_transcript, label: *const u8, llen: usize, msg: *const u8, mlen: usize);
    pub fn lcpc_transcript_append_messages(
        t: *mut lcpc_transcript,
        label: *const u8,
        llen: usize,
        msgs: *const u8,
        mlen: usize,
        n: usize,
    );
    pub fn lcpc_transcript_challenge_bytes(t: *mut lcpc_transcript, label: *const u8, llen: usize, out: *mut u8, n: usize);
    pub fn lcpc_transcript_free(t: *mut lcpc_transcript);

    // ---- prove / verify (lib.rs:304-311 -> 1004-1093; 518-527 -> 832-952) ----
    pub fn lcpc_prove(
        cm: *mut lcpc_commit_t,
        outer_tensor: *const u64,
        n_outer: u64,
        tr: *mut lcpc_transcript,
        proof: *mut *mut u8,
        proof_len: *mut u64,
        cols_opened: *mut u64,
    ) -> c_int;
    pub fn lcpc_verify(
        ctx: *mut lcpc_ctx,
        root: *const u8,
        outer_tensor: *const u64,
        n_outer: u64,
        inner_tensor: *const u64,
        n_inner: u64,
        proof: *const u8,
        proof_len: u64,
        tr: *mut lcpc_transcript,
        eval_out: *mut u64,
    ) -> c_int;
    pub fn lcpc_root_bincode(root: *const u8, out: *mut u8);
    pub fn lcpc_free(p: *mut c_void);

    // ---- row-sharded commit across GPUs ----
    pub fn lcpc_shard_layout(
        ctx: *const lcpc_ctx,
        n_rows_total: u64,
        row_begin: *mut u64,
        row_end: *mut u64,
        chunk_begin: *mut u64,
        chunk_end: *mut u64,
        n_chunks_total: *mut u64,
    ) -> c_int;
    pub fn lcpc_shard_nodes(
        n_chunks_total: u64,
        shard_count: u32,
        shard_rank: u32,
        n_nodes: *mut u32,
        first_chunk: *mut u64,
        log_size: *mut u32,
    ) -> c_int;
    pub fn lcpc_shard_nodes_field(
        field: u32,
        n_chunks_total: u64,
        shard_count: u32,
        shard_rank: u32,
        n_nodes: *mut u32,
        first_chunk: *mut u64,
        log_size: *mut u32,
    ) -> c_int;
    pub fn lcpc_comm_unique_id(id: *mut u8) -> c_int;
    pub fn lcpc_comm_init(ctx: *mut lcpc_ctx, id: *const u8, rank: u32, world: u32) -> c_int;
    pub fn lcpc_comm_destroy(ctx: *mut lcpc_ctx) -> c_int;
    pub fn lcpc_commit_sharded_device(
        cm: *mut lcpc_commit_t,
        coeffs_local_dev: *const u64,
        n_rows_total: u64,
        stream: *mut c_void,
        flags: u32,
        root: *mut u8,
    ) -> c_int;
    pub fn lcpc_prove_sharded_rccl(
        cm: *mut lcpc_commit_t,
        outer_tensor: *const u64,
        n_outer: u64,
        tr: *mut lcpc_transcript,
        proof: *mut *mut u8,
        proof_len: *mut u64,
        cols_opened: *mut u64,
    ) -> c_int;
    pub fn lcpc_commit_shard_device(
        cm: *mut lcpc_commit_t,
        coeffs_local_dev: *const u64,
        n_rows_total: u64,
        stream: *mut c_void,
        flags: u32,
        nodes_dev: *mut u8,
    ) -> c_int;
    pub fn lcpc_commit_finish_device(
        cm: *mut lcpc_commit_t,
        gathered_dev: *mut u8,
        n_rows_total: u64,
        slots_per_rank: u32,
        stream: *mut c_void,
        root: *mut u8,
    ) -> c_int;
    pub fn lcpc_collapse_device(
        cm: *mut lcpc_commit_t,
        tensors_dev: *const u64,
        n_tensors: u32,
        stream: *mut c_void,
        polys_dev: *mut u64,
    ) -> c_int;
    pub fn lcpc_field_sum_device(
        ctx: *mut lcpc_ctx,
        parts_dev: *const u64,
        n_parts: u32,
        n_elems: u64,
        stream: *mut c_void,
        out_dev: *mut u64,
    ) -> c_int;
    pub fn lcpc_prove_sharded_bytes(ctx: *const lcpc_ctx, n_rows_total: u64) -> u64;
    pub fn lcpc_prove_sharded(
        cm: *mut lcpc_commit_t,
        outer_tensor: *const u64,
        n_outer: u64,
        tr: *mut lcpc_transcript,
        send_dev: *mut u8,
        recv_dev: *mut u8,
        max_bytes: u64,
        allgather: lcpc_allgather_fn,
        user: *mut c_void,
        proof: *mut *mut u8,
        proof_len: *mut u64,
        cols_opened: *mut u64,
    ) -> c_int;

    // ---- measurement hooks ----
    pub fn lcpc_shard_exchange_probe(cm: *mut lcpc_commit_t, stream: *mut c_void, bytes_in: *mut u64) -> c_int;
    pub fn lcpc_comm_rccl_version(version: *mut c_int) -> c_int;
    pub fn lcpc_set_timing(cm: *mut lcpc_commit_t, enable: c_int) -> c_int;
    pub fn lcpc_get_timings(cm: *mut lcpc_commit_t, out: *mut lcpc_timings) -> c_int;
}

#[cfg(test)]
mod tests {
    use super::*;

    /// the library on the link line is the one this file describes
    #[test]
    fn abi_version_matches() {
        assert_eq!(unsafe { lcpc_abi_version() }, LCPC_ABI_VERSION);
    }

    #[test]
    fn struct_layouts() {
        assert_eq!(std::mem::size_of::<lcpc_params>(), 72);
        assert_eq!(std::mem::size_of::<lcpc_timings>(), 40);
    }
}
