//! `LcEncoding` implementors and a device-resident `LcCommit` on top of `liblcpc_hip.so`.
//!
//! What the reference offers for this path and what stands in for it here:
//!
//! | reference (conroi/lcpc)                                              | this crate                                      |
//! |----------------------------------------------------------------------|-------------------------------------------------|
//! | `LigeroEncodingRho<Ft, Rn, Rd>` (lcpc-ligero-pc/src/lib.rs:31-186)   | [`HipLigeroEncodingRho<Ft, Rn, Rd>`]            |
//! | `LigeroEncoding<Ft>` (lib.rs:189)                                    | [`HipLigeroEncoding<Ft>`]                       |
//! | `SdigEncodingS<Ft, S>` (lcpc-brakedown-pc/src/lib.rs:41-176)         | [`HipSdigEncodingS<Ft, S>`]                     |
//! | `SdigEncoding<Ft>` (lib.rs:179)                                      | [`HipSdigEncoding<Ft>`]                         |
//! | `LcCommit::<D, E>::commit / prove / get_root` (lcpc-2d lib.rs:270-312) | [`HipCommit::commit / prove / get_root`]      |
//! | `LcEvalProof::verify` (lcpc-2d lib.rs:518-527)                       | the reference's own, on the proof `prove` returns; or [`verify_on_device`] |
//! | `merlin::Transcript`                                                 | [`HipTranscript`] (byte-exact STROBE-128 restatement inside the library) |
//!
//! The encoders implement `lcpc_2d::LcEncoding` with the reference's labels, so the reference's generic code -- its
//! `verify`, its tests -- runs on them unchanged; `encode` is the batched single-row entry point (`lcpc_encode_rows`), which is
//! what the verifier calls (lcpc-2d lib.rs:886, 918).  The reference's `LcCommit::commit` would also work on them, one `encode`
//! call per row from Rayon, but that is the slow seam: [`HipCommit`] replaces `commit` and `prove` one level up, with the
//! commitment resident in HBM.  Proofs come back as the reference's own `LcEvalProof<Blake3, E>` (through its bincode form, the
//! only door: the fields are private).
//!
//! Not compiled in the image this repository is built in (no Rust there); `tests/test_rust_bindings.py` checks the FFI layer
//! against the header, and the calls made here are the ones `lcpc_amd/__init__.py` (ctypes) makes in every GPU test.
#![deny(missing_docs)]

pub use lcpc_hip_sys as sys;

use blake3::Hasher as Blake3;
use digest::Output;
use ff::PrimeField;
use lcpc_2d::{def_labels, FieldHash, LcEncoding, LcEvalProof, LcRoot, ProverError, SizedField, VerifierError};
use lcpc_brakedown_pc::codespec::{SdigCode1, SdigCode2, SdigCode3, SdigCode4, SdigCode5, SdigCode6, SdigSpecification};
use serde::{Deserialize, Serialize};
use std::ffi::CStr;
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};
use std::sync::Arc;
use typenum::{Unsigned, U1, U2};

// ---------------------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------------------

/// A negative `lcpc_status` that has no counterpart in the reference's error enums (device runtime, exchange, arguments),
/// with the library's detail string.  `LcEncoding::Err` of the encoders below.
#[derive(Debug, Clone)]
pub struct HipError {
    /// the `lcpc_status` value (include/lcpc_hip.h:46-74)
    pub status: c_int,
    /// `lcpc_last_error` / `lcpc_commit_last_error` at the time of the failure (may be empty)
    pub detail: String,
}

impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        let what = unsafe { CStr::from_ptr(sys::lcpc_strerror(self.status)) }.to_string_lossy();
        if self.detail.is_empty() {
            write!(f, "lcpc_hip: {} ({})", what, self.status)
        } else {
            write!(f, "lcpc_hip: {} ({}): {}", what, self.status, self.detail)
        }
    }
}

impl std::error::Error for HipError {}

fn cstr(p: *const std::os::raw::c_char) -> String {
    if p.is_null() {
        String::new()
    } else {
        unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
    }
}

/// status of a prover entry point -> the reference's `ProverError` (lcpc-2d lib.rs:111-131)
fn prover_error(status: c_int, detail: String) -> ProverError<HipError> {
    match status {
        sys::LCPC_ERR_TOO_BIG => ProverError::TooBig,
        sys::LCPC_ERR_COMMIT => ProverError::Commit,
        sys::LCPC_ERR_COLUMN_NUMBER => ProverError::ColumnNumber,
        sys::LCPC_ERR_OUTER_TENSOR => ProverError::OuterTensor,
        _ => ProverError::Encode(HipError { status, detail }),
    }
}

/// status of `lcpc_verify` -> the reference's `VerifierError` (lcpc-2d lib.rs:137-166)
fn verifier_error(status: c_int, detail: String) -> VerifierError<HipError> {
    match status {
        sys::LCPC_VERR_NUM_COL_OPENS => VerifierError::NumColOpens,
        sys::LCPC_VERR_COLUMN_PATH => VerifierError::ColumnPath,
        sys::LCPC_VERR_COLUMN_EVAL => VerifierError::ColumnEval,
        sys::LCPC_VERR_COLUMN_DEGREE => VerifierError::ColumnDegree,
        sys::LCPC_VERR_OUTER_TENSOR => VerifierError::OuterTensor,
        sys::LCPC_VERR_INNER_TENSOR => VerifierError::InnerTensor,
        sys::LCPC_VERR_ENCODING_DIMS => VerifierError::EncodingDims,
        _ => VerifierError::Encode(HipError { status, detail }),
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fields and code specifications the library implements
// ---------------------------------------------------------------------------------------------------------------------

/// A prime field the device library implements.
///
/// # Safety
/// `Self` must be exactly `LIMBS` little-endian `u64` limbs holding the Montgomery representation (R = 2^(64 LIMBS)) of the
/// element modulo the prime the library associates with `FIELD` -- what `#[derive(PrimeField)]` of `ff_derive` produces for the
/// four fields of `lcpc-test-fields` (lcpc-test-fields/src/lib.rs:13-59).  Slices of `Self` are handed to the library as
/// `*const u64` without conversion.
pub unsafe trait HipField: PrimeField + FieldHash + SizedField + Serialize + for<'de> Deserialize<'de> {
    /// `LCPC_FT*`
    const FIELD: u32;
    /// limbs per element
    const LIMBS: usize;
}

unsafe impl HipField for lcpc_test_fields::ft63::Ft63 {
    const FIELD: u32 = sys::LCPC_FT63;
    const LIMBS: usize = 1;
}
unsafe impl HipField for lcpc_test_fields::ft127::Ft127 {
    const FIELD: u32 = sys::LCPC_FT127;
    const LIMBS: usize = 2;
}
unsafe impl HipField for lcpc_test_fields::ft191::Ft191 {
    const FIELD: u32 = sys::LCPC_FT191;
    const LIMBS: usize = 3;
}
unsafe impl HipField for lcpc_test_fields::ft255::Ft255 {
    const FIELD: u32 = sys::LCPC_FT255;
    const LIMBS: usize = 4;
}

fn limbs_of<Ft: HipField>(s: &[Ft]) -> *const u64 {
    debug_assert_eq!(std::mem::size_of::<Ft>(), 8 * Ft::LIMBS);
    s.as_ptr() as *const u64
}
fn limbs_of_mut<Ft: HipField>(s: &mut [Ft]) -> *mut u64 {
    debug_assert_eq!(std::mem::size_of::<Ft>(), 8 * Ft::LIMBS);
    s.as_mut_ptr() as *mut u64
}

/// The six code specifications of lcpc-brakedown-pc/src/codespec.rs:169-232 by the number the library knows them by.
pub trait HipSdigCode: SdigSpecification {
    /// 1..=6 (`lcpc_params.sdig_code`)
    const CODE: u32;
}
impl HipSdigCode for SdigCode1 {
    const CODE: u32 = 1;
}
impl HipSdigCode for SdigCode2 {
    const CODE: u32 = 2;
}
impl HipSdigCode for SdigCode3 {
    const CODE: u32 = 3;
}
impl HipSdigCode for SdigCode4 {
    const CODE: u32 = 4;
}
impl HipSdigCode for SdigCode5 {
    const CODE: u32 = 5;
}
impl HipSdigCode for SdigCode6 {
    const CODE: u32 = 6;
}

// ---------------------------------------------------------------------------------------------------------------------
// the encoder context
// ---------------------------------------------------------------------------------------------------------------------

/// Which GPU, and which row shard of how many (one encoder / process per GPU; `Shard::NONE` = unsharded).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct Placement {
    /// HIP device ordinal
    pub device: i32,
    /// this encoder holds shard `shard_rank` ...
    pub shard_rank: u32,
    /// ... of `shard_count` (0 or 1: unsharded)
    pub shard_count: u32,
}

impl Default for Placement {
    fn default() -> Self {
        Placement { device: 0, shard_rank: 0, shard_count: 1 }
    }
}

/// owner of an `lcpc_ctx*`: immutable after creation and usable from several host threads at once (include/lcpc_hip.h:16-21),
/// like the `&E` the reference shares across Rayon workers (lcpc-2d lib.rs:74-104)
struct Ctx(*mut sys::lcpc_ctx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { sys::lcpc_ctx_destroy(self.0) }
    }
}
impl std::fmt::Debug for Ctx {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "lcpc_ctx@{:p}", self.0)
    }
}

fn ctx_create(p: &sys::lcpc_params) -> Result<(Arc<Ctx>, usize, usize), HipError> {
    assert_eq!(unsafe { sys::lcpc_abi_version() }, sys::LCPC_ABI_VERSION, "liblcpc_hip.so: ABI version mismatch");
    let mut raw = std::ptr::null_mut();
    let rc = unsafe { sys::lcpc_ctx_create(p, &mut raw) };
    if rc != 0 {
        return Err(HipError { status: rc, detail: String::new() });
    }
    let ctx = Arc::new(Ctx(raw));
    let (mut nr, mut np, mut nc) = (0u64, 0u64, 0u64);
    let rc = unsafe { sys::lcpc_get_dims(raw, 1, &mut nr, &mut np, &mut nc) };
    if rc != 0 {
        return Err(HipError { status: rc, detail: cstr(unsafe { sys::lcpc_last_error(raw) }) });
    }
    Ok((ctx, np as usize, nc as usize))
}

/// What [`HipCommit`] needs from an encoder: its device context.  Implemented by the two encoders below.
pub trait HipEncoding: LcEncoding<Err = HipError>
where
    Self::F: HipField,
{
    /// the `lcpc_ctx*` behind this encoder
    fn raw_ctx(&self) -> *mut sys::lcpc_ctx;

    /// `lcpc_comm_init`: give a sharded encoder its RCCL communicator.  `id`: the 128 bytes rank 0 drew with
    /// [`comm_unique_id`] and distributed by whatever the host uses.  Collective: every rank calls it.
    fn comm_init(&self, id: &[u8; 128], rank: u32, world: u32) -> Result<(), HipError> {
        match unsafe { sys::lcpc_comm_init(self.raw_ctx(), id.as_ptr(), rank, world) } {
            0 => Ok(()),
            rc => Err(HipError { status: rc, detail: cstr(unsafe { sys::lcpc_last_error(self.raw_ctx()) }) }),
        }
    }

    /// `lcpc_shard_layout`: (row_begin, row_end, chunk_begin, chunk_end, n_chunks_total) of this encoder's shard
    fn shard_layout(&self, n_rows_total: usize) -> Result<(usize, usize, usize, usize, usize), HipError> {
        let mut v = [0u64; 5];
        let p = v.as_mut_ptr();
        let rc = unsafe { sys::lcpc_shard_layout(self.raw_ctx(), n_rows_total as u64, p, p.add(1), p.add(2), p.add(3), p.add(4)) };
        if rc != 0 {
            return Err(HipError { status: rc, detail: String::new() });
        }
        Ok((v[0] as usize, v[1] as usize, v[2] as usize, v[3] as usize, v[4] as usize))
    }
}

/// `ncclGetUniqueId` through the library (`lcpc_comm_unique_id`); `LCPC_ERR_NO_RCCL` when librccl cannot be loaded.
pub fn comm_unique_id() -> Result<[u8; 128], HipError> {
    let mut id = [0u8; 128];
    match unsafe { sys::lcpc_comm_unique_id(id.as_mut_ptr()) } {
        0 => Ok(id),
        rc => Err(HipError { status: rc, detail: String::new() }),
    }
}

// ---- Ligero ------------------------------------------------------------------------------------------------------------------

/// Drop-in for `LigeroEncodingRho<Ft, Rn, Rd>` (lcpc-ligero-pc/src/lib.rs:31-186): the twiddle tables live on the GPU.
#[derive(Clone, Debug)]
pub struct HipLigeroEncodingRho<Ft, Rn, Rd> {
    ctx: Arc<Ctx>,
    n_per_row: usize,
    n_cols: usize,
    _p: PhantomData<(Ft, Rn, Rd)>,
}

/// Drop-in for `LigeroEncoding<Ft>` (rate 1/2, lcpc-ligero-pc/src/lib.rs:189)
pub type HipLigeroEncoding<Ft> = HipLigeroEncodingRho<Ft, U1, U2>;

impl<Ft, Rn, Rd> HipLigeroEncodingRho<Ft, Rn, Rd>
where
    Ft: HipField,
    Rn: Unsigned + std::fmt::Debug + Sync,
    Rd: Unsigned + std::fmt::Debug + Sync,
{
    fn params(place: Placement) -> sys::lcpc_params {
        sys::lcpc_params {
            field: Ft::FIELD,
            encoding: sys::LCPC_ENC_LIGERO,
            hash: sys::LCPC_HASH_BLAKE3,
            rho_num: Rn::to_u32(),
            rho_den: Rd::to_u32(),
            device: place.device,
            shard_rank: place.shard_rank,
            shard_count: place.shard_count,
            ..Default::default()
        }
    }

    /// `LigeroEncodingRho::new(len)` (ligero lib.rs:121-124) on device 0
    pub fn new(len: usize) -> Self {
        Self::new_on(len, Placement::default()).expect("lcpc_ctx_create")
    }

    /// `new(len)` with an explicit device / shard; `Err` where the reference's `unwrap` / `assert!`s would fire, or when the
    /// device cannot be used
    pub fn new_on(len: usize, place: Placement) -> Result<Self, HipError> {
        let mut p = Self::params(place);
        p.n_coeffs = len as u64;
        let (ctx, n_per_row, n_cols) = ctx_create(&p)?;
        Ok(Self { ctx, n_per_row, n_cols, _p: PhantomData })
    }

    /// `LigeroEncodingRho::new_ml(n_vars)` (ligero lib.rs:128-135)
    pub fn new_ml(n_vars: usize) -> Self {
        let p = Self::params(Placement::default());
        let (mut nr, mut np, mut nc) = (0u64, 0u64, 0u64);
        let rc = unsafe { sys::lcpc_static_get_dims_ml(&p, n_vars as u32, &mut nr, &mut np, &mut nc) };
        assert_eq!(rc, 0, "new_ml: the reference's assert!s on the split fire here (LCPC_ERR_DIMS)");
        Self::new_from_dims(np as usize, nc as usize)
    }

    /// `LigeroEncodingRho::new_from_dims(n_per_row, n_cols)` (ligero lib.rs:138-148)
    pub fn new_from_dims(n_per_row: usize, n_cols: usize) -> Self {
        Self::new_from_dims_on(n_per_row, n_cols, Placement::default()).expect("lcpc_ctx_create")
    }

    /// `new_from_dims` with an explicit device / shard
    pub fn new_from_dims_on(n_per_row: usize, n_cols: usize, place: Placement) -> Result<Self, HipError> {
        let mut p = Self::params(place);
        p.n_per_row = n_per_row as u64;
        p.n_cols = n_cols as u64;
        let (ctx, n_per_row, n_cols) = ctx_create(&p)?;
        Ok(Self { ctx, n_per_row, n_cols, _p: PhantomData })
    }
}

// ---- Brakedown ---------------------------------------------------------------------------------------------------------------

/// Drop-in for `SdigEncodingS<Ft, S>` (lcpc-brakedown-pc/src/lib.rs:41-176): the expander matrices are generated on the host
/// exactly as `matgen::generate` does (same ChaCha20 streams, same draw order) and live on the GPU.
#[derive(Clone, Debug)]
pub struct HipSdigEncodingS<Ft, S> {
    ctx: Arc<Ctx>,
    n_per_row: usize,
    n_cols: usize,
    _p: PhantomData<(Ft, S)>,
}

/// Drop-in for `SdigEncoding<Ft>` (`SdigCodeDflt` = `SdigCode3`, lcpc-brakedown-pc/src/lib.rs:19, 179)
pub type HipSdigEncoding<Ft> = HipSdigEncodingS<Ft, SdigCode3>;

impl<Ft, S> HipSdigEncodingS<Ft, S>
where
    Ft: HipField,
    S: HipSdigCode + std::fmt::Debug + Clone + Sync,
{
    fn params(seed: u64, place: Placement) -> sys::lcpc_params {
        sys::lcpc_params {
            field: Ft::FIELD,
            encoding: sys::LCPC_ENC_SDIG,
            hash: sys::LCPC_HASH_BLAKE3,
            rho_num: 1,
            rho_den: 2,
            sdig_code: S::CODE,
            seed,
            device: place.device,
            shard_rank: place.shard_rank,
            shard_count: place.shard_count,
            ..Default::default()
        }
    }

    /// `SdigEncodingS::new(len, seed)` (brakedown lib.rs:103-110) on device 0
    pub fn new(len: usize, seed: u64) -> Self {
        Self::new_on(len, seed, Placement::default()).expect("lcpc_ctx_create")
    }

    /// `new(len, seed)` with an explicit device / shard
    pub fn new_on(len: usize, seed: u64, place: Placement) -> Result<Self, HipError> {
        let mut p = Self::params(seed, place);
        p.n_coeffs = len as u64;
        let (ctx, n_per_row, n_cols) = ctx_create(&p)?;
        Ok(Self { ctx, n_per_row, n_cols, _p: PhantomData })
    }

    /// `SdigEncodingS::new_ml(n_vars, seed)` (brakedown lib.rs:114-123)
    pub fn new_ml(n_vars: usize, seed: u64) -> Self {
        let p = Self::params(seed, Placement::default());
        let (mut nr, mut np, mut nc) = (0u64, 0u64, 0u64);
        let rc = unsafe { sys::lcpc_static_get_dims_ml(&p, n_vars as u32, &mut nr, &mut np, &mut nc) };
        assert_eq!(rc, 0, "new_ml");
        Self::new_from_dims(np as usize, nc as usize, seed)
    }

    /// `SdigEncodingS::new_from_dims(n_per_row, n_cols, seed)` (brakedown lib.rs:126-137)
    pub fn new_from_dims(n_per_row: usize, n_cols: usize, seed: u64) -> Self {
        Self::new_from_dims_on(n_per_row, n_cols, seed, Placement::default()).expect("lcpc_ctx_create")
    }

    /// `new_from_dims` with an explicit device / shard
    pub fn new_from_dims_on(n_per_row: usize, n_cols: usize, seed: u64, place: Placement) -> Result<Self, HipError> {
        let mut p = Self::params(seed, place);
        p.n_per_row = n_per_row as u64;
        p.n_cols = n_cols as u64;
        let (ctx, n_per_row, n_cols) = ctx_create(&p)?;
        Ok(Self { ctx, n_per_row, n_cols, _p: PhantomData })
    }
}

// ---- LcEncoding for both (lcpc-2d/src/lib.rs:74-104) -------------------------------------------------------------------------

/// `encode` of both implementors: one row of `n_cols` elements, the first `n_per_row` the message and the rest zero on entry
/// (the contract of lcpc-2d lib.rs:651-652), encoded in place by `lcpc_encode_rows`
fn encode_one<Ft: HipField>(ctx: &Ctx, n_cols: usize, row: &mut [Ft]) -> Result<(), HipError> {
    assert_eq!(row.len(), n_cols);
    match unsafe { sys::lcpc_encode_rows(ctx.0, limbs_of_mut(row), 1) } {
        0 => Ok(()),
        rc => Err(HipError { status: rc, detail: cstr(unsafe { sys::lcpc_last_error(ctx.0) }) }),
    }
}

impl<Ft, Rn, Rd> LcEncoding for HipLigeroEncodingRho<Ft, Rn, Rd>
where
    Ft: HipField,
    Rn: Unsigned + std::fmt::Debug + Sync + Clone,
    Rd: Unsigned + std::fmt::Debug + Sync + Clone,
{
    type F = Ft;
    type Err = HipError;

    def_labels!(ligero_pc); // the literal labels of the reference's implementor (macros.rs:31-34)

    fn encode<T: AsMut<[Ft]>>(&self, mut inp: T) -> Result<(), HipError> {
        encode_one(&self.ctx, self.n_cols, inp.as_mut())
    }

    fn get_dims(&self, len: usize) -> (usize, usize, usize) {
        ((len + self.n_per_row - 1) / self.n_per_row, self.n_per_row, self.n_cols)
    }

    fn dims_ok(&self, n_per_row: usize, n_cols: usize) -> bool {
        unsafe { sys::lcpc_dims_ok(self.ctx.0, n_per_row as u64, n_cols as u64) == 1 }
    }

    fn get_n_col_opens(&self) -> usize {
        unsafe { sys::lcpc_get_n_col_opens(self.ctx.0) as usize }
    }

    fn get_n_degree_tests(&self) -> usize {
        unsafe { sys::lcpc_get_n_degree_tests(self.ctx.0) as usize }
    }
}

impl<Ft, S> LcEncoding for HipSdigEncodingS<Ft, S>
where
    Ft: HipField,
    S: HipSdigCode + std::fmt::Debug + Clone + Sync,
{
    type F = Ft;
    type Err = HipError;

    def_labels!(sdig_pc);

    fn encode<T: AsMut<[Ft]>>(&self, mut inp: T) -> Result<(), HipError> {
        encode_one(&self.ctx, self.n_cols, inp.as_mut())
    }

    fn get_dims(&self, len: usize) -> (usize, usize, usize) {
        ((len + self.n_per_row - 1) / self.n_per_row, self.n_per_row, self.n_cols)
    }

    fn dims_ok(&self, n_per_row: usize, n_cols: usize) -> bool {
        unsafe { sys::lcpc_dims_ok(self.ctx.0, n_per_row as u64, n_cols as u64) == 1 }
    }

    fn get_n_col_opens(&self) -> usize {
        unsafe { sys::lcpc_get_n_col_opens(self.ctx.0) as usize }
    }

    fn get_n_degree_tests(&self) -> usize {
        unsafe { sys::lcpc_get_n_degree_tests(self.ctx.0) as usize }
    }
}

impl<Ft, Rn, Rd> HipEncoding for HipLigeroEncodingRho<Ft, Rn, Rd>
where
    Ft: HipField,
    Rn: Unsigned + std::fmt::Debug + Sync + Clone,
    Rd: Unsigned + std::fmt::Debug + Sync + Clone,
{
    fn raw_ctx(&self) -> *mut sys::lcpc_ctx {
        self.ctx.0
    }
}

impl<Ft, S> HipEncoding for HipSdigEncodingS<Ft, S>
where
    Ft: HipField,
    S: HipSdigCode + std::fmt::Debug + Clone + Sync,
{
    fn raw_ctx(&self) -> *mut sys::lcpc_ctx {
        self.ctx.0
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// merlin::Transcript
// ---------------------------------------------------------------------------------------------------------------------

/// The transcript `prove` / `verify` mutate (`tr: &mut Transcript`, lcpc-2d lib.rs:304-311, 518-527): the library's byte-exact
/// restatement of merlin 2.0 (pinned to merlin's own test vector).  Same three calls, same results as `merlin::Transcript`.
pub struct HipTranscript(*mut sys::lcpc_transcript);
unsafe impl Send for HipTranscript {}

impl HipTranscript {
    /// `Transcript::new(label)`
    pub fn new(label: &'static [u8]) -> Self {
        HipTranscript(unsafe { sys::lcpc_transcript_new(label.as_ptr(), label.len()) })
    }
    /// `append_message(label, message)`
    pub fn append_message(&mut self, label: &'static [u8], message: &[u8]) {
        unsafe { sys::lcpc_transcript_append_message(self.0, label.as_ptr(), label.len(), message.as_ptr(), message.len()) }
    }
    /// `challenge_bytes(label, dest)`
    pub fn challenge_bytes(&mut self, label: &'static [u8], dest: &mut [u8]) {
        unsafe { sys::lcpc_transcript_challenge_bytes(self.0, label.as_ptr(), label.len(), dest.as_mut_ptr(), dest.len()) }
    }
    /// the raw handle, for the `sys` entry points
    pub fn raw(&mut self) -> *mut sys::lcpc_transcript {
        self.0
    }
}
impl Clone for HipTranscript {
    fn clone(&self) -> Self {
        HipTranscript(unsafe { sys::lcpc_transcript_clone(self.0) })
    }
}
impl Drop for HipTranscript {
    fn drop(&mut self) {
        unsafe { sys::lcpc_transcript_free(self.0) }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LcCommit in HBM
// ---------------------------------------------------------------------------------------------------------------------

/// An `LcCommit<Blake3, E>` (lcpc-2d lib.rs:172-184) whose `comm`, `coeffs` and `hashes` stay on the device.  Borrows its
/// encoder like the reference's `commit(&coeffs, &enc)`; many may be live under one encoder; one host thread at a time each.
pub struct HipCommit<'a, E>
where
    E: HipEncoding,
    E::F: HipField,
{
    cm: *mut sys::lcpc_commit_t,
    enc: &'a E,
}

unsafe impl<'a, E> Send for HipCommit<'a, E>
where
    E: HipEncoding,
    E::F: HipField,
{
}

impl<'a, E> Drop for HipCommit<'a, E>
where
    E: HipEncoding,
    E::F: HipField,
{
    fn drop(&mut self) {
        unsafe { sys::lcpc_commit_destroy(self.cm) }
    }
}

impl<'a, E> HipCommit<'a, E>
where
    E: HipEncoding,
    E::F: HipField,
{
    fn detail(&self) -> String {
        cstr(unsafe { sys::lcpc_commit_last_error(self.cm) })
    }

    /// an empty commitment bound to `enc` (`lcpc_commit_create`); the commit methods below fill and refill it
    pub fn empty(enc: &'a E) -> Result<Self, HipError> {
        let mut cm = std::ptr::null_mut();
        match unsafe { sys::lcpc_commit_create(enc.raw_ctx(), &mut cm) } {
            0 => Ok(HipCommit { cm, enc }),
            rc => Err(HipError { status: rc, detail: String::new() }),
        }
    }

    /// `LcCommit::<Blake3, E>::commit(&coeffs, &enc)` (lcpc-2d lib.rs:299-301 -> 622-671) from host memory
    pub fn commit(coeffs: &[E::F], enc: &'a E) -> Result<Self, ProverError<HipError>> {
        let me = Self::empty(enc).map_err(ProverError::Encode)?;
        me.recommit(coeffs)?;
        Ok(me)
    }

    /// commit again into this object (the buffers are reused: what a benchmark loop wants)
    pub fn recommit(&self, coeffs: &[E::F]) -> Result<(), ProverError<HipError>> {
        match unsafe { sys::lcpc_commit(self.cm, limbs_of(coeffs), coeffs.len() as u64, std::ptr::null_mut()) } {
            0 => Ok(()),
            rc => Err(prover_error(rc, self.detail())),
        }
    }

    /// the same with the coefficients already resident in HBM: `coeffs_dev` is a HIP device pointer to `n_coeffs` elements,
    /// the work is enqueued on `stream` (a `hipStream_t`) and -- with `want_root == false` -- the call returns without
    /// synchronising.  `borrow`: `LcCommit.coeffs` aliases the caller's buffer instead of copying it.
    ///
    /// # Safety
    /// `coeffs_dev` must be a valid device allocation of `n_coeffs` elements on the encoder's device, and with `borrow` must
    /// stay valid and unchanged while the commitment is live.
    pub unsafe fn recommit_device(
        &self,
        coeffs_dev: *const u64,
        n_coeffs: usize,
        stream: *mut c_void,
        borrow: bool,
        want_root: bool,
    ) -> Result<Option<Output<Blake3>>, ProverError<HipError>> {
        let mut root = [0u8; 32];
        let flags = if borrow { sys::LCPC_COMMIT_BORROW_COEFFS } else { 0 };
        let rp = if want_root { root.as_mut_ptr() } else { std::ptr::null_mut() };
        match sys::lcpc_commit_device(self.cm, coeffs_dev, n_coeffs as u64, stream, flags, rp) {
            0 => Ok(if want_root { Some(Output::<Blake3>::clone_from_slice(&root)) } else { None }),
            rc => Err(prover_error(rc, self.detail())),
        }
    }

    /// a row shard of a commitment with the exchange on RCCL inside the library (`lcpc_commit_sharded_device`): this rank's
    /// rows (`shard_layout` says which) at `coeffs_local_dev`.  Collective; the encoder needs `comm_init` first.
    ///
    /// # Safety
    /// as [`HipCommit::recommit_device`].  `async_tail` (`LCPC_COMMIT_ASYNC_TAIL`): exchange, leaf digests and tree run on the
    /// commitment's own stream and `stream` is free after the local column hash -- fill a second `HipCommit` of the same encoder
    /// next and its encode overlaps this one's wire time.
    pub unsafe fn recommit_sharded_device(
        &self,
        coeffs_local_dev: *const u64,
        n_rows_total: usize,
        stream: *mut c_void,
        borrow: bool,
        async_tail: bool,
        want_root: bool,
    ) -> Result<Option<Output<Blake3>>, ProverError<HipError>> {
        let mut root = [0u8; 32];
        let flags = (if borrow { sys::LCPC_COMMIT_BORROW_COEFFS } else { 0 }) | (if async_tail { sys::LCPC_COMMIT_ASYNC_TAIL } else { 0 });
        let rp = if want_root { root.as_mut_ptr() } else { std::ptr::null_mut() };
        match sys::lcpc_commit_sharded_device(self.cm, coeffs_local_dev, n_rows_total as u64, stream, flags, rp) {
            0 => Ok(if want_root { Some(Output::<Blake3>::clone_from_slice(&root)) } else { None }),
            rc => Err(prover_error(rc, self.detail())),
        }
    }

    /// `comm.get_root()` (lcpc-2d lib.rs:276-281): the reference's own `LcRoot`, whose fields are private -- built through its
    /// bincode form (lib.rs:373-398: an 8-byte length and the 32 digest bytes, `lcpc_root_bincode`).  `root.as_ref()` is the
    /// `&Output<D>` that `verify` takes, `root.into_raw()` the digest, exactly as with the reference's commitment.
    pub fn get_root(&self) -> LcRoot<Blake3, E> {
        let raw = self.get_root_raw();
        let mut wire = [0u8; 40];
        unsafe { sys::lcpc_root_bincode(raw.as_ptr(), wire.as_mut_ptr()) };
        bincode::deserialize(&wire).expect("lcpc_root_bincode writes the bincode layout of LcRoot")
    }

    /// the Merkle root as the raw digest (what `get_root().into_raw()` gives, without the detour)
    pub fn get_root_raw(&self) -> Output<Blake3> {
        let mut r = [0u8; 32];
        let rc = unsafe { sys::lcpc_get_root(self.cm, r.as_mut_ptr()) };
        assert_eq!(rc, 0, "get_root on an empty commitment");
        Output::<Blake3>::clone_from_slice(&r)
    }

    /// `check_comm` (lcpc-2d lib.rs:673-688): the commitment's fields and `enc` belong together -- here: `enc` is (a clone of)
    /// the encoder the device buffers were filled under
    fn check_comm(&self, enc: &E) -> Result<(), ProverError<HipError>> {
        if enc.raw_ctx() == self.enc.raw_ctx() {
            Ok(())
        } else {
            Err(ProverError::Commit)
        }
    }

    fn dims(&self) -> (usize, usize, usize) {
        let (mut nr, mut np, mut nc, mut nh) = (0u64, 0u64, 0u64, 0u64);
        let rc = unsafe { sys::lcpc_commit_dims(self.cm, &mut nr, &mut np, &mut nc, &mut nh) };
        assert_eq!(rc, 0, "dims of an empty commitment");
        (nr as usize, np as usize, nc as usize)
    }
    /// `get_n_rows` (lcpc-2d lib.rs:293-296)
    pub fn get_n_rows(&self) -> usize {
        self.dims().0
    }
    /// `get_n_per_row` (lcpc-2d lib.rs:283-286)
    pub fn get_n_per_row(&self) -> usize {
        self.dims().1
    }
    /// `get_n_cols` (lcpc-2d lib.rs:288-291)
    pub fn get_n_cols(&self) -> usize {
        self.dims().2
    }

    /// `comm.prove(&outer_tensor, &enc, &mut tr)` (lcpc-2d lib.rs:304-311 -> 1004-1093), argument for argument: the proof is the
    /// reference's own type, obtained through its bincode form (lib.rs:597-609).  `enc` must be the encoder the commitment was
    /// made with -- the reference's `check_comm(comm, enc)` (lib.rs:1015) -- else `ProverError::Commit`.
    pub fn prove(&self, outer_tensor: &[E::F], enc: &E, tr: &mut HipTranscript) -> Result<LcEvalProof<Blake3, E>, ProverError<HipError>> {
        self.check_comm(enc)?;
        let bytes = self.prove_bytes(outer_tensor, tr, false)?;
        Ok(bincode::deserialize(&bytes).expect("lcpc_prove returns the bincode layout of LcEvalProof"))
    }

    /// the same on a row-sharded commitment, the three all-gathers on RCCL (`lcpc_prove_sharded_rccl`); every rank returns the
    /// unsharded proof byte for byte
    pub fn prove_sharded(&self, outer_tensor: &[E::F], enc: &E, tr: &mut HipTranscript) -> Result<LcEvalProof<Blake3, E>, ProverError<HipError>> {
        self.check_comm(enc)?;
        let bytes = self.prove_bytes(outer_tensor, tr, true)?;
        Ok(bincode::deserialize(&bytes).expect("lcpc_prove_sharded_rccl returns the bincode layout of LcEvalProof"))
    }

    /// the proof as `bincode::serialize(&proof)` bytes
    pub fn prove_bytes(&self, outer_tensor: &[E::F], tr: &mut HipTranscript, sharded: bool) -> Result<Vec<u8>, ProverError<HipError>> {
        let (mut p, mut n) = (std::ptr::null_mut::<u8>(), 0u64);
        let (t, nt) = (limbs_of(outer_tensor), outer_tensor.len() as u64);
        let rc = unsafe {
            if sharded {
                sys::lcpc_prove_sharded_rccl(self.cm, t, nt, tr.raw(), &mut p, &mut n, std::ptr::null_mut())
            } else {
                sys::lcpc_prove(self.cm, t, nt, tr.raw(), &mut p, &mut n, std::ptr::null_mut())
            }
        };
        if rc != 0 {
            return Err(prover_error(rc, self.detail()));
        }
        let bytes = unsafe { std::slice::from_raw_parts(p, n as usize) }.to_vec();
        unsafe { sys::lcpc_free(p as *mut c_void) };
        Ok(bytes)
    }

    /// `collapse_columns` (lcpc-2d lib.rs:1095-1123) for one tensor: `poly[j] = sum_r coeffs[r][j] * tensor[r]`
    pub fn collapse_columns(&self, tensor: &[E::F]) -> Result<Vec<E::F>, ProverError<HipError>> {
        let mut poly = vec![<E::F as ff::Field>::zero(); self.get_n_per_row()];
        match unsafe { sys::lcpc_collapse(self.cm, limbs_of(tensor), 1, limbs_of_mut(&mut poly)) } {
            0 => Ok(poly),
            rc => Err(prover_error(rc, self.detail())),
        }
    }

    /// hand the whole commitment to the reference: the bytes `bincode::deserialize::<LcCommit<Blake3, _>>` reads
    /// (lcpc-2d lib.rs:186-268), streamed to `w` in pieces of at most 64 MiB
    pub fn serialize_into<W: std::io::Write>(&self, w: &mut W) -> Result<(), HipError> {
        unsafe extern "C" fn sink<W: std::io::Write>(user: *mut c_void, data: *const u8, len: u64) -> c_int {
            // a panic in `W::write_all` must not unwind through the C++ frames of the library (undefined behaviour): it becomes
            // the callback's failure code, and the library returns LCPC_ERR_ARG to `serialize_into`
            let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| {
                let w = &mut *(user as *mut W);
                w.write_all(std::slice::from_raw_parts(data, len as usize)).is_err()
            }));
            match r {
                Ok(failed) => failed as c_int,
                Err(_) => 1,
            }
        }
        match unsafe { sys::lcpc_commit_bincode_write(self.cm, Some(sink::<W>), w as *mut W as *mut c_void) } {
            0 => Ok(()),
            rc => Err(HipError { status: rc, detail: self.detail() }),
        }
    }

    /// take a commitment the reference serialised (`bincode::serialize_into(&mut file, &comm)`) into HBM; the Merkle tree is
    /// rebuilt from `comm` on the device and must agree with the stream's `hashes`
    pub fn deserialize_from<R: std::io::Read>(enc: &'a E, r: &mut R) -> Result<Self, ProverError<HipError>> {
        unsafe extern "C" fn source<R: std::io::Read>(user: *mut c_void, data: *mut u8, len: u64) -> c_int {
            // (no unwinding across the C frames: see `serialize_into`)
            let res = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| {
                let r = &mut *(user as *mut R);
                r.read_exact(std::slice::from_raw_parts_mut(data, len as usize)).is_err()
            }));
            match res {
                Ok(failed) => failed as c_int,
                Err(_) => 1,
            }
        }
        let me = Self::empty(enc).map_err(ProverError::Encode)?;
        match unsafe { sys::lcpc_commit_from_bincode(me.cm, Some(source::<R>), r as *mut R as *mut c_void, std::ptr::null_mut()) } {
            0 => Ok(me),
            rc => Err(prover_error(rc, me.detail())),
        }
    }

    /// the encoder this commitment was made with
    pub fn encoding(&self) -> &'a E {
        self.enc
    }
}

/// `proof.verify(root, outer, inner, &enc, &mut tr)` (lcpc-2d lib.rs:518-527 -> 832-952) run by the library on the proof's
/// bincode bytes: the row encodes and column checks on the GPU.  The reference's own `verify` on the deserialised proof, with
/// the same encoder, gives the same answer (it calls `enc.encode` once per row).
pub fn verify_on_device<E>(
    proof_bytes: &[u8],
    root: &Output<Blake3>,
    outer_tensor: &[E::F],
    inner_tensor: &[E::F],
    enc: &E,
    tr: &mut HipTranscript,
) -> Result<E::F, VerifierError<HipError>>
where
    E: HipEncoding,
    E::F: HipField,
{
    let mut out = [0u64; 4];
    let rc = unsafe {
        sys::lcpc_verify(
            enc.raw_ctx(),
            root.as_ptr(),
            limbs_of(outer_tensor),
            outer_tensor.len() as u64,
            limbs_of(inner_tensor),
            inner_tensor.len() as u64,
            proof_bytes.as_ptr(),
            proof_bytes.len() as u64,
            tr.raw(),
            out.as_mut_ptr(),
        )
    };
    if rc != 0 {
        return Err(verifier_error(rc, cstr(unsafe { sys::lcpc_last_error(enc.raw_ctx()) })));
    }
    // L Montgomery limbs are the element itself (HipField's contract)
    Ok(unsafe { std::ptr::read(out.as_ptr() as *const E::F) })
}

#[cfg(test)]
mod tests {
    //! the reference's end-to-end test (lcpc-ligero-pc/src/tests.rs:100-177) with the GPU encoder in place of `LigeroEncoding`,
    //! and the cross-check that makes it a drop-in: same root and same proof bytes as the reference's own `commit` / `prove`.
    use super::*;
    use ff::Field;
    use lcpc_ligero_pc::{LigeroCommit, LigeroEncoding};
    use lcpc_test_fields::ft255::Ft255;
    use merlin::Transcript;

    fn powers(x: Ft255, n: usize, step: usize) -> Vec<Ft255> {
        let xs = x.pow_vartime(&[step as u64]);
        std::iter::successors(Some(Ft255::one()), |p| Some(*p * xs)).take(n).collect()
    }

    #[test]
    fn commit_prove_verify_equals_reference() {
        let len = 1usize << 16;
        let mut rng = rand::thread_rng();
        let coeffs: Vec<Ft255> = std::iter::repeat_with(|| Ft255::random(&mut rng)).take(len).collect();

        let enc = HipLigeroEncoding::<Ft255>::new(len);
        let ref_enc = LigeroEncoding::<Ft255>::new(len);
        assert_eq!(enc.get_dims(len), ref_enc.get_dims(len));
        assert_eq!(enc.get_n_col_opens(), ref_enc.get_n_col_opens());
        assert_eq!(enc.get_n_degree_tests(), ref_enc.get_n_degree_tests());

        let comm = HipCommit::commit(&coeffs, &enc).unwrap();
        let ref_comm = LigeroCommit::<Blake3, Ft255>::commit(&coeffs, &ref_enc).unwrap();
        let root = comm.get_root(); // an LcRoot<Blake3, _>, as the reference's
        assert_eq!(root.as_ref(), ref_comm.get_root().as_ref());
        assert_eq!(bincode::serialize(&root).unwrap(), bincode::serialize(&ref_comm.get_root()).unwrap());

        let x = Ft255::random(&mut rng);
        let (nr, np, _) = enc.get_dims(len);
        let inner = powers(x, np, 1);
        let outer = powers(x, nr, np);

        let mut tr = HipTranscript::new(b"test transcript");
        tr.append_message(b"polycommit", root.as_ref());
        let pf_bytes = comm.prove_bytes(&outer, &mut tr, false).unwrap();

        let mut rtr = Transcript::new(b"test transcript");
        rtr.append_message(b"polycommit", root.as_ref());
        let ref_pf = ref_comm.prove(&outer, &ref_enc, &mut rtr).unwrap();
        assert_eq!(pf_bytes, bincode::serialize(&ref_pf).unwrap());

        // the reference's verifier on the GPU prover's proof, with the reference's encoder ...
        let pf: lcpc_ligero_pc::LigeroEvalProof<Blake3, Ft255> = bincode::deserialize(&pf_bytes).unwrap();
        let mut vtr = Transcript::new(b"test transcript");
        vtr.append_message(b"polycommit", root.as_ref());
        let ev = pf.verify(root.as_ref(), &outer, &inner, &ref_enc, &mut vtr).unwrap();
        // ... and the library's verifier: same evaluation
        let mut htr = HipTranscript::new(b"test transcript");
        htr.append_message(b"polycommit", root.as_ref());
        assert_eq!(ev, verify_on_device(&pf_bytes, root.as_ref(), &outer, &inner, &enc, &mut htr).unwrap());
        // and the typed form, argument for argument the reference's call: the same proof
        let mut tr2 = HipTranscript::new(b"test transcript");
        tr2.append_message(b"polycommit", root.as_ref());
        let pf2 = comm.prove(&outer, &enc, &mut tr2).unwrap();
        assert_eq!(bincode::serialize(&pf2).unwrap(), pf_bytes);
        // check_comm: another encoder than the commitment's is refused as the reference refuses inconsistent fields
        let other = HipLigeroEncoding::<Ft255>::new(len);
        assert!(matches!(comm.prove(&outer, &other, &mut tr2), Err(ProverError::Commit)));
    }
}
