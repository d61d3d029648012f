//! latticefold-hip: the MI355X prover hot path behind the trait surface of `crates/latticefold`.
//!
//! The reference forbids `unsafe` (`crates/latticefold/src/lib.rs:4`), so the FFI lives in `latticefold-hip-sys` and this crate is the safe
//! layer a caller of the reference switches to:
//!
//! | reference (file:line)                                                         | here                                   |
//! |-------------------------------------------------------------------------------|----------------------------------------|
//! | `AjtaiCommitmentScheme::{new, rand, commit, commit_ntt, kappa, width}` `commitment/commitment_scheme.rs:17-77` | [`HipAjtai`]            |
//! | `Witness::{from_w_ccs, from_f, from_f_coeff, commit}` `arith.rs:230-362`        | [`HipWitness`] (device resident)       |
//! | `PoseidonTranscript` (`Transcript`, `TranscriptWithShortChallenges`) `transcript/poseidon.rs:17-75` | [`HipTranscript`] -- same traits, same challenges |
//! | `LinearizationProver::prove` `nifs/linearization.rs:26-52`                      | `impl LinearizationProver for HipLinearizationProver` |
//! | `DecompositionProver::prove` `nifs/decomposition/structs.rs:48-64`              | `impl DecompositionProver for HipDecompositionProver` |
//! | `FoldingProver::prove` `nifs/folding/structs.rs:43-70`                          | `impl FoldingProver for HipFoldingProver` |
//! | `NIFSProver::prove` `nifs.rs:48-103`                                            | [`HipNIFSProver::prove`] (same signature) and [`HipNIFSProver::prove_resident`] |
//!
//! Two ways in.  (1) The trait methods take the reference's host types (`&Witness<NTT>`, `&CCS<NTT>`, `&AjtaiCommitmentScheme<NTT>`); every call marshals them
//! (Montgomery -> canonical words: `into_bigint`, O(N), outside the kernels) into the session installed with [`HipSession::install`] -- the CCS and the Ajtai
//! matrix are uploaded once, witnesses per call.  (2) A prover that folds in a loop keeps [`HipWitness`] handles and calls [`HipNIFSProver::prove_resident`]:
//! nothing but the instance (kappa + l ring elements) and the proof crosses PCIe.
//!
//! The library runs the Fiat-Shamir transcript itself (host Poseidon next to the GPU work), so the transcript handed to a prover must be a [`HipTranscript`]:
//! the trait methods are generic over `impl Transcript<NTT>` and recognise it by type; any other transcript type gets `Err(..)`/panic with that message.
//! `HipTranscript` implements the reference's two transcript traits over the same sponge (pinned by the reference's Poseidon and challenge KATs), so the
//! reference's *verifier* runs on it unchanged.
//!
//! NOT COMPILED in the build image (no Rust toolchain, no network for the git dependency stark-rings): written against the public API of the reference at
//! /root/reference and of stark-rings@886a89f as the reference uses it; `tests/test_rust_wrapper_cpu.py` checks names and arities against the reference sources.
#![forbid(unsafe_op_in_unsafe_fn)]

use core::marker::PhantomData;
use std::sync::{Arc, Mutex, OnceLock};

use ark_ff::{Field, PrimeField};
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize};
use cyclotomic_rings::{challenge_set::LatticefoldChallengeSet, rings::SuitableRing};
use latticefold::{
    arith::{Witness, CCCS, CCS, LCCCS},
    commitment::{AjtaiCommitmentScheme, Commitment, CommitmentError},
    decomposition_parameters::DecompositionParams,
    nifs::{
        decomposition::{DecompositionProof, DecompositionProver},
        error::{DecompositionError, FoldingError, LatticefoldError, LinearizationError},
        folding::{FoldingProof, FoldingProver},
        linearization::{LinearizationProof, LinearizationProver},
        LFProof,
    },
    transcript::{Transcript, TranscriptWithShortChallenges},
    utils::sumcheck::Proof as SumcheckProof,
};
use latticefold_hip_sys as sys;
use stark_rings::{
    balanced_decomposition::{recompose, DecomposeToVec},
    PolyRing,
};
use stark_rings_linalg::ops::Transpose;
use stark_rings_poly::mle::DenseMultilinearExtension;

// ------------------------------------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------------------------------------
#[derive(Debug, thiserror::Error)]
pub enum HipError {
    #[error("liblfhip: {0} ({1})")]
    Lib(String, i32),
    #[error("the transcript handed to a Hip* prover must be a latticefold_hip::HipTranscript (the library runs the Fiat-Shamir schedule itself)")]
    ForeignTranscript,
    #[error("no session: call HipSession::install(&ccs, &scheme) first")]
    NoSession,
    #[error("ring {0} is not one the library implements (GoldilocksRingNTT, BabyBearRingNTT)")]
    Ring(&'static str),
}

fn chk(rc: i32, what: &str) -> Result<(), HipError> {
    if rc == sys::LF_OK {
        return Ok(());
    }
    // SAFETY: lf_strerror returns a pointer to a static NUL-terminated string for every code
    let msg = unsafe { std::ffi::CStr::from_ptr(sys::lf_strerror(rc)) }.to_string_lossy().into_owned();
    Err(HipError::Lib(format!("{what}: {msg}"), rc))
}

// ------------------------------------------------------------------------------------------------------------------------------------
// marshalling: ring elements <-> canonical little-endian u64 words (NTT form slot-major, each slot its tau base-field coordinates -- the order
// `coeffs().iter().flat_map(to_base_prime_field_elements)` yields, transcript/poseidon.rs:40-47; coefficient form X^0 .. X^(d-1))
// ------------------------------------------------------------------------------------------------------------------------------------
pub trait RingWords: Sized {
    /// u64 words per element (24 Goldilocks, 72 BabyBear)
    const WORDS: usize;
    fn to_words(&self, out: &mut Vec<u64>);
    fn from_words(w: &[u64]) -> Self;
}

fn fp_to_u64<F: PrimeField>(x: &F) -> u64 {
    x.into_bigint().as_ref()[0] // both primes are below 2^64: one limb
}

impl<R: PolyRing> RingWords for R
where
    R::BaseRing: Field,
    <R::BaseRing as Field>::BasePrimeField: PrimeField,
{
    const WORDS: usize = R::dimension() * (<R::BaseRing as Field>::extension_degree() as usize);

    fn to_words(&self, out: &mut Vec<u64>) {
        for c in self.coeffs() {
            for f in c.to_base_prime_field_elements() {
                out.push(fp_to_u64(&f));
            }
        }
    }

    fn from_words(w: &[u64]) -> Self {
        let tau = <R::BaseRing as Field>::extension_degree() as usize;
        let coeffs: Vec<R::BaseRing> = w
            .chunks(tau)
            .map(|c| {
                let fs: Vec<_> = c.iter().map(|&x| <<R::BaseRing as Field>::BasePrimeField as From<u64>>::from(x)).collect();
                <R::BaseRing as Field>::from_base_prime_field_elems(&fs).expect("tau coordinates")
            })
            .collect();
        R::from(coeffs)
    }
}

fn flatten<R: RingWords>(v: &[R]) -> Vec<u64> {
    let mut out = Vec::with_capacity(v.len() * R::WORDS);
    for x in v {
        x.to_words(&mut out);
    }
    out
}

fn unflatten<R: RingWords>(w: &[u64]) -> Vec<R> {
    w.chunks(R::WORDS).map(R::from_words).collect()
}

fn ring_id<NTT: SuitableRing>() -> Result<i32, HipError> {
    match <NTT as RingWords>::WORDS {
        24 => Ok(sys::LF_RING_GOLDILOCKS),
        72 => Ok(sys::LF_RING_BABYBEAR),
        _ => Err(HipError::Ring(core::any::type_name::<NTT>())),
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// context / session
// ------------------------------------------------------------------------------------------------------------------------------------
/// One GPU context (`lf_ctx`): owns the device copy of the CCS, the Ajtai matrix (int8 byte planes) and every scratch buffer.  Thread-safe per context
/// (the library serialises calls on an internal mutex).
pub struct HipContext {
    raw: *mut sys::lf_ctx,
    params: Mutex<Option<sys::lf_params>>,
}
// SAFETY: the C library guards every entry point of a context with its own mutex (include/lfhip.h, "Threading")
unsafe impl Send for HipContext {}
unsafe impl Sync for HipContext {}

impl HipContext {
    pub fn new<NTT: SuitableRing>(device: i32) -> Result<Arc<Self>, HipError> {
        let mut raw = core::ptr::null_mut();
        // SAFETY: out-pointer to a local; the library fills it on success only
        chk(unsafe { sys::lf_ctx_create_ring(&mut raw, device, ring_id::<NTT>()?) }, "lf_ctx_create_ring")?;
        Ok(Arc::new(Self { raw, params: Mutex::new(None) }))
    }

    /// `CCS` (arith.rs:50-74) + `DecompositionParams` -> the device copy (CSR matrices, multisets, constants)
    pub fn load_ccs<NTT: SuitableRing, P: DecompositionParams>(&self, ccs: &CCS<NTT>, kappa: usize) -> Result<(), HipError> {
        let p = sys::lf_params {
            s: ccs.s as u32,
            wit_len: (ccs.n - ccs.l - 1) as u32,
            l: ccs.l as u32,
            L: P::L as u32,
            K: P::K as u32,
            b: P::B_SMALL as u32,
            B: P::B as u64,
            kappa: kappa as u32,
            t: ccs.t as u32,
            q: ccs.q as u32,
            d: ccs.d as u32,
        };
        let (mut rowptrs, mut cols, mut vals) = (Vec::new(), Vec::new(), Vec::new());
        for m in &ccs.M {
            let (mut rp, mut ci, mut vv) = (vec![0u32], Vec::new(), Vec::new());
            for row in &m.coeffs {
                for (v, c) in row {
                    ci.push(*c as u32);
                    v.to_words(&mut vv);
                }
                rp.push(ci.len() as u32);
            }
            rp.resize(ccs.m + 1, *rp.last().unwrap()); // rows past the matrix are empty (pad_rows_to)
            rowptrs.push(rp);
            cols.push(ci);
            vals.push(vv);
        }
        let (mut s_off, mut s_idx) = (vec![0u32], Vec::new());
        for set in &ccs.S {
            s_idx.extend(set.iter().map(|&j| j as u32));
            s_off.push(s_idx.len() as u32);
        }
        let c = flatten(&ccs.c);
        let rp: Vec<*const u32> = rowptrs.iter().map(|v| v.as_ptr()).collect();
        let cp: Vec<*const u32> = cols.iter().map(|v| v.as_ptr()).collect();
        let vp: Vec<*const u64> = vals.iter().map(|v| v.as_ptr()).collect();
        // SAFETY: every pointer refers to a live Vec of the length the header documents for it; the library copies before returning
        chk(unsafe { sys::lf_ccs_load(self.raw, &p, rp.as_ptr(), cp.as_ptr(), vp.as_ptr(), s_off.as_ptr(), s_idx.as_ptr(), c.as_ptr()) }, "lf_ccs_load")?;
        *self.params.lock().unwrap() = Some(p);
        Ok(())
    }

    fn params(&self) -> Result<sys::lf_params, HipError> {
        self.params.lock().unwrap().ok_or(HipError::NoSession)
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        // SAFETY: raw came from lf_ctx_create_ring and is destroyed exactly once
        unsafe { sys::lf_ctx_destroy(self.raw) }
    }
}

/// The (CCS, Ajtai matrix) a process folds under, installed once; the trait impls below (whose signatures carry `&CCS` and `&AjtaiCommitmentScheme` per call,
/// as the reference's do) use it instead of re-uploading 5 GB per step.  Identity is checked by shape (m, n, t, kappa, width).
pub struct HipSession {
    pub ctx: Arc<HipContext>,
    shape: (usize, usize, usize, usize, usize),
}
static SESSION: OnceLock<Mutex<Option<Arc<HipSession>>>> = OnceLock::new();

impl HipSession {
    pub fn install<NTT: SuitableRing, P: DecompositionParams>(device: i32, ccs: &CCS<NTT>, scheme: &HipAjtai<NTT>) -> Result<Arc<Self>, HipError> {
        scheme.ctx.load_ccs::<NTT, P>(ccs, scheme.kappa())?;
        let _ = device;
        let s = Arc::new(Self { ctx: scheme.ctx.clone(), shape: (ccs.m, ccs.n, ccs.t, scheme.kappa(), scheme.width()) });
        *SESSION.get_or_init(|| Mutex::new(None)).lock().unwrap() = Some(s.clone());
        Ok(s)
    }

    fn current() -> Result<Arc<Self>, HipError> {
        SESSION.get_or_init(|| Mutex::new(None)).lock().unwrap().clone().ok_or(HipError::NoSession)
    }

    fn check<NTT: SuitableRing>(&self, ccs: &CCS<NTT>) -> Result<(), HipError> {
        if (ccs.m, ccs.n, ccs.t) == (self.shape.0, self.shape.1, self.shape.2) { Ok(()) } else { Err(HipError::NoSession) }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// AjtaiCommitmentScheme (commitment/commitment_scheme.rs:17-114)
// ------------------------------------------------------------------------------------------------------------------------------------
/// `AjtaiCommitmentScheme<NTT>` with the matrix on the device (int8 byte planes in MFMA operand order; `lf_ajtai_load`).
pub struct HipAjtai<NTT> {
    ctx: Arc<HipContext>,
    kappa: usize,
    n: usize,
    _r: PhantomData<NTT>,
}

impl<NTT: SuitableRing> HipAjtai<NTT> {
    /// `AjtaiCommitmentScheme::new(matrix)`: row-major kappa x n ring elements (NTT form)
    pub fn new(ctx: Arc<HipContext>, matrix: &[Vec<NTT>]) -> Result<Self, HipError> {
        let (kappa, n) = (matrix.len(), matrix.first().map_or(0, |r| r.len()));
        let mut words = Vec::with_capacity(kappa * n * NTT::WORDS);
        for row in matrix {
            for x in row {
                x.to_words(&mut words);
            }
        }
        // SAFETY: words holds kappa * n elements; the library copies
        chk(unsafe { sys::lf_ajtai_load(ctx.raw, words.as_ptr(), kappa, n) }, "lf_ajtai_load")?;
        Ok(Self { ctx, kappa, n, _r: PhantomData })
    }

    /// `AjtaiCommitmentScheme::rand(kappa, n, rng)`: an i.i.d. uniform matrix generated ON the device from `seed` (SplitMix64; the reference's `rand` fills
    /// every entry of a row with one element -- SURVEY 3.3 -- which nothing depends on)
    pub fn rand(ctx: Arc<HipContext>, kappa: usize, n: usize, seed: u64) -> Result<Self, HipError> {
        // SAFETY: plain scalars
        chk(unsafe { sys::lf_ajtai_generate(ctx.raw, seed, kappa, n) }, "lf_ajtai_generate")?;
        Ok(Self { ctx, kappa, n, _r: PhantomData })
    }

    /// `commit` / `commit_ntt` (commitment_scheme.rs:37-54,75-77): the int8 general commit (lf_ajtai_i8g.hip)
    pub fn commit_ntt(&self, f: &[NTT]) -> Result<Commitment<NTT>, CommitmentError> {
        if f.len() != self.n {
            return Err(CommitmentError::WrongWitnessLength(f.len(), self.n));
        }
        let w = flatten(f);
        let mut out = vec![0u64; self.kappa * NTT::WORDS];
        // SAFETY: w holds n elements, out kappa elements
        let rc = unsafe { sys::lf_ajtai_commit(self.ctx.raw, w.as_ptr(), self.n, 1, out.as_mut_ptr()) };
        if rc != sys::LF_OK {
            return Err(CommitmentError::WrongWitnessLength(f.len(), self.n));
        }
        Ok(Commitment::from(unflatten(&out)))
    }

    pub fn commit(&self, f: &[NTT]) -> Result<Commitment<NTT>, CommitmentError> {
        self.commit_ntt(f)
    }

    pub fn kappa(&self) -> usize {
        self.kappa
    }

    pub fn width(&self) -> usize {
        self.n
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Witness (arith.rs:213-386)
// ------------------------------------------------------------------------------------------------------------------------------------
/// A witness resident on the device: centred int32 coefficient planes of `f_coeff` (+ `f`, `w_ccs` once a fold step built them); f-hat stays virtual.
pub struct HipWitness<NTT> {
    ctx: Arc<HipContext>,
    raw: *mut sys::lf_witness,
    _r: PhantomData<NTT>,
}
// SAFETY: the handle is only ever passed to entry points that lock the owning context
unsafe impl<NTT> Send for HipWitness<NTT> {}

impl<NTT: SuitableRing> HipWitness<NTT> {
    fn make(ctx: &Arc<HipContext>, words: &[u64], f: unsafe extern "C" fn(*mut sys::lf_ctx, *const u64, *mut *mut sys::lf_witness) -> i32, what: &str) -> Result<Self, HipError> {
        let mut raw = core::ptr::null_mut();
        // SAFETY: words has the length the entry point documents (checked by the callers below against the loaded parameters)
        chk(unsafe { f(ctx.raw, words.as_ptr(), &mut raw) }, what)?;
        Ok(Self { ctx: ctx.clone(), raw, _r: PhantomData })
    }

    /// `Witness::from_w_ccs::<P>(w_ccs)` (arith.rs:230-248): ICRT, gadget decomposition, on the device
    pub fn from_w_ccs(ctx: &Arc<HipContext>, w_ccs: &[NTT]) -> Result<Self, HipError> {
        assert_eq!(w_ccs.len(), ctx.params()?.wit_len as usize);
        Self::make(ctx, &flatten(w_ccs), sys::lf_witness_from_w_ccs, "lf_witness_from_w_ccs")
    }

    /// The same witness built NEXT TO a running fold step (a chain's next instance): the PCIe upload, the ICRT and the gadget decomposition run on a worker
    /// thread and the context's lowest-priority stream; [`HipPendingWitness::wait`] hands the witness over.
    pub fn from_w_ccs_begin(ctx: &Arc<HipContext>, w_ccs: &[NTT]) -> Result<HipPendingWitness<NTT>, HipError> {
        assert_eq!(w_ccs.len(), ctx.params()?.wit_len as usize);
        let words = flatten(w_ccs).into_boxed_slice();
        let mut job = core::ptr::null_mut();
        // SAFETY: `words` holds wit_len elements and lives in the returned value until the job has finished
        chk(unsafe { sys::lf_witness_from_w_ccs_begin(ctx.raw, words.as_ptr(), &mut job) }, "lf_witness_from_w_ccs_begin")?;
        Ok(HipPendingWitness { ctx: ctx.clone(), job, _words: words, _r: PhantomData })
    }

    /// `Witness::from_f::<P>(f)` (arith.rs:299-313)
    pub fn from_f(ctx: &Arc<HipContext>, f: &[NTT]) -> Result<Self, HipError> {
        Self::make(ctx, &flatten(f), sys::lf_witness_from_f, "lf_witness_from_f")
    }

    /// `Witness::from_f_coeff::<P>(f_coeff)` (arith.rs:324-338)
    pub fn from_f_coeff(ctx: &Arc<HipContext>, f_coeff: &[NTT::CoefficientRepresentation]) -> Result<Self, HipError> {
        Self::make(ctx, &flatten(f_coeff), sys::lf_witness_from_f_coeff, "lf_witness_from_f_coeff")
    }

    /// upload of a reference `Witness` (its `f_coeff`; `f`, `w_ccs`, `f_hat` are functions of it)
    pub fn from_reference(ctx: &Arc<HipContext>, w: &Witness<NTT>) -> Result<Self, HipError> {
        Self::from_f_coeff(ctx, &w.f_coeff)
    }

    /// `Witness::commit::<P>(&scheme)` (arith.rs:357-362): from the int32 planes, five base-128 digit planes on the matrix cores
    pub fn commit(&self, scheme: &HipAjtai<NTT>) -> Result<Commitment<NTT>, CommitmentError> {
        let mut out = vec![0u64; scheme.kappa() * NTT::WORDS];
        // SAFETY: out holds kappa elements
        let rc = unsafe { sys::lf_witness_commit(self.ctx.raw, self.raw, out.as_mut_ptr()) };
        if rc != sys::LF_OK {
            return Err(CommitmentError::WrongWitnessLength(0, scheme.width()));
        }
        Ok(Commitment::from(unflatten(&out)))
    }

    /// download as a reference `Witness` (`f_coeff` from the device; the rest through the reference's own constructor)
    pub fn to_reference<P: DecompositionParams>(&self) -> Result<Witness<NTT>, HipError> {
        let p = self.ctx.params()?;
        let n = (p.wit_len * p.L) as usize;
        let mut w = vec![0u64; n * <NTT::CoefficientRepresentation as RingWords>::WORDS];
        // SAFETY: w holds N coefficient-form elements
        chk(unsafe { sys::lf_witness_get_f_coeff(self.ctx.raw, self.raw, w.as_mut_ptr()) }, "lf_witness_get_f_coeff")?;
        Ok(Witness::from_f_coeff::<P>(unflatten(&w)))
    }

    /// `Witness::within_bound` (arith.rs:372-386) with centred representatives: max |coefficient| < bound
    pub fn within_bound(&self, bound: u64) -> Result<bool, HipError> {
        let p = self.ctx.params()?;
        let n = (p.wit_len * p.L) as usize;
        let mut f = vec![0u64; n * NTT::WORDS];
        let (mut ok, mut mx) = (0i32, 0u64);
        // SAFETY: f holds N elements; ok / mx are plain out-parameters
        unsafe {
            chk(sys::lf_witness_get_f(self.ctx.raw, self.raw, f.as_mut_ptr()), "lf_witness_get_f")?;
            chk(sys::lf_linf_check(self.ctx.raw, f.as_ptr(), n, bound, 0, &mut ok, &mut mx), "lf_linf_check")?;
        }
        Ok(ok != 0)
    }
}

/// A witness being ingested next to a fold step ([`HipWitness::from_w_ccs_begin`]); owns the host words the job reads.
pub struct HipPendingWitness<NTT> {
    ctx: Arc<HipContext>,
    job: *mut sys::lf_witness_job,
    _words: Box<[u64]>,
    _r: PhantomData<NTT>,
}
// SAFETY: the job handle is only ever passed to lf_witness_job_finish, once
unsafe impl<NTT> Send for HipPendingWitness<NTT> {}

impl<NTT> HipPendingWitness<NTT> {
    pub fn wait(mut self) -> Result<HipWitness<NTT>, HipError> {
        let mut raw = core::ptr::null_mut();
        let job = core::mem::replace(&mut self.job, core::ptr::null_mut());
        // SAFETY: job came from lf_witness_from_w_ccs_begin and is finished exactly once
        chk(unsafe { sys::lf_witness_job_finish(job, &mut raw) }, "lf_witness_job_finish")?;
        Ok(HipWitness { ctx: self.ctx.clone(), raw, _r: PhantomData })
    }
}

impl<NTT> Drop for HipPendingWitness<NTT> {
    fn drop(&mut self) {
        if !self.job.is_null() {
            // SAFETY: an abandoned job: wait for the worker (it reads `_words`), free what it made
            unsafe { sys::lf_witness_job_finish(self.job, core::ptr::null_mut()) };
        }
    }
}

impl<NTT> Drop for HipWitness<NTT> {
    fn drop(&mut self) {
        // SAFETY: raw came from an lf_witness_* constructor and is freed exactly once (a witness may outlive its context: the library allows it)
        unsafe { sys::lf_witness_free(self.raw) }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// transcript (transcript.rs:13-51, transcript/poseidon.rs:17-75)
// ------------------------------------------------------------------------------------------------------------------------------------
/// The library's host Poseidon sponge (AVX-512 IFMA lanes) behind the reference's transcript traits.
pub struct HipTranscript<NTT, CS> {
    raw: *mut sys::lf_transcript,
    _m: PhantomData<(NTT, CS)>,
}
// SAFETY: a transcript handle is plain host memory owned by this value
unsafe impl<NTT, CS> Send for HipTranscript<NTT, CS> {}

impl<NTT: SuitableRing, CS> Default for HipTranscript<NTT, CS> {
    fn default() -> Self {
        // SAFETY: returns a fresh handle (never null: aborts on allocation failure like `new`)
        Self { raw: unsafe { sys::lf_transcript_new_ring(ring_id::<NTT>().expect("ring")) }, _m: PhantomData }
    }
}

impl<NTT, CS> Clone for HipTranscript<NTT, CS> {
    fn clone(&self) -> Self {
        // SAFETY: raw is a live handle
        Self { raw: unsafe { sys::lf_transcript_clone(self.raw) }, _m: PhantomData }
    }
}

impl<NTT, CS> Drop for HipTranscript<NTT, CS> {
    fn drop(&mut self) {
        // SAFETY: freed exactly once
        unsafe { sys::lf_transcript_free(self.raw) }
    }
}

impl<NTT: SuitableRing, CS> Transcript<NTT> for HipTranscript<NTT, CS> {
    /// the Poseidon parameters are the ring's own (`GetPoseidonParams`), regenerated and checksummed inside the library: nothing to configure
    type TranscriptConfig = ();

    fn new(_config: &Self::TranscriptConfig) -> Self {
        Self::default()
    }

    fn absorb(&mut self, v: &NTT) {
        let mut w = Vec::with_capacity(NTT::WORDS);
        v.to_words(&mut w);
        // SAFETY: one ring element
        unsafe { sys::lf_transcript_absorb_ring(self.raw, w.as_ptr(), 1) }
    }

    fn absorb_slice(&mut self, v: &[NTT]) {
        let w = flatten(v);
        // SAFETY: v.len() ring elements
        unsafe { sys::lf_transcript_absorb_ring(self.raw, w.as_ptr(), v.len()) }
    }

    fn get_challenge(&mut self) -> NTT::BaseRing {
        let tau = <NTT::BaseRing as Field>::extension_degree() as usize;
        let mut w = vec![0u64; tau];
        // SAFETY: tau words out
        unsafe { sys::lf_transcript_get_challenge(self.raw, w.as_mut_ptr()) }
        let fs: Vec<_> = w.iter().map(|&x| <<NTT::BaseRing as Field>::BasePrimeField as From<u64>>::from(x)).collect();
        <NTT::BaseRing as Field>::from_base_prime_field_elems(&fs).expect("tau coordinates")
    }

    fn squeeze_bytes(&mut self, n: usize) -> Vec<u8> {
        let mut out = vec![0u8; n];
        // SAFETY: n bytes out
        unsafe { sys::lf_transcript_squeeze_bytes(self.raw, out.as_mut_ptr(), n) }
        out
    }
}

impl<NTT: SuitableRing, CS: LatticefoldChallengeSet<NTT>> TranscriptWithShortChallenges<NTT> for HipTranscript<NTT, CS> {
    type ChallengeSet = CS;

    fn get_short_challenge(&mut self) -> NTT::CoefficientRepresentation {
        let mut w = vec![0u64; <NTT::CoefficientRepresentation as RingWords>::WORDS];
        // SAFETY: one coefficient-form element out
        unsafe { sys::lf_transcript_get_short_challenge(self.raw, w.as_mut_ptr()) }
        <NTT::CoefficientRepresentation as RingWords>::from_words(&w)
    }
}

/// The trait methods of the provers are generic over `impl Transcript<NTT>`; the library needs ITS sponge.  Identity by type name (both are this crate's type).
fn as_hip<NTT: SuitableRing, X: Transcript<NTT>>(t: &mut X) -> Result<*mut sys::lf_transcript, HipError> {
    let name = core::any::type_name::<X>();
    if !name.starts_with("latticefold_hip::HipTranscript<") {
        return Err(HipError::ForeignTranscript);
    }
    // SAFETY: X is HipTranscript<NTT, _> (checked above); its first field is the handle and the layout is repr(Rust) of (ptr, ZST): read through a typed pointer
    let h = unsafe { &*(t as *mut X as *const HipTranscript<NTT, ()>) };
    Ok(h.raw)
}

// ------------------------------------------------------------------------------------------------------------------------------------
// flat instance / proof layouts (include/lfhip.h): LCCCS = r[s] v[tau] cm[kappa] u[t] x_w[l] h;  CCCS = cm[kappa] x_ccs[l]
// ------------------------------------------------------------------------------------------------------------------------------------
fn lcccs_words<NTT: SuitableRing>(x: &LCCCS<NTT>) -> Vec<u64> {
    let mut w = flatten(&x.r);
    w.extend(flatten(&x.v));
    w.extend(flatten(x.cm.as_ref()));
    w.extend(flatten(&x.u));
    w.extend(flatten(&x.x_w));
    x.h.to_words(&mut w);
    w
}

fn lcccs_from<NTT: SuitableRing>(w: &[u64], p: &sys::lf_params) -> LCCCS<NTT> {
    let tau = <NTT::BaseRing as Field>::extension_degree() as usize;
    let e: Vec<NTT> = unflatten(w);
    let (s, k, t, l) = (p.s as usize, p.kappa as usize, p.t as usize, p.l as usize);
    let mut i = 0;
    let mut take = |n: usize| { let v = e[i..i + n].to_vec(); i += n; v };
    let (r, v, cm, u, x_w) = (take(s), take(tau), take(k), take(t), take(l));
    LCCCS { r, v, cm: Commitment::from(cm), u, x_w, h: e[i] }
}

fn cccs_words<NTT: SuitableRing>(x: &CCCS<NTT>) -> Vec<u64> {
    let mut w = flatten(x.cm.as_ref());
    w.extend(flatten(&x.x_ccs));
    w
}

/// `Proof(Vec<ProverMsg { evaluations }>)` has crate-private fields (utils/sumcheck.rs:41-42, sumcheck/prover.rs:13-17) but derives `CanonicalDeserialize`:
/// a derived encoding is field by field, so `Vec<Vec<NTT>>` serialises to exactly the bytes of `Vec<ProverMsg<NTT>>` -- whatever the per-element layout of
/// the ring type is (it is arkworks' on both sides).
fn sumcheck_proof<NTT: SuitableRing>(e: &[NTT], rounds: usize, per: usize) -> SumcheckProof<NTT> {
    let msgs: Vec<Vec<NTT>> = (0..rounds).map(|i| e[i * per..(i + 1) * per].to_vec()).collect();
    let mut buf = Vec::new();
    msgs.serialize_uncompressed(&mut buf).expect("in-memory serialisation");
    SumcheckProof::deserialize_uncompressed(&buf[..]).expect("Vec<Vec<R>> and Vec<ProverMsg<R>> share one encoding")
}

fn lin_proof_from<NTT: SuitableRing>(e: &[NTT], p: &sys::lf_params) -> LinearizationProof<NTT> {
    let tau = <NTT::BaseRing as Field>::extension_degree() as usize;
    let (s, per, t) = (p.s as usize, p.d as usize + 2, p.t as usize);
    LinearizationProof { linearization_sumcheck: sumcheck_proof(e, s, per), v: e[s * per..s * per + tau].to_vec(), u: e[s * per + tau..s * per + tau + t].to_vec() }
}

fn dec_proof_from<NTT: SuitableRing>(e: &[NTT], p: &sys::lf_params) -> DecompositionProof<NTT> {
    let tau = <NTT::BaseRing as Field>::extension_degree() as usize;
    let (k, t, l, kap) = (p.K as usize, p.t as usize, p.l as usize + 1, p.kappa as usize);
    let rows = |off: usize, w: usize| -> Vec<Vec<NTT>> { (0..k).map(|i| e[off + i * w..off + (i + 1) * w].to_vec()).collect() };
    let (o_v, o_x, o_y) = (k * t, k * t + k * tau, k * t + k * tau + k * l);
    DecompositionProof { u_s: rows(0, t), v_s: rows(o_v, tau), x_s: rows(o_x, l), y_s: rows(o_y, kap).into_iter().map(Commitment::from_vec_raw).collect() }
}

fn fold_proof_from<NTT: SuitableRing>(e: &[NTT], p: &sys::lf_params) -> FoldingProof<NTT> {
    let tau = <NTT::BaseRing as Field>::extension_degree() as usize;
    let (s, per, k2, t) = (p.s as usize, 2 * p.b as usize + 1, 2 * p.K as usize, p.t as usize);
    let o = s * per;
    FoldingProof {
        pointshift_sumcheck_proof: sumcheck_proof(e, s, per),
        theta_s: (0..k2).map(|i| e[o + i * tau..o + (i + 1) * tau].to_vec()).collect(),
        eta_s: (0..k2).map(|i| e[o + k2 * tau + i * t..o + k2 * tau + (i + 1) * t].to_vec()).collect(),
    }
}

fn section_lens<NTT: SuitableRing>(p: &sys::lf_params) -> (usize, usize, usize) {
    let tau = <NTT::BaseRing as Field>::extension_degree() as usize;
    let lin = p.s as usize * (p.d as usize + 2) + tau + p.t as usize;
    let dec = p.K as usize * (p.t as usize + tau + p.l as usize + 1 + p.kappa as usize);
    let fold = p.s as usize * (2 * p.b as usize + 1) + 2 * p.K as usize * (tau + p.t as usize);
    (lin, dec, fold)
}

// ------------------------------------------------------------------------------------------------------------------------------------
// the three sub-provers and NIFSProver
// ------------------------------------------------------------------------------------------------------------------------------------
pub struct HipLinearizationProver<NTT, T> {
    _r: PhantomData<NTT>,
    _t: PhantomData<T>,
}

impl<NTT: SuitableRing, T: Transcript<NTT>> HipLinearizationProver<NTT, T> {
    /// resident form of `LinearizationProver::prove`
    pub fn prove_resident(ctx: &Arc<HipContext>, cm_i: &CCCS<NTT>, wit: &HipWitness<NTT>, transcript: &mut impl Transcript<NTT>) -> Result<(LCCCS<NTT>, LinearizationProof<NTT>), HipError> {
        let p = ctx.params()?;
        let tr = as_hip::<NTT, _>(transcript)?;
        let (lin, _, _) = section_lens::<NTT>(&p);
        // SAFETY: &p is a live lf_params
        let ll = unsafe { sys::lf_lcccs_len_ring(&p, ring_id::<NTT>()?) };
        let cccs = cccs_words(cm_i);
        let (mut lc, mut pr) = (vec![0u64; ll * NTT::WORDS], vec![0u64; lin * NTT::WORDS]);
        // SAFETY: buffer lengths follow include/lfhip.h (lf_lcccs_len_ring elements out, s (d + 2) + tau + t proof elements out)
        chk(unsafe { sys::lf_linearize(ctx.raw, tr, cccs.as_ptr(), wit.raw, lc.as_mut_ptr(), pr.as_mut_ptr()) }, "lf_linearize")?;
        Ok((lcccs_from(&lc, &p), lin_proof_from(&unflatten::<NTT>(&pr), &p)))
    }
}

impl<NTT: SuitableRing, T: Transcript<NTT>> LinearizationProver<NTT, T> for HipLinearizationProver<NTT, T> {
    fn prove(
        cm_i: &CCCS<NTT>,
        wit: &Witness<NTT>,
        transcript: &mut impl Transcript<NTT>,
        ccs: &CCS<NTT>,
    ) -> Result<(LCCCS<NTT>, LinearizationProof<NTT>), LinearizationError<NTT>> {
        let run = || -> Result<_, HipError> {
            let s = HipSession::current()?;
            s.check(ccs)?;
            let w = HipWitness::from_reference(&s.ctx, wit)?;
            Self::prove_resident(&s.ctx, cm_i, &w, transcript)
        };
        run().map_err(|e| LinearizationError::ParametersError(e.to_string()))
    }
}

pub struct HipDecompositionProver<NTT, T> {
    _r: PhantomData<NTT>,
    _t: PhantomData<T>,
}

impl<NTT: SuitableRing, T: Transcript<NTT>> HipDecompositionProver<NTT, T> {
    /// resident form: the K decomposed LCCCS and the proof; the K decomposed witnesses stay virtual on the device (base-b parts of `wit`)
    pub fn prove_resident(ctx: &Arc<HipContext>, cm_i: &LCCCS<NTT>, wit: &HipWitness<NTT>, transcript: &mut impl Transcript<NTT>) -> Result<(Vec<LCCCS<NTT>>, DecompositionProof<NTT>), HipError> {
        let p = ctx.params()?;
        let tr = as_hip::<NTT, _>(transcript)?;
        let (_, dec, _) = section_lens::<NTT>(&p);
        // SAFETY: as above
        let ll = unsafe { sys::lf_lcccs_len_ring(&p, ring_id::<NTT>()?) };
        let lc = lcccs_words(cm_i);
        let (mut outs, mut pr) = (vec![0u64; p.K as usize * ll * NTT::WORDS], vec![0u64; dec * NTT::WORDS]);
        // SAFETY: K LCCCS and K (t + tau + l + 1 + kappa) proof elements out
        chk(unsafe { sys::lf_decomposition_prove(ctx.raw, tr, lc.as_ptr(), wit.raw, outs.as_mut_ptr(), pr.as_mut_ptr()) }, "lf_decomposition_prove")?;
        let lcs = outs.chunks(ll * NTT::WORDS).map(|w| lcccs_from(w, &p)).collect();
        Ok((lcs, dec_proof_from(&unflatten::<NTT>(&pr), &p)))
    }
}

impl<NTT: SuitableRing, T: Transcript<NTT>> DecompositionProver<NTT, T> for HipDecompositionProver<NTT, T> {
    /// The trait hands back host objects the GPU path never materialises: `mz_mles` comes back EMPTY (only `LFFoldingProver` consumes it, and
    /// [`HipFoldingProver`] rebuilds M z on the device), and the K decomposed witnesses are rebuilt on the host from `wit.f_coeff` with the reference's own
    /// `decompose_to_vec(b, K).transpose()` + `Witness::from_f_coeff` -- O(K N) host work: use `prove_resident` in a loop.
    fn prove<P: DecompositionParams>(
        cm_i: &LCCCS<NTT>,
        wit: &Witness<NTT>,
        transcript: &mut impl Transcript<NTT>,
        ccs: &CCS<NTT>,
        _scheme: &AjtaiCommitmentScheme<NTT>,
    ) -> Result<(Vec<Vec<DenseMultilinearExtension<NTT>>>, Vec<LCCCS<NTT>>, Vec<Witness<NTT>>, DecompositionProof<NTT>), DecompositionError> {
        let run = || -> Result<_, HipError> {
            let s = HipSession::current()?;
            s.check(ccs)?;
            let w = HipWitness::from_reference(&s.ctx, wit)?;
            Self::prove_resident(&s.ctx, cm_i, &w, transcript)
        };
        let (lcs, proof) = run().map_err(|e| { eprintln!("latticefold-hip: {e}"); DecompositionError::IncorrectLength })?;
        // decompose_B_vec_into_k_vec (nifs/decomposition/utils.rs:45-49; pub(super) there)
        let parts: Vec<Vec<NTT::CoefficientRepresentation>> = wit.f_coeff.decompose_to_vec(P::B_SMALL as u128, P::K).transpose();
        let wits = parts.into_iter().map(|f| Witness::from_f_coeff::<P>(f)).collect();
        Ok((Vec::new(), lcs, wits, proof))
    }
}

pub struct HipFoldingProver<NTT, T> {
    _r: PhantomData<NTT>,
    _t: PhantomData<T>,
}

impl<NTT: SuitableRing, T: TranscriptWithShortChallenges<NTT>> HipFoldingProver<NTT, T> {
    /// resident form: `cm_i_s` = the 2K decomposed LCCCS, `w_left` / `w_right` the witnesses they are the base-b parts of
    pub fn prove_resident(ctx: &Arc<HipContext>, cm_i_s: &[LCCCS<NTT>], w_left: &HipWitness<NTT>, w_right: &HipWitness<NTT>, transcript: &mut impl TranscriptWithShortChallenges<NTT>) -> Result<(LCCCS<NTT>, HipWitness<NTT>, FoldingProof<NTT>), HipError> {
        let p = ctx.params()?;
        let tr = as_hip::<NTT, _>(transcript)?;
        let (_, _, fold) = section_lens::<NTT>(&p);
        // SAFETY: as above
        let ll = unsafe { sys::lf_lcccs_len_ring(&p, ring_id::<NTT>()?) };
        let mut ins = Vec::with_capacity(cm_i_s.len() * ll * NTT::WORDS);
        for x in cm_i_s {
            ins.extend(lcccs_words(x));
        }
        let (mut lc, mut pr, mut w_out) = (vec![0u64; ll * NTT::WORDS], vec![0u64; fold * NTT::WORDS], core::ptr::null_mut());
        // SAFETY: 2K LCCCS in, one LCCCS and s (2b + 1) + 2K (tau + t) proof elements out; w_out receives a fresh handle
        chk(unsafe { sys::lf_folding_prove(ctx.raw, tr, ins.as_ptr(), w_left.raw, w_right.raw, lc.as_mut_ptr(), &mut w_out, pr.as_mut_ptr()) }, "lf_folding_prove")?;
        Ok((lcccs_from(&lc, &p), HipWitness { ctx: ctx.clone(), raw: w_out, _r: PhantomData }, fold_proof_from(&unflatten::<NTT>(&pr), &p)))
    }
}

impl<NTT: SuitableRing, T: TranscriptWithShortChallenges<NTT>> FoldingProver<NTT, T> for HipFoldingProver<NTT, T> {
    /// `w_s` = 2K decomposed witnesses (K of the accumulator, K of the fresh instance): recomposed on the host (sum_k b^k f_k) and uploaded as the two
    /// witnesses whose virtual base-b parts they are; `mz_mles` is ignored (M z is rebuilt on the device).
    fn prove<P: DecompositionParams>(
        cm_i_s: &[LCCCS<NTT>],
        w_s: Vec<Witness<NTT>>,
        transcript: &mut impl TranscriptWithShortChallenges<NTT>,
        ccs: &CCS<NTT>,
        _mz_mles: &[Vec<DenseMultilinearExtension<NTT>>],
    ) -> Result<(LCCCS<NTT>, Witness<NTT>, FoldingProof<NTT>), FoldingError<NTT>> {
        let run = || -> Result<_, HipError> {
            let s = HipSession::current()?;
            s.check(ccs)?;
            let k = P::K;
            let recompose = |ws: &[Witness<NTT>]| -> Vec<NTT::CoefficientRepresentation> {
                (0..ws[0].f_coeff.len())
                    .map(|i| recompose(&ws.iter().map(|w| w.f_coeff[i]).collect::<Vec<_>>(), P::B_SMALL as u128))     // sum_k b^k f_k[i]
                    .collect()
            };
            let wl = HipWitness::from_f_coeff(&s.ctx, &recompose(&w_s[..k]))?;
            let wr = HipWitness::from_f_coeff(&s.ctx, &recompose(&w_s[k..]))?;
            let (lc, w0, proof) = Self::prove_resident(&s.ctx, cm_i_s, &wl, &wr, transcript)?;
            Ok((lc, w0.to_reference::<P>()?, proof))
        };
        run().map_err(|e| { eprintln!("latticefold-hip: {e}"); FoldingError::IncorrectLength })
    }
}

/// `NIFSProver<NTT, P, T>` (nifs.rs:36-103)
pub struct HipNIFSProver<NTT, P, T> {
    _r: PhantomData<NTT>,
    _p: PhantomData<P>,
    _t: PhantomData<T>,
}

impl<NTT: SuitableRing, P: DecompositionParams, T: TranscriptWithShortChallenges<NTT>> HipNIFSProver<NTT, P, T> {
    /// `NIFSProver::prove` with the reference's signature: one fold step (lf_fold_step).  The host `Witness`es are uploaded and the folded one downloaded per
    /// call; a chain should use [`Self::prove_resident`].
    pub fn prove(
        acc: &LCCCS<NTT>,
        w_acc: &Witness<NTT>,
        cm_i: &CCCS<NTT>,
        w_i: &Witness<NTT>,
        transcript: &mut impl TranscriptWithShortChallenges<NTT>,
        ccs: &CCS<NTT>,
        _scheme: &AjtaiCommitmentScheme<NTT>,
    ) -> Result<(LCCCS<NTT>, Witness<NTT>, LFProof<NTT>), LatticefoldError<NTT>> {
        let run = || -> Result<_, HipError> {
            let s = HipSession::current()?;
            s.check(ccs)?;
            let (wa, wi) = (HipWitness::from_reference(&s.ctx, w_acc)?, HipWitness::from_reference(&s.ctx, w_i)?);
            let (lc, w0, proof) = Self::prove_resident(&s.ctx, acc, &wa, cm_i, &wi, transcript)?;
            Ok((lc, w0.to_reference::<P>()?, proof))
        };
        run().map_err(|e| LatticefoldError::LinearizationError(LinearizationError::ParametersError(e.to_string())))
    }

    /// The same step on resident witnesses: acc / cm_i / proof cross PCIe, nothing else.
    pub fn prove_resident(
        ctx: &Arc<HipContext>,
        acc: &LCCCS<NTT>,
        w_acc: &HipWitness<NTT>,
        cm_i: &CCCS<NTT>,
        w_i: &HipWitness<NTT>,
        transcript: &mut impl TranscriptWithShortChallenges<NTT>,
    ) -> Result<(LCCCS<NTT>, HipWitness<NTT>, LFProof<NTT>), HipError> {
        let p = ctx.params()?;
        let tr = as_hip::<NTT, _>(transcript)?;
        let (lin, dec, fold) = section_lens::<NTT>(&p);
        // SAFETY: as above
        let ll = unsafe { sys::lf_lcccs_len_ring(&p, ring_id::<NTT>()?) };
        let (a, c) = (lcccs_words(acc), cccs_words(cm_i));
        let (mut lc, mut pr, mut w_out) = (vec![0u64; ll * NTT::WORDS], vec![0u64; (lin + 2 * dec + fold) * NTT::WORDS], core::ptr::null_mut());
        // SAFETY: buffer lengths per include/lfhip.h (lf_lcccs_len_ring / lf_proof_len_ring); w_out receives a fresh handle
        chk(unsafe { sys::lf_fold_step(ctx.raw, tr, a.as_ptr(), w_acc.raw, c.as_ptr(), w_i.raw, lc.as_mut_ptr(), &mut w_out, pr.as_mut_ptr()) }, "lf_fold_step")?;
        let e: Vec<NTT> = unflatten(&pr);
        let proof = LFProof {
            linearization_proof: lin_proof_from(&e[..lin], &p),
            decomposition_proof_l: dec_proof_from(&e[lin..lin + dec], &p),
            decomposition_proof_r: dec_proof_from(&e[lin + dec..lin + 2 * dec], &p),
            folding_proof: fold_proof_from(&e[lin + 2 * dec..], &p),
        };
        Ok((lcccs_from(&lc, &p), HipWitness { ctx: ctx.clone(), raw: w_out, _r: PhantomData }, proof))
    }
}
