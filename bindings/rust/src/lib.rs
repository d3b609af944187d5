//! Safe wrappers over `sys.rs` for the entry points the bridge's verifier calls (INTEGRATION.md has the call-site table).
//! Not built in the repository this file lives in (no cargo there); kept next to the header it is generated from.
mod sys;
pub use sys::*;
use std::ffi::CStr;

pub struct Ctx(*mut mina_ctx);
unsafe impl Send for Ctx {}          // one context per worker thread; not Sync

#[derive(Debug)]
pub struct MinaError(pub i32, pub String);

fn check(rc: i32) -> Result<(), MinaError> {
    if rc == 0 { Ok(()) } else {
        let msg = unsafe { CStr::from_ptr(mina_last_error()) }.to_string_lossy().into_owned();
        Err(MinaError(rc, msg))
    }
}

impl Ctx {
    pub fn new(device: i32) -> Result<Self, MinaError> {
        let mut p = std::ptr::null_mut();
        check(unsafe { mina_ctx_create(device, &mut p) })?;
        Ok(Ctx(p))
    }
    /// `SRS::<G>::create(depth)` on the GPU (regenerates srs/vesta.srs / srs/pallas.srs point for point).
    pub fn srs_create(&mut self, curve: i32, depth: u32) -> Result<(), MinaError> {
        check(unsafe { mina_srs_create(self.0, curve, depth) })
    }
    /// openmina `accumulator_check` for `sg.len() / 64` proofs, one deterministic verdict each.
    pub fn accumulator_check_multi(&mut self, curve: i32, k: u32, prechallenges: &[u8], sg: &[u8]) -> Result<Vec<bool>, MinaError> {
        let n = sg.len() / 64;
        assert_eq!(prechallenges.len(), n * k as usize * 16);
        let mut v = vec![0u8; n];
        check(unsafe { mina_accumulator_check_multi(self.0, curve, k, n, prechallenges.as_ptr(), sg.as_ptr(), v.as_mut_ptr()) })?;
        Ok(v.into_iter().map(|b| b != 0).collect())
    }
    /// `SRS::verify` on a batch; the two RNG draws of upstream are explicit arguments.  Any failure is `false`.
    pub fn ipa_batch_check(&mut self, curve: i32, openings: &[mina_ipa_opening], rand_base: &[u8; 32], sg_rand_base: &[u8; 32]) -> bool {
        let mut v = 0u8;
        let rc = unsafe { mina_ipa_batch_check(self.0, curve, openings.len(), openings.as_ptr(), rand_base.as_ptr(), sg_rand_base.as_ptr(), &mut v) };
        rc == 0 && v == 1
    }
}

/// Stands where Aligned's `verify_mina_state_ffi` stands: the bytes of `bincode::serialize(&MinaStateProof)` / `(&MinaStatePubInputs)`
/// (core/src/aligned.rs:33-36).  Process-wide context, every failure is `false`.
/// NEVER COMPILED: this crate is generated from include/mina_verify.h and diffed against it (tests/test_abi.py); the image it was written in has no cargo.
/// Before calling it a drop-in, build it and link it under the caller of core/src/aligned.rs:31-58.
pub fn verify_mina_state(proof: &[u8], pub_input: &[u8]) -> bool {
    unsafe { mina_verify_state(proof.as_ptr(), proof.len(), pub_input.as_ptr(), pub_input.len()) }
}
/// Stands where `verify_account_inclusion_ffi` stands (core/src/aligned.rs:46-49); never compiled either (see above).
pub fn verify_account_inclusion(proof: &[u8], pub_input: &[u8]) -> bool {
    unsafe { mina_verify_account(proof.as_ptr(), proof.len(), pub_input.as_ptr(), pub_input.len()) }
}
/// Batch form: one GPU pipeline for all proofs, one verdict each.
pub fn verify_mina_state_batch(proofs: &[&[u8]], pub_inputs: &[&[u8]]) -> Vec<bool> {
    let n = proofs.len();
    let (pp, pl): (Vec<_>, Vec<_>) = proofs.iter().map(|p| (p.as_ptr(), p.len())).unzip();
    let (qp, ql): (Vec<_>, Vec<_>) = pub_inputs.iter().map(|p| (p.as_ptr(), p.len())).unzip();
    let mut v = vec![0u8; n];
    let rc = unsafe { mina_verify_state_batch(n, pp.as_ptr(), pl.as_ptr(), qp.as_ptr(), ql.as_ptr(), v.as_mut_ptr()) };
    v.into_iter().map(|b| rc == 0 && b == 1).collect()
}

impl Drop for Ctx {
    fn drop(&mut self) { unsafe { mina_ctx_destroy(self.0) } }
}
