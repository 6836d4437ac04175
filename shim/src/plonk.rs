//! The Plonky2 surface the reference's circuit code is written against (`CircuitBuilder`, `CircuitData::prove` / `verify`,
//! `PartialWitness`, the target types), over the library's circuit API (`p2mt_cb_*`, `p2mt_circuit_*`, `p2mt_pw_*` in
//! include/p2mt.h).  Names and argument order follow plonky2 @3b21b87d as the reference calls it
//! (/root/reference/src/mmr/common.rs:5-58, mmr_plonky2_verifier.rs:13-91,122-150, mmr_plonky2_verifier_1_recursion.rs:20-140,
//! 180-220), so that the three circuit modules below read like the reference's own.  These are NOT plonky2's types: the prover
//! behind `prove` is the MI355X one, configured as `CircuitConfig::standard_recursion_config()` and nothing else.
//! Source only: the image this was written in has no Rust toolchain.
use crate::{ffi, ok};
use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::field::types::{Field, PrimeField64};
use plonky2::hash::hash_types::HashOut;
use std::rc::Rc;

/// plonky2::iop::target::Target (an opaque 64-bit handle of the library).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct Target(pub u64);
/// plonky2::iop::target::BoolTarget
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct BoolTarget {
    pub target: Target,
}
/// plonky2::hash::hash_types::HashOutTarget
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct HashOutTarget {
    pub elements: [Target; 4],
}
impl HashOutTarget {
    pub fn from_vec(v: Vec<Target>) -> Self {
        assert!(v.len() == 4);
        HashOutTarget { elements: [v[0], v[1], v[2], v[3]] }
    }
}
/// plonky2::plonk::proof::ProofWithPublicInputsTarget<2>: one target per word of the proof (library order), the public inputs last.
#[derive(Clone, Debug)]
pub struct ProofWithPublicInputsTarget {
    pub(crate) words: Vec<u64>,
    pub public_inputs: Vec<Target>,
}
/// plonky2::plonk::circuit_data::VerifierCircuitTarget: constants_sigmas_cap [16][4] | circuit_digest [4]
#[derive(Clone, Debug)]
pub struct VerifierCircuitTarget {
    pub(crate) words: [u64; 68],
}
/// plonky2::plonk::proof::ProofWithPublicInputs<F, C, 2>: the proof words `CircuitData::prove` returns.
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct ProofWithPublicInputs {
    pub words: Vec<u64>,
    pub public_inputs: Vec<GoldilocksField>,
}

struct Handle(*mut ffi::p2mt_circuit_data);
impl Drop for Handle {
    fn drop(&mut self) {
        unsafe { ffi::p2mt_circuit_destroy(self.0) };
    }
}

/// `circuit_data.common` / `.verifier_only` of an inner circuit: what `add_virtual_proof_with_pis`, `verify_proof` and
/// `set_verifier_data_target` need (the built circuit itself, shared).
#[derive(Clone)]
pub struct CommonCircuitData {
    inner: Rc<Handle>,
}
pub type VerifierOnlyCircuitData = CommonCircuitData;
pub struct ProverOnlyCircuitData {
    pub public_inputs: Vec<Target>,
}

/// plonky2::plonk::circuit_data::CircuitData<GoldilocksField, PoseidonGoldilocksConfig, 2>
pub struct CircuitData {
    handle: Rc<Handle>,
    pub common: CommonCircuitData,
    pub verifier_only: VerifierOnlyCircuitData,
    pub prover_only: ProverOnlyCircuitData,
    proof_len: usize,
    num_public_inputs: usize,
}
impl CircuitData {
    /// circuit_data.prove(pw) (mmr_plonky2_verifier.rs:148)
    pub fn prove(&self, pw: PartialWitness) -> anyhow::Result<ProofWithPublicInputs> {
        let mut words = vec![0u64; self.proof_len];
        let rc = unsafe { ffi::p2mt_circuit_prove(self.handle.0, pw.0, words.as_mut_ptr(), words.len()) };
        if rc != ffi::P2MT_OK {
            let msg = unsafe { std::ffi::CStr::from_ptr(ffi::p2mt_last_error()) }.to_string_lossy().into_owned();
            anyhow::bail!("prove: p2mt status {rc}: {msg}");
        }
        let public_inputs = words[self.proof_len - self.num_public_inputs..].iter().map(|w| GoldilocksField::from_canonical_u64(*w)).collect();
        Ok(ProofWithPublicInputs { words, public_inputs })
    }
    /// circuit_data.verify(proof) (mmr_plonky2_verifier.rs:150)
    pub fn verify(&self, proof: ProofWithPublicInputs) -> anyhow::Result<()> {
        let (mut accepted, mut reason) = (0i32, 0i32);
        ok(unsafe { ffi::p2mt_circuit_verify(self.handle.0, proof.words.as_ptr(), proof.words.len(), &mut accepted, &mut reason) });
        if accepted == 1 { Ok(()) } else { anyhow::bail!("proof rejected (reason {reason})") }
    }
}

/// plonky2::iop::witness::PartialWitness<GoldilocksField> + the WitnessWrite methods the reference uses
pub struct PartialWitness(*mut ffi::p2mt_partial_witness);
impl PartialWitness {
    pub fn new() -> Self {
        let mut p = std::ptr::null_mut();
        ok(unsafe { ffi::p2mt_pw_create(&mut p) });
        PartialWitness(p)
    }
    pub fn set_target(&mut self, t: Target, v: GoldilocksField) {
        ok(unsafe { ffi::p2mt_pw_set_target(self.0, t.0, v.to_canonical_u64()) })
    }
    pub fn set_bool_target(&mut self, t: BoolTarget, v: bool) {
        ok(unsafe { ffi::p2mt_pw_set_target(self.0, t.target.0, v as u64) })
    }
    pub fn set_hash_target(&mut self, t: HashOutTarget, v: HashOut<GoldilocksField>) {
        for i in 0..4 {
            self.set_target(t.elements[i], v.elements[i]);
        }
    }
    /// pw.set_proof_with_pis_target(&target, &proof) (mmr_plonky2_verifier_1_recursion.rs:201)
    pub fn set_proof_with_pis_target(&mut self, t: &ProofWithPublicInputsTarget, proof: &ProofWithPublicInputs) {
        assert!(t.words.len() == proof.words.len());
        ok(unsafe { ffi::p2mt_pw_set_proof_with_pis_target(self.0, t.words.as_ptr(), proof.words.as_ptr(), proof.words.len()) })
    }
    /// pw.set_verifier_data_target(&target, &inner.verifier_only) (:202)
    pub fn set_verifier_data_target(&mut self, t: &VerifierCircuitTarget, inner: &VerifierOnlyCircuitData) {
        ok(unsafe { ffi::p2mt_pw_set_verifier_data_target(self.0, t.words.as_ptr(), inner.inner.0) })
    }
}
impl Drop for PartialWitness {
    fn drop(&mut self) {
        unsafe { ffi::p2mt_pw_destroy(self.0) };
    }
}

/// plonky2::plonk::circuit_builder::CircuitBuilder<GoldilocksField, 2>::new(CircuitConfig::standard_recursion_config())
pub struct CircuitBuilder(*mut ffi::p2mt_circuit_builder);
macro_rules! cb1 {
    ($self:ident, $f:ident $(, $a:expr)*) => {{
        let mut out = 0u64;
        ok(unsafe { ffi::$f($self.0 $(, $a)*, &mut out) });
        Target(out)
    }};
}
impl CircuitBuilder {
    pub fn new() -> Self {
        let mut p = std::ptr::null_mut();
        ok(unsafe { ffi::p2mt_cb_create(&mut p) });
        CircuitBuilder(p)
    }
    pub fn add_virtual_target(&mut self) -> Target { cb1!(self, p2mt_cb_add_virtual_target) }
    pub fn add_virtual_bool_target_safe(&mut self) -> BoolTarget { BoolTarget { target: cb1!(self, p2mt_cb_add_virtual_bool_target_safe) } }
    pub fn add_virtual_hash(&mut self) -> HashOutTarget {
        HashOutTarget { elements: [self.add_virtual_target(), self.add_virtual_target(), self.add_virtual_target(), self.add_virtual_target()] }
    }
    pub fn constant(&mut self, c: GoldilocksField) -> Target { cb1!(self, p2mt_cb_constant, c.to_canonical_u64()) }
    pub fn zero(&mut self) -> Target { self.constant(GoldilocksField::ZERO) }
    pub fn one(&mut self) -> Target { self.constant(GoldilocksField::ONE) }
    pub fn connect(&mut self, x: Target, y: Target) { ok(unsafe { ffi::p2mt_cb_connect(self.0, x.0, y.0) }) }
    pub fn mul(&mut self, x: Target, y: Target) -> Target { cb1!(self, p2mt_cb_mul, x.0, y.0) }
    pub fn mul_add(&mut self, x: Target, y: Target, z: Target) -> Target { cb1!(self, p2mt_cb_mul_add, x.0, y.0, z.0) }
    pub fn not(&mut self, b: BoolTarget) -> BoolTarget { BoolTarget { target: cb1!(self, p2mt_cb_not, b.target.0) } }
    pub fn or(&mut self, b1: BoolTarget, b2: BoolTarget) -> BoolTarget { BoolTarget { target: cb1!(self, p2mt_cb_or, b1.target.0, b2.target.0) } }
    pub fn is_equal(&mut self, x: Target, y: Target) -> BoolTarget { BoolTarget { target: cb1!(self, p2mt_cb_is_equal, x.0, y.0) } }
    fn hash4(&mut self, inputs: Vec<Target>, no_pad: bool) -> HashOutTarget {
        let words: Vec<u64> = inputs.iter().map(|t| t.0).collect();
        let mut out = [0u64; 4];
        ok(unsafe {
            if no_pad { ffi::p2mt_cb_hash_n_to_hash_no_pad(self.0, words.as_ptr(), words.len(), out.as_mut_ptr()) }
            else { ffi::p2mt_cb_hash_or_noop(self.0, words.as_ptr(), words.len(), out.as_mut_ptr()) }
        });
        HashOutTarget { elements: out.map(Target) }
    }
    /// builder.hash_or_noop::<PoseidonHash>(inputs)
    pub fn hash_or_noop(&mut self, inputs: Vec<Target>) -> HashOutTarget { self.hash4(inputs, false) }
    /// builder.hash_n_to_hash_no_pad::<PoseidonHash>(inputs)
    pub fn hash_n_to_hash_no_pad(&mut self, inputs: Vec<Target>) -> HashOutTarget { self.hash4(inputs, true) }
    pub fn register_public_input(&mut self, t: Target) { self.register_public_inputs(&[t]) }
    pub fn register_public_inputs(&mut self, ts: &[Target]) {
        let words: Vec<u64> = ts.iter().map(|t| t.0).collect();
        ok(unsafe { ffi::p2mt_cb_register_public_inputs(self.0, words.as_ptr(), words.len()) })
    }
    /// builder.add_virtual_proof_with_pis(&inner.common) (mmr_plonky2_verifier_1_recursion.rs:95)
    pub fn add_virtual_proof_with_pis(&mut self, inner: &CommonCircuitData) -> ProofWithPublicInputsTarget {
        let mut info = unsafe { std::mem::zeroed::<ffi::p2mt_circuit_info>() };
        ok(unsafe { ffi::p2mt_circuit_get_info(inner.inner.0, &mut info) });
        let mut words = vec![0u64; info.proof_len as usize];
        ok(unsafe { ffi::p2mt_cb_add_virtual_proof_with_pis(self.0, inner.inner.0, words.as_mut_ptr(), words.len()) });
        let npi = info.num_public_inputs as usize;
        let public_inputs = words[words.len() - npi..].iter().map(|w| Target(*w)).collect();
        ProofWithPublicInputsTarget { words, public_inputs }
    }
    /// builder.add_virtual_verifier_data(cap_height) (:98)
    pub fn add_virtual_verifier_data(&mut self, cap_height: usize) -> VerifierCircuitTarget {
        let mut words = [0u64; 68];
        ok(unsafe { ffi::p2mt_cb_add_virtual_verifier_data(self.0, cap_height as u32, words.as_mut_ptr()) });
        VerifierCircuitTarget { words }
    }
    /// builder.verify_proof::<PoseidonGoldilocksConfig>(&proof, &verifier_data, &inner.common) (:101-104)
    pub fn verify_proof(&mut self, proof: &ProofWithPublicInputsTarget, vd: &VerifierCircuitTarget, inner: &CommonCircuitData) {
        ok(unsafe { ffi::p2mt_cb_verify_proof(self.0, proof.words.as_ptr(), proof.words.len(), vd.words.as_ptr(), inner.inner.0) })
    }
    /// builder.build::<PoseidonGoldilocksConfig>() (consumes the builder, as plonky2's does)
    pub fn build(self) -> CircuitData {
        let mut c = std::ptr::null_mut();
        ok(unsafe { ffi::p2mt_cb_build(self.0, &mut c) });
        let handle = Rc::new(Handle(c));
        let mut info = unsafe { std::mem::zeroed::<ffi::p2mt_circuit_info>() };
        ok(unsafe { ffi::p2mt_circuit_get_info(c, &mut info) });
        let npi = info.num_public_inputs as usize;
        let mut pis = vec![0u64; npi];
        ok(unsafe { ffi::p2mt_circuit_public_inputs(c, pis.as_mut_ptr()) });
        let common = CommonCircuitData { inner: handle.clone() };
        CircuitData {
            handle,
            verifier_only: common.clone(),
            common,
            prover_only: ProverOnlyCircuitData { public_inputs: pis.into_iter().map(Target).collect() },
            proof_len: info.proof_len as usize,
            num_public_inputs: npi,
        }
    }
}
impl Drop for CircuitBuilder {
    fn drop(&mut self) {
        unsafe { ffi::p2mt_cb_destroy(self.0) };
    }
}
