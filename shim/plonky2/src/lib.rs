//! plonky2's module tree as the reference imports it (/root/reference/src/mmr/mmr_plonky2_verifier.rs:1-4, :18-20, :30-34, :89;
//! mmr_plonky2_verifier_1_recursion.rs:1-4, :84-140; common.rs:1; merkle_mountain_ranges.rs:3; simple_merkle_tree.rs:8), every item
//! forwarding to libp2mt_hip.so (include/p2mt.h).  F is always GoldilocksField, D always 2, the configuration always
//! `standard_recursion_config()`: the generic parameters exist so that the reference's type annotations resolve, and anything else is
//! refused at compile time by the trait bounds below.
pub use plonky2_field as field;

use p2mt_sys as ffi;
use p2mt_sys::ok;

/// GoldilocksField <-> the canonical u64 the ABI speaks
pub trait Gl64: Copy {
    fn to_u64(&self) -> u64;
    fn from_u64(x: u64) -> Self;
}
impl Gl64 for field::goldilocks_field::GoldilocksField {
    fn to_u64(&self) -> u64 {
        use field::types::PrimeField64;
        self.to_canonical_u64()
    }
    fn from_u64(x: u64) -> Self {
        use field::types::Field;
        Self::from_canonical_u64(x)
    }
}

pub mod iop {
    pub mod target {
        /// plonky2::iop::target::Target (an opaque 64-bit handle of the library's builder)
        #[derive(Clone, Copy, Debug, PartialEq, Eq, Hash)]
        pub struct Target(pub u64);
        /// plonky2::iop::target::BoolTarget
        #[derive(Clone, Copy, Debug, PartialEq, Eq)]
        pub struct BoolTarget {
            pub target: Target,
            pub(crate) _private: (),
        }
        impl BoolTarget {
            pub fn new_unsafe(target: Target) -> Self {
                BoolTarget { target, _private: () }
            }
        }
    }
    pub mod witness {
        use super::target::{BoolTarget, Target};
        use crate::hash::hash_types::{HashOut, HashOutTarget};
        use crate::plonk::circuit_data::{VerifierCircuitTarget, VerifierOnlyCircuitData};
        use crate::plonk::config::GenericConfig;
        use crate::plonk::proof::{ProofWithPublicInputs, ProofWithPublicInputsTarget};
        use crate::{ffi, ok, Gl64};
        use std::marker::PhantomData;

        /// plonky2::iop::witness::PartialWitness<F>
        pub struct PartialWitness<F: Gl64> {
            pub(crate) raw: *mut ffi::p2mt_partial_witness,
            _f: PhantomData<F>,
        }
        impl<F: Gl64> PartialWitness<F> {
            pub fn new() -> Self {
                let mut p = std::ptr::null_mut();
                ok(unsafe { ffi::p2mt_pw_create(&mut p) });
                PartialWitness { raw: p, _f: PhantomData }
            }
        }
        impl<F: Gl64> Drop for PartialWitness<F> {
            fn drop(&mut self) {
                unsafe { ffi::p2mt_pw_destroy(self.raw) };
            }
        }
        /// plonky2::iop::witness::WitnessWrite<F>: the setters the reference's tests call (mmr_plonky2_verifier.rs:126-146,
        /// mmr_plonky2_verifier_1_recursion.rs:183-213)
        pub trait WitnessWrite<F: Gl64> {
            fn set_target(&mut self, target: Target, value: F);
            fn set_bool_target(&mut self, target: BoolTarget, value: bool) {
                self.set_target(target.target, F::from_u64(value as u64))
            }
            fn set_hash_target(&mut self, ht: HashOutTarget, value: HashOut<F>) {
                for i in 0..4 {
                    self.set_target(ht.elements[i], value.elements[i]);
                }
            }
            fn set_proof_with_pis_target<C: GenericConfig<D, F = F>, const D: usize>(
                &mut self,
                proof_with_pis_target: &ProofWithPublicInputsTarget<D>,
                proof_with_pis: &ProofWithPublicInputs<F, C, D>,
            );
            fn set_verifier_data_target<C: GenericConfig<D, F = F>, const D: usize>(
                &mut self,
                vdt: &VerifierCircuitTarget,
                vd: &VerifierOnlyCircuitData<C, D>,
            );
        }
        impl<F: Gl64> WitnessWrite<F> for PartialWitness<F> {
            fn set_target(&mut self, target: Target, value: F) {
                ok(unsafe { ffi::p2mt_pw_set_target(self.raw, target.0, value.to_u64()) })
            }
            fn set_proof_with_pis_target<C: GenericConfig<D, F = F>, const D: usize>(
                &mut self,
                t: &ProofWithPublicInputsTarget<D>,
                proof: &ProofWithPublicInputs<F, C, D>,
            ) {
                assert!(t.words.len() == proof.words.len());
                ok(unsafe { ffi::p2mt_pw_set_proof_with_pis_target(self.raw, t.words.as_ptr(), proof.words.as_ptr(), proof.words.len()) })
            }
            fn set_verifier_data_target<C: GenericConfig<D, F = F>, const D: usize>(
                &mut self,
                vdt: &VerifierCircuitTarget,
                vd: &VerifierOnlyCircuitData<C, D>,
            ) {
                ok(unsafe { ffi::p2mt_pw_set_verifier_data_target(self.raw, vdt.words.as_ptr(), vd.handle.0) })
            }
        }
    }
}

pub mod hash {
    pub mod hash_types {
        use crate::iop::target::Target;
        /// plonky2::hash::hash_types::HashOut<F>
        #[derive(Clone, Copy, Debug, PartialEq, Eq)]
        pub struct HashOut<F> {
            pub elements: [F; 4],
        }
        /// plonky2::hash::hash_types::HashOutTarget
        #[derive(Clone, Copy, Debug, PartialEq, Eq)]
        pub struct HashOutTarget {
            pub elements: [Target; 4],
        }
        impl HashOutTarget {
            pub fn from_vec(elements: Vec<Target>) -> Self {
                assert!(elements.len() == 4);
                HashOutTarget { elements: [elements[0], elements[1], elements[2], elements[3]] }
            }
        }
    }
    pub mod poseidon {
        use super::hash_types::HashOut;
        use crate::plonk::config::{AlgebraicHasher, Hasher};
        use crate::{ffi, ok, Gl64};
        /// plonky2::hash::poseidon::PoseidonHash
        #[derive(Clone, Copy, Debug, PartialEq, Eq)]
        pub struct PoseidonHash;
        /// The per-hash trait the reference's tree code calls (simple_merkle_tree.rs:23,33; merkle_mountain_ranges.rs:91,111,125,233):
        /// one hash per call is useless as a GPU boundary, so these run the library's HOST permutation
        /// (p2mt_host_poseidon_permute, csrc/host_poseidon.hip: plonky2's own CPU speed); the bulk path is the
        /// plonky2_merkle_trees_gpu crate (MMR::extend / from_leaves).
        impl<F: Gl64> Hasher<F> for PoseidonHash {
            type Hash = HashOut<F>;
            fn hash_no_pad(input: &[F]) -> HashOut<F> {
                let mut s = [0u64; 12];
                for chunk in input.chunks(8) {
                    for (k, x) in chunk.iter().enumerate() {
                        s[k] = x.to_u64();
                    }
                    ok(unsafe { ffi::p2mt_host_poseidon_permute(s.as_ptr(), s.as_mut_ptr(), 1) });
                }
                HashOut { elements: [F::from_u64(s[0]), F::from_u64(s[1]), F::from_u64(s[2]), F::from_u64(s[3])] }
            }
            fn hash_or_noop(inputs: &[F]) -> HashOut<F> {
                if inputs.len() <= 4 {
                    let mut e = [F::from_u64(0); 4];
                    e[..inputs.len()].copy_from_slice(inputs);
                    HashOut { elements: e }
                } else {
                    Self::hash_no_pad(inputs)
                }
            }
            fn two_to_one(left: HashOut<F>, right: HashOut<F>) -> HashOut<F> {
                let mut s = [0u64; 12];
                for k in 0..4 {
                    s[k] = left.elements[k].to_u64();
                    s[4 + k] = right.elements[k].to_u64();
                }
                ok(unsafe { ffi::p2mt_host_poseidon_permute(s.as_ptr(), s.as_mut_ptr(), 1) });
                HashOut { elements: [F::from_u64(s[0]), F::from_u64(s[1]), F::from_u64(s[2]), F::from_u64(s[3])] }
            }
        }
        impl<F: Gl64> AlgebraicHasher<F> for PoseidonHash {}
    }
}

pub mod plonk {
    pub mod config {
        use crate::hash::poseidon::PoseidonHash;
        use crate::Gl64;
        /// plonky2::plonk::config::Hasher<F>
        pub trait Hasher<F: Gl64>: Sized {
            type Hash;
            fn hash_no_pad(input: &[F]) -> Self::Hash;
            fn hash_or_noop(inputs: &[F]) -> Self::Hash;
            fn two_to_one(left: Self::Hash, right: Self::Hash) -> Self::Hash;
        }
        /// plonky2::plonk::config::AlgebraicHasher<F>: what `builder.hash_or_noop::<H>` is generic over
        pub trait AlgebraicHasher<F: Gl64>: Hasher<F> {}
        /// plonky2::plonk::config::GenericConfig<D>
        pub trait GenericConfig<const D: usize>: Clone + 'static {
            type F: Gl64;
            type Hasher: AlgebraicHasher<Self::F>;
            type InnerHasher: AlgebraicHasher<Self::F>;
        }
        /// plonky2::plonk::config::PoseidonGoldilocksConfig
        #[derive(Clone, Copy, Debug, PartialEq, Eq)]
        pub struct PoseidonGoldilocksConfig;
        impl GenericConfig<2> for PoseidonGoldilocksConfig {
            type F = crate::field::goldilocks_field::GoldilocksField;
            type Hasher = PoseidonHash;
            type InnerHasher = PoseidonHash;
        }
    }

    pub mod proof {
        use crate::iop::target::Target;
        use crate::plonk::config::GenericConfig;
        use crate::Gl64;
        use std::marker::PhantomData;
        /// plonky2::plonk::proof::ProofWithPublicInputs<F, C, D>: the proof words `CircuitData::prove` returns (the library's order;
        /// `p2mt_proof_to_bytes` gives plonky2's `to_bytes` layout), public inputs last
        #[derive(Clone, Debug)]
        pub struct ProofWithPublicInputs<F: Gl64, C: GenericConfig<D, F = F>, const D: usize> {
            pub(crate) words: Vec<u64>,
            pub public_inputs: Vec<F>,
            pub(crate) _c: PhantomData<C>,
        }
        /// plonky2::plonk::proof::ProofWithPublicInputsTarget<D>: one target per proof word, the public inputs last
        #[derive(Clone, Debug)]
        pub struct ProofWithPublicInputsTarget<const D: usize> {
            pub(crate) words: Vec<u64>,
            pub public_inputs: Vec<Target>,
        }
    }

    pub mod circuit_data {
        use crate::iop::target::Target;
        use crate::iop::witness::PartialWitness;
        use crate::plonk::config::GenericConfig;
        use crate::plonk::proof::ProofWithPublicInputs;
        use crate::{ffi, ok, Gl64};
        use std::marker::PhantomData;
        use std::rc::Rc;

        /// plonky2::fri::FriConfig (the fields the reference reads: `.common.config.fri_config.cap_height`, :99)
        #[derive(Clone, Debug, PartialEq, Eq)]
        pub struct FriConfig {
            pub rate_bits: usize,
            pub cap_height: usize,
            pub proof_of_work_bits: u32,
            pub num_query_rounds: usize,
        }
        /// plonky2::plonk::circuit_data::CircuitConfig
        #[derive(Clone, Debug, PartialEq, Eq)]
        pub struct CircuitConfig {
            pub num_wires: usize,
            pub num_routed_wires: usize,
            pub num_constants: usize,
            pub security_bits: usize,
            pub num_challenges: usize,
            pub zero_knowledge: bool,
            pub max_quotient_degree_factor: usize,
            pub fri_config: FriConfig,
        }
        impl CircuitConfig {
            /// the one configuration the library's prover implements (and the one the reference uses, mmr_plonky2_verifier.rs:30)
            pub fn standard_recursion_config() -> Self {
                CircuitConfig {
                    num_wires: 135,
                    num_routed_wires: 80,
                    num_constants: 2,
                    security_bits: 100,
                    num_challenges: 2,
                    zero_knowledge: false,
                    max_quotient_degree_factor: 8,
                    fri_config: FriConfig { rate_bits: 3, cap_height: 4, proof_of_work_bits: 16, num_query_rounds: 28 },
                }
            }
        }

        pub(crate) struct Handle(pub(crate) *mut ffi::p2mt_circuit_data);
        impl Drop for Handle {
            fn drop(&mut self) {
                unsafe { ffi::p2mt_circuit_destroy(self.0) };
            }
        }
        /// plonky2::plonk::circuit_data::CommonCircuitData<F, D>
        #[derive(Clone)]
        pub struct CommonCircuitData<F: Gl64, const D: usize> {
            pub config: CircuitConfig,
            pub(crate) handle: Rc<Handle>,
            pub(crate) _f: PhantomData<F>,
        }
        /// plonky2::plonk::circuit_data::VerifierOnlyCircuitData<C, D>
        #[derive(Clone)]
        pub struct VerifierOnlyCircuitData<C: GenericConfig<D>, const D: usize> {
            pub(crate) handle: Rc<Handle>,
            pub(crate) _c: PhantomData<C>,
        }
        /// plonky2::plonk::circuit_data::ProverOnlyCircuitData<F, C, D> (`.public_inputs`, mmr_plonky2_verifier.rs:141)
        pub struct ProverOnlyCircuitData<F: Gl64, C: GenericConfig<D, F = F>, const D: usize> {
            pub public_inputs: Vec<Target>,
            pub(crate) _c: PhantomData<(F, C)>,
        }
        /// plonky2::plonk::circuit_data::VerifierCircuitTarget: constants_sigmas_cap [16][4] | circuit_digest [4]
        #[derive(Clone, Debug)]
        pub struct VerifierCircuitTarget {
            pub(crate) words: [u64; 68],
        }
        /// plonky2::plonk::circuit_data::CircuitData<F, C, D>
        pub struct CircuitData<F: Gl64, C: GenericConfig<D, F = F>, const D: usize> {
            pub prover_only: ProverOnlyCircuitData<F, C, D>,
            pub verifier_only: VerifierOnlyCircuitData<C, D>,
            pub common: CommonCircuitData<F, D>,
            pub(crate) handle: Rc<Handle>,
            pub(crate) proof_len: usize,
            pub(crate) num_public_inputs: usize,
        }
        impl<F: Gl64, C: GenericConfig<D, F = F>, const D: usize> CircuitData<F, C, D> {
            /// circuit_data.prove(pw) (mmr_plonky2_verifier.rs:148): witness generation, the three commitments, quotient, openings and FRI
            /// on the device (p2mt_circuit_prove)
            pub fn prove(&self, inputs: PartialWitness<F>) -> anyhow::Result<ProofWithPublicInputs<F, C, D>> {
                let mut words = vec![0u64; self.proof_len];
                let rc = unsafe { ffi::p2mt_circuit_prove(self.handle.0, inputs.raw, words.as_mut_ptr(), words.len()) };
                if rc != ffi::P2MT_OK {
                    let msg = unsafe { std::ffi::CStr::from_ptr(ffi::p2mt_last_error()) }.to_string_lossy().into_owned();
                    anyhow::bail!("prove: p2mt status {rc}: {msg}");
                }
                let public_inputs = words[self.proof_len - self.num_public_inputs..].iter().map(|w| F::from_u64(*w)).collect();
                Ok(ProofWithPublicInputs { words, public_inputs, _c: PhantomData })
            }
            /// circuit_data.verify(proof) (mmr_plonky2_verifier.rs:150)
            pub fn verify(&self, proof_with_pis: ProofWithPublicInputs<F, C, D>) -> anyhow::Result<()> {
                let (mut accepted, mut reason) = (0i32, 0i32);
                let w = &proof_with_pis.words;
                ok(unsafe { ffi::p2mt_circuit_verify(self.handle.0, w.as_ptr(), w.len(), &mut accepted, &mut reason) });
                if accepted == 1 {
                    Ok(())
                } else {
                    anyhow::bail!("proof rejected (reason {reason})")
                }
            }
        }
    }

    pub mod circuit_builder {
        use crate::hash::hash_types::HashOutTarget;
        use crate::iop::target::{BoolTarget, Target};
        use crate::plonk::circuit_data::{CircuitConfig, CircuitData, CommonCircuitData, Handle, ProverOnlyCircuitData, VerifierCircuitTarget,
                                         VerifierOnlyCircuitData};
        use crate::plonk::config::{AlgebraicHasher, GenericConfig};
        use crate::plonk::proof::ProofWithPublicInputsTarget;
        use crate::{ffi, ok, Gl64};
        use std::marker::PhantomData;
        use std::rc::Rc;

        /// plonky2::plonk::circuit_builder::CircuitBuilder<F, D>
        pub struct CircuitBuilder<F: Gl64, const D: usize> {
            raw: *mut ffi::p2mt_circuit_builder,
            config: CircuitConfig,
            _f: PhantomData<F>,
        }
        macro_rules! cb1 {
            ($self:ident, $f:ident $(, $a:expr)*) => {{
                let mut out = 0u64;
                ok(unsafe { ffi::$f($self.raw $(, $a)*, &mut out) });
                Target(out)
            }};
        }
        impl<F: Gl64, const D: usize> CircuitBuilder<F, D> {
            /// CircuitBuilder::<F, D>::new(config) (mmr_plonky2_verifier.rs:31): the library implements standard_recursion_config()
            pub fn new(config: CircuitConfig) -> Self {
                assert!(D == 2, "the library's prover works in the quadratic extension (D = 2)");
                assert!(config == CircuitConfig::standard_recursion_config(), "only CircuitConfig::standard_recursion_config() is implemented");
                let mut p = std::ptr::null_mut();
                ok(unsafe { ffi::p2mt_cb_create(&mut p) });
                CircuitBuilder { raw: p, config, _f: PhantomData }
            }
            pub fn add_virtual_target(&mut self) -> Target {
                cb1!(self, p2mt_cb_add_virtual_target)
            }
            pub fn add_virtual_bool_target_safe(&mut self) -> BoolTarget {
                BoolTarget::new_unsafe(cb1!(self, p2mt_cb_add_virtual_bool_target_safe))
            }
            pub fn add_virtual_hash(&mut self) -> HashOutTarget {
                HashOutTarget { elements: [self.add_virtual_target(), self.add_virtual_target(), self.add_virtual_target(), self.add_virtual_target()] }
            }
            pub fn constant(&mut self, c: F) -> Target {
                cb1!(self, p2mt_cb_constant, c.to_u64())
            }
            pub fn zero(&mut self) -> Target {
                self.constant(F::from_u64(0))
            }
            pub fn one(&mut self) -> Target {
                self.constant(F::from_u64(1))
            }
            pub fn connect(&mut self, x: Target, y: Target) {
                ok(unsafe { ffi::p2mt_cb_connect(self.raw, x.0, y.0) })
            }
            pub fn mul(&mut self, x: Target, y: Target) -> Target {
                cb1!(self, p2mt_cb_mul, x.0, y.0)
            }
            pub fn mul_add(&mut self, x: Target, y: Target, z: Target) -> Target {
                cb1!(self, p2mt_cb_mul_add, x.0, y.0, z.0)
            }
            pub fn not(&mut self, b: BoolTarget) -> BoolTarget {
                BoolTarget::new_unsafe(cb1!(self, p2mt_cb_not, b.target.0))
            }
            pub fn or(&mut self, b1: BoolTarget, b2: BoolTarget) -> BoolTarget {
                BoolTarget::new_unsafe(cb1!(self, p2mt_cb_or, b1.target.0, b2.target.0))
            }
            pub fn is_equal(&mut self, x: Target, y: Target) -> BoolTarget {
                BoolTarget::new_unsafe(cb1!(self, p2mt_cb_is_equal, x.0, y.0))
            }
            fn hash4(&mut self, inputs: Vec<Target>, no_pad: bool) -> HashOutTarget {
                let words: Vec<u64> = inputs.iter().map(|t| t.0).collect();
                let mut out = [0u64; 4];
                ok(unsafe {
                    if no_pad {
                        ffi::p2mt_cb_hash_n_to_hash_no_pad(self.raw, words.as_ptr(), words.len(), out.as_mut_ptr())
                    } else {
                        ffi::p2mt_cb_hash_or_noop(self.raw, words.as_ptr(), words.len(), out.as_mut_ptr())
                    }
                });
                HashOutTarget { elements: out.map(Target) }
            }
            /// builder.hash_or_noop::<PoseidonHash>(inputs) (mmr_plonky2_verifier.rs:34)
            pub fn hash_or_noop<H: AlgebraicHasher<F>>(&mut self, inputs: Vec<Target>) -> HashOutTarget {
                self.hash4(inputs, false)
            }
            /// builder.hash_n_to_hash_no_pad::<PoseidonHash>(inputs) (:79)
            pub fn hash_n_to_hash_no_pad<H: AlgebraicHasher<F>>(&mut self, inputs: Vec<Target>) -> HashOutTarget {
                self.hash4(inputs, true)
            }
            pub fn register_public_input(&mut self, target: Target) {
                self.register_public_inputs(&[target])
            }
            pub fn register_public_inputs(&mut self, targets: &[Target]) {
                let words: Vec<u64> = targets.iter().map(|t| t.0).collect();
                ok(unsafe { ffi::p2mt_cb_register_public_inputs(self.raw, words.as_ptr(), words.len()) })
            }
            /// builder.add_virtual_proof_with_pis(&inner.common) (mmr_plonky2_verifier_1_recursion.rs:95)
            pub fn add_virtual_proof_with_pis(&mut self, common_data: &CommonCircuitData<F, D>) -> ProofWithPublicInputsTarget<D> {
                let mut info = unsafe { std::mem::zeroed::<ffi::p2mt_circuit_info>() };
                ok(unsafe { ffi::p2mt_circuit_get_info(common_data.handle.0, &mut info) });
                let mut words = vec![0u64; info.proof_len as usize];
                ok(unsafe { ffi::p2mt_cb_add_virtual_proof_with_pis(self.raw, common_data.handle.0, words.as_mut_ptr(), words.len()) });
                let npi = info.num_public_inputs as usize;
                let public_inputs = words[words.len() - npi..].iter().map(|w| Target(*w)).collect();
                ProofWithPublicInputsTarget { words, public_inputs }
            }
            /// builder.add_virtual_verifier_data(cap_height) (:98)
            pub fn add_virtual_verifier_data(&mut self, cap_height: usize) -> VerifierCircuitTarget {
                let mut words = [0u64; 68];
                ok(unsafe { ffi::p2mt_cb_add_virtual_verifier_data(self.raw, cap_height as u32, words.as_mut_ptr()) });
                VerifierCircuitTarget { words }
            }
            /// builder.verify_proof::<PoseidonGoldilocksConfig>(&proof, &verifier_data, &inner.common) (:101-104)
            pub fn verify_proof<C: GenericConfig<D, F = F>>(
                &mut self,
                proof_with_pis: &ProofWithPublicInputsTarget<D>,
                inner_verifier_data: &VerifierCircuitTarget,
                inner_common_data: &CommonCircuitData<F, D>,
            ) {
                let w = &proof_with_pis.words;
                ok(unsafe { ffi::p2mt_cb_verify_proof(self.raw, w.as_ptr(), w.len(), inner_verifier_data.words.as_ptr(), inner_common_data.handle.0) })
            }
            /// builder.build::<C>() (mmr_plonky2_verifier.rs:89): selectors, constants, sigmas, the constants_sigmas commitment and the
            /// circuit digest are computed by the library (p2mt_cb_build); consumes the builder, as plonky2's does
            pub fn build<C: GenericConfig<D, F = F>>(self) -> CircuitData<F, C, D> {
                let mut c = std::ptr::null_mut();
                ok(unsafe { ffi::p2mt_cb_build(self.raw, &mut c) });
                let handle = Rc::new(Handle(c));
                let mut info = unsafe { std::mem::zeroed::<ffi::p2mt_circuit_info>() };
                ok(unsafe { ffi::p2mt_circuit_get_info(c, &mut info) });
                let npi = info.num_public_inputs as usize;
                let mut pis = vec![0u64; npi];
                ok(unsafe { ffi::p2mt_circuit_public_inputs(c, pis.as_mut_ptr()) });
                CircuitData {
                    prover_only: ProverOnlyCircuitData { public_inputs: pis.into_iter().map(Target).collect(), _c: PhantomData },
                    verifier_only: VerifierOnlyCircuitData { handle: handle.clone(), _c: PhantomData },
                    common: CommonCircuitData { config: self.config.clone(), handle: handle.clone(), _f: PhantomData },
                    handle,
                    proof_len: info.proof_len as usize,
                    num_public_inputs: npi,
                }
            }
        }
        impl<F: Gl64, const D: usize> Drop for CircuitBuilder<F, D> {
            fn drop(&mut self) {
                unsafe { ffi::p2mt_cb_destroy(self.raw) };
            }
        }
    }
}
