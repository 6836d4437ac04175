//! /root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:20-140: the inner circuit (Merkle path, peaks as public inputs) and
//! the outer circuit (plonky2's in-circuit verifier of the inner proof -- built by the library, p2mt_cb_verify_proof -- plus the
//! peaks check and the root), with the reference's names, arguments and tuple returns.
use crate::mmr::common::{equal, or_list, pick_hash};
use crate::plonk::{BoolTarget, CircuitBuilder, CircuitData, CommonCircuitData, HashOutTarget, ProofWithPublicInputsTarget, Target, VerifierCircuitTarget};

/// :20-75 -> (circuit data, leaf target, path element targets); public inputs: the peaks, 4 words each.
pub fn verify_inner_merkle_proof_circuit(nr_merkle_proof_elms: usize, nr_peaks: usize) -> (CircuitData, Target, Vec<(HashOutTarget, BoolTarget)>) {
    let mut proof_targets: Vec<(HashOutTarget, BoolTarget)> = Vec::new();
    let mut builder = CircuitBuilder::new();
    let leaf_to_prove = builder.add_virtual_target();
    let mut next_hash = builder.hash_or_noop([leaf_to_prove].to_vec());
    for _ in 0..nr_merkle_proof_elms {
        let merkle_proof_elm = builder.add_virtual_hash();
        let elm_on_left = builder.add_virtual_bool_target_safe();
        proof_targets.push((merkle_proof_elm, elm_on_left));
        let option1 = builder.hash_or_noop([merkle_proof_elm.elements.to_vec(), next_hash.elements.to_vec()].concat());
        let option2 = builder.hash_or_noop([next_hash.elements.to_vec(), merkle_proof_elm.elements.to_vec()].concat());
        next_hash = pick_hash(&mut builder, option1, option2, elm_on_left);
    }
    let mut equals: Vec<BoolTarget> = Vec::new();
    for _ in 0..nr_peaks {
        let peak = builder.add_virtual_hash();
        for elm in peak.elements {
            builder.register_public_input(elm);
        }
        equals.push(equal(&mut builder, peak, next_hash));
    }
    let hash_in_peaks = or_list(&mut builder, equals);
    let one = builder.one();
    builder.connect(one, hash_in_peaks.target);
    (builder.build(), leaf_to_prove, proof_targets)
}

/// :84-140 -> (circuit data, where the inner proof goes in the witness, where the inner verifier data goes, peak targets).
/// The inner proof's first four public inputs (its FIRST peak: quirk Q4 of SURVEY.md App. C) must appear among the given peaks.
pub fn complete_verification_circuit_with_inner_proof(
    inner_proof_circuit_data_common: CommonCircuitData,
    nr_peaks: usize,
) -> (CircuitData, ProofWithPublicInputsTarget, VerifierCircuitTarget, Vec<HashOutTarget>) {
    let mut builder = CircuitBuilder::new();
    let prev_proof_target = builder.add_virtual_proof_with_pis(&inner_proof_circuit_data_common);
    let prev_proof_verifier_data = builder.add_virtual_verifier_data(4);  // inner.config.fri_config.cap_height
    builder.verify_proof(&prev_proof_target, &prev_proof_verifier_data, &inner_proof_circuit_data_common);
    let mut targets: Vec<HashOutTarget> = Vec::new();
    let mut peaks: Vec<HashOutTarget> = Vec::new();
    let mut equals: Vec<BoolTarget> = Vec::new();
    let prev_hash = HashOutTarget::from_vec(prev_proof_target.public_inputs[0..4].to_vec());
    for _ in 0..nr_peaks {
        let peak = builder.add_virtual_hash();
        peaks.push(peak);
        targets.push(peak);
        equals.push(equal(&mut builder, peak, prev_hash));
    }
    let hash_in_peaks = or_list(&mut builder, equals);
    let one = builder.one();
    builder.connect(one, hash_in_peaks.target);
    if peaks.len() > 1 {
        let root = builder.hash_n_to_hash_no_pad(peaks.into_iter().flat_map(|x| x.elements).collect());
        builder.register_public_inputs(&root.elements);
    } else {
        builder.register_public_inputs(&peaks[0].elements);
    }
    (builder.build(), prev_proof_target, prev_proof_verifier_data, targets)
}
