//! /root/reference/src/mmr/mmr_plonky2_verifier.rs:13-91: `verify_mmr_proof_circuit` with the reference's name, arguments and tuple
//! return, over the library's circuit builder (crate::plonk).  The returned `CircuitData` proves on the MI355X: `data.prove(pw)`.
use crate::mmr::common::{equal, or_list, pick_hash};
use crate::plonk::{BoolTarget, CircuitBuilder, CircuitData, HashOutTarget, Target};

/// -> (circuit data, leaf target, (sibling hash, sibling-on-the-left flag) per path element, peak targets); the public input is
/// the bagged root (the single peak itself when there is one).
pub fn verify_mmr_proof_circuit(nr_merkle_proof_elms: usize, nr_peaks: usize) -> (CircuitData, Target, Vec<(HashOutTarget, BoolTarget)>, Vec<HashOutTarget>) {
    let mut proof_targets: Vec<(HashOutTarget, BoolTarget)> = Vec::new();
    let mut peak_targets: Vec<HashOutTarget> = Vec::new();
    let mut builder = CircuitBuilder::new();  // CircuitConfig::standard_recursion_config() (:30)
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
    let mut peaks: Vec<HashOutTarget> = Vec::new();
    let mut equals: Vec<BoolTarget> = Vec::new();
    for _ in 0..nr_peaks {
        let peak = builder.add_virtual_hash();
        peaks.push(peak);
        peak_targets.push(peak);
        equals.push(equal(&mut builder, peak, next_hash));
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
    (builder.build(), leaf_to_prove, proof_targets, peak_targets)
}
