"""Mirror of src/mmr/mmr_plonky2_verifier_1_recursion.rs: the INNER circuit (verify_inner_merkle_proof_circuit, :20-75),
written against circuit.CircuitBuilder as the reference writes it against plonky2's.

The OUTER circuit (complete_verification_circuit_with_inner_proof, :84-140) calls plonky2's in-circuit verifier
(builder.verify_proof), which the library builds (csrc/p2mt_recursion.hip: CosetInterpolation / RandomAccess / Reducing /
ReducingExtension / ArithmeticExtension / MulExtension / BaseSum / PoseidonMds gates and their generators)."""
from .circuit import CircuitBuilder
from .mmr_plonky2_verifier import equal, or_list, pick_hash


def verify_inner_merkle_proof_circuit(nr_merkle_proof_elms, nr_peaks):
    """:20-75 -> (circuit_data, leaf target, [(HashOutTarget, BoolTarget)]); the public inputs are the peaks"""
    proof_targets = []
    builder = CircuitBuilder()
    leaf_to_prove = builder.add_virtual_target()
    next_hash = builder.hash_or_noop([leaf_to_prove])
    for _ in range(nr_merkle_proof_elms):
        merkle_proof_elm = builder.add_virtual_hash()
        elm_on_left = builder.add_virtual_bool_target_safe()
        proof_targets.append((merkle_proof_elm, elm_on_left))
        option1 = builder.hash_or_noop(merkle_proof_elm + next_hash)
        option2 = builder.hash_or_noop(next_hash + merkle_proof_elm)
        next_hash = pick_hash(builder, option1, option2, elm_on_left)
    equals = []
    for _ in range(nr_peaks):
        peak = builder.add_virtual_hash()
        for elm in peak:
            builder.register_public_input(elm)
        equals.append(equal(builder, peak, next_hash))
    hash_in_peaks = or_list(builder, equals)
    builder.connect(builder.one(), hash_in_peaks)
    data = builder.build()
    return data, leaf_to_prove, proof_targets


def complete_verification_circuit_with_inner_proof(inner_proof_circuit_data_common, nr_peaks):
    """:84-140 -> (circuit_data, ProofWithPublicInputsTarget, VerifierCircuitTarget, [peak HashOutTargets]).
    Outer circuit: verifies the inner proof in-circuit (plonky2's recursive verifier, built by the library), checks that the
    inner proof's first four public inputs -- its FIRST peak, quirk Q4 of SURVEY.md App. C -- appear among the given peaks, and
    exposes the bagged root as its public input."""
    inner = inner_proof_circuit_data_common
    builder = CircuitBuilder()
    prev_proof_target = builder.add_virtual_proof_with_pis(inner)
    prev_proof_verifier_data = builder.add_virtual_verifier_data(4)   # inner.config.fri_config.cap_height
    builder.verify_proof(prev_proof_target, prev_proof_verifier_data, inner)
    targets, peaks, equals = [], [], []
    prev_hash = prev_proof_target.public_inputs[0:4]
    for _ in range(nr_peaks):
        peak = builder.add_virtual_hash()
        peaks.append(peak)
        targets.append(peak)
        equals.append(equal(builder, peak, prev_hash))
    hash_in_peaks = or_list(builder, equals)
    builder.connect(builder.one(), hash_in_peaks)
    if len(peaks) > 1:
        root = builder.hash_n_to_hash_no_pad([e for p in peaks for e in p])
        builder.register_public_inputs(root)
    else:
        builder.register_public_inputs(peaks[0])
    return builder.build(), prev_proof_target, prev_proof_verifier_data, targets
