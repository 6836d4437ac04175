"""Mirror of the reference's circuit code: src/mmr/common.rs (equal, or_list, pick_hash) and
src/mmr/mmr_plonky2_verifier.rs (verify_mmr_proof_circuit), written against circuit.CircuitBuilder exactly as the
reference writes them against plonky2's."""
from .circuit import CircuitBuilder


def equal(builder, first, second):
    """common.rs:5-16 (the reference combines the four element equalities with `or`)."""
    elm0 = builder.is_equal(first[0], second[0])
    elm1 = builder.is_equal(first[1], second[1])
    elm2 = builder.is_equal(first[2], second[2])
    elm3 = builder.is_equal(first[3], second[3])
    elm0_or_elm1 = builder.or_(elm0, elm1)
    elm2_or_elm3 = builder.or_(elm2, elm3)
    return builder.or_(elm0_or_elm1, elm2_or_elm3)


def or_list(builder, ins):
    """common.rs:18-38"""
    assert len(ins) > 0
    if len(ins) == 1:
        return ins[0]
    if len(ins) == 2:
        return builder.or_(ins[0], ins[1])
    pairs = []
    for i in range(0, len(ins), 2):
        pairs.append(builder.or_(ins[i], ins[i + 1]) if i + 1 < len(ins) else ins[i])
    return or_list(builder, pairs)


def pick_hash(builder, option1, option2, pick_left):
    """common.rs:42-58: option1 if pick_left else option2"""
    opposite = builder.not_(pick_left)
    t = [builder.mul(option2[i], opposite) for i in range(4)]
    return [builder.mul_add(option1[i], pick_left, t[i]) for i in range(4)]


def verify_mmr_proof_circuit(nr_merkle_proof_elms, nr_peaks):
    """mmr_plonky2_verifier.rs:13-91 -> (circuit_data, leaf target, [(HashOutTarget, BoolTarget)], [peak HashOutTargets])"""
    proof_targets, peak_targets = [], []
    builder = CircuitBuilder()
    leaf_to_prove = builder.add_virtual_target()
    next_hash = builder.hash_or_noop([leaf_to_prove])
    for _ in range(nr_merkle_proof_elms):
        merkle_proof_elm = builder.add_virtual_hash()
        elm_on_left = builder.add_virtual_bool_target_safe()
        proof_targets.append((merkle_proof_elm, elm_on_left))
        option1 = builder.hash_or_noop(merkle_proof_elm + next_hash)  # sibling on the left
        option2 = builder.hash_or_noop(next_hash + merkle_proof_elm)  # sibling on the right
        next_hash = pick_hash(builder, option1, option2, elm_on_left)
    peaks, equals = [], []
    for _ in range(nr_peaks):
        peak = builder.add_virtual_hash()
        peaks.append(peak)
        peak_targets.append(peak)
        equals.append(equal(builder, peak, next_hash))
    hash_in_peaks = or_list(builder, equals)
    builder.connect(builder.one(), hash_in_peaks)
    if len(peaks) > 1:
        root = builder.hash_n_to_hash_no_pad([e for p in peaks for e in p])
        builder.register_public_inputs(root)
    else:
        builder.register_public_inputs(peaks[0])
    data = builder.build()
    return data, leaf_to_prove, proof_targets, peak_targets
