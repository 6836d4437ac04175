#!/usr/bin/env python3
"""Writes tools/plonky2_crosscheck/p2mt_vectors.json: what this repo's restatement of plonky2 computes for the smallest circuit of
the reference -- verify_mmr_proof_circuit(1, 2) (/root/reference/src/mmr/mmr_plonky2_verifier.rs:13-91) for the MMR of leaves
(1, 2, 3), leaf index 1 -- so that a maintainer with cargo can compare it with real plonky2 (tools/plonky2_crosscheck/README.md).
The values come from the oracle (oracle/circuit.py + oracle/*.c); tests/test_circuit_gpu.py shows the HIP path equals it bit for bit."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle_lib import Oracle  # noqa: E402
from oracle import circuit as OC  # noqa: E402


def main():
    o = Oracle()
    leaves = np.array([1, 2, 3], np.uint64)
    m = o.mmr(leaves)
    pr = m.get_proof_normal_index(1)
    root = m.bagging_the_peaks()
    cd, leaf_t, proof_ts, peak_ts = OC.verify_mmr_proof_circuit(o, len(pr["siblings"]), len(pr["peaks"]))
    pw = {leaf_t: 2}
    for (ht, bt), s, l in zip(proof_ts, pr["siblings"], pr["lefts"]):
        for k in range(4):
            pw[ht[k]] = int(s[k])
        pw[bt] = int(l)
    for pt, pk in zip(peak_ts, pr["peaks"]):
        for k in range(4):
            pw[pt[k]] = int(pk[k])
    for k, t in enumerate(cd.public_inputs):
        pw[t] = int(root[k])
    tr = {}
    proof = cd.prove(pw, trace=tr)
    assert cd.verify(proof) == (True, 0)
    # the same circuit and statement under every alternative of the recalled conventions that has one (oracle/circuit.py CONVENTIONS):
    # the circuit digest opens the transcript, so each alternative has its own challenges and its own proof
    variants = {}
    default_mode = OC.CONVENTIONS["digest_domain_separator"]
    for mode in OC.DIGEST_DOMAIN_SEPARATORS:
        OC.CONVENTIONS["digest_domain_separator"] = mode
        try:
            vcd, vleaf, vproof_ts, vpeak_ts = OC.verify_mmr_proof_circuit(o, len(pr["siblings"]), len(pr["peaks"]))
            vpw = {vleaf: 2}
            for (ht, bt), sgl, l in zip(vproof_ts, pr["siblings"], pr["lefts"]):
                for k in range(4):
                    vpw[ht[k]] = int(sgl[k])
                vpw[bt] = int(l)
            for pt, pk in zip(vpeak_ts, pr["peaks"]):
                for k in range(4):
                    vpw[pt[k]] = int(pk[k])
            for k, t in enumerate(vcd.public_inputs):
                vpw[t] = int(root[k])
            vproof = vcd.prove(vpw)
            assert vcd.verify(vproof) == (True, 0)
            variants[mode] = {"circuit_digest": [int(x) for x in vcd.circuit_digest], "proof_words": [int(x) for x in vproof],
                              "is_default": mode == default_mode}
        finally:
            OC.CONVENTIONS["digest_domain_separator"] = default_mode
    assert variants[default_mode]["proof_words"] == [int(x) for x in proof]
    out = {
        "circuit": "verify_mmr_proof_circuit(nr_merkle_proof_elms = %d, nr_peaks = %d)" % (len(pr["siblings"]), len(pr["peaks"])),
        "mmr_leaves": [1, 2, 3], "leaf_index": 1,
        "merkle_proof": [{"hash": [int(x) for x in s], "on_left": bool(l)} for s, l in zip(pr["siblings"], pr["lefts"])],
        "peaks": [[int(x) for x in p] for p in pr["peaks"]], "root": [int(x) for x in root],
        "degree_bits": cd.degree_bits,
        "gates_sorted": [OC.GATE_ID[g] for g in cd.gates],
        "selector_groups": [list(g) for g in cd.groups], "selector_indices": cd.selector_indices,
        "gate_rows": [OC.GATE_ID[g[0]] for g in cd.gate_instances],
        "k_is_first4": [int(x) for x in cd.k_is[:4]],
        "circuit_digest": [int(x) for x in cd.circuit_digest],
        "constants_sigmas_cap_0": [int(x) for x in cd.cs_cap.reshape(-1, 4)[0]],
        "public_inputs_hash": [int(x) for x in tr["pi_hash"]],
        "wires_cap_0": [int(x) for x in tr["caps"][0].reshape(-1, 4)[0]],
        "plonk_betas": [int(x) for x in tr["betas"]], "plonk_gammas": [int(x) for x in tr["gammas"]],
        "plonk_alphas": [int(x) for x in tr["alphas"]], "plonk_zeta": [int(x) for x in tr["zeta"]],
        "proof_layout": "wires_cap[16][4] | zs_partial_products_cap[16][4] | quotient_polys_cap[16][4] | constants[%d][2] | "
                        "plonk_sigmas[80][2] | wires[135][2] | plonk_zs[2][2] | plonk_zs_next[2][2] | partial_products[18][2] | "
                        "quotient_polys[16][2] | FriProof | public_inputs[4]" % (cd.num_selectors + 2),
        "proof_words": [int(x) for x in proof],
        "digest_domain_separator_default": default_mode,
        "digest_domain_separator_variants": variants,
        "notes": ["unused PublicInputGate wires are ZERO here (plonky2's randomize_unused_pi_wires puts random values there): "
                  "wires_cap and everything after it differ from a plonky2-generated proof, but this proof is a valid witness and "
                  "plonky2's verify() must accept it if every convention in DESIGN.md's checklist matches",
                  "pow_witness is the smallest valid one"],
    }
    path = os.path.join(ROOT, "tools", "plonky2_crosscheck", "p2mt_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, len(proof), "proof words; digest", out["circuit_digest"])


if __name__ == "__main__":
    main()
