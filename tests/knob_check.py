"""Child process of tests/test_parity_gpu.py::test_env_knobs: the runtime reads its P2MT_* environment knobs once, at
p2mt_init, so every setting is checked in a fresh process.  Exit code 0 = the non-default kernels selected by the
environment are bit-exact against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from conftest import splitmix_leaves  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def main():
    pkg = ge.load_package()
    pkg.init(0)
    o = Oracle()
    # MMR: ragged size crossing every stage (per-lane subtrees or tiles, level kernels, quad, wave)
    for n in (1 << 17) + 4099, 1 << 15, 777:
        leaves = splitmix_leaves(n, 4242 + n)
        m = pkg.MMR.from_leaves(leaves)
        want = o.mmr(leaves)
        assert np.array_equal(m.elements, want.elements), "MMR elements differ at n=%d" % n
        assert np.array_equal(m.bagging_the_peaks(), want.bagging_the_peaks())
    # power-of-two tree with 2^14 leaves (quad-sized levels)
    leaves = splitmix_leaves(1 << 14, 99)
    t = pkg.MerkleTree.build(leaves)
    assert np.array_equal(t.root, o.merkle_build(leaves)[2])
    # commit at the d = 12 shape (register-blocked vs radix-2 LDE) and a small one
    rng = np.random.default_rng(3)
    for log_n, w in ((12, 9), (6, 20)):
        polys = rng.integers(0, 0xFFFFFFFF00000001, size=(w, 1 << log_n), dtype=np.uint64)
        pb = pkg.PolynomialBatch.from_values(polys)
        leaves_o, dig_o, cap_o = o.polynomial_batch_commit(polys, True)
        assert np.array_equal(pb.merkle_tree.cap, cap_o) and np.array_equal(pb.merkle_tree.leaves, leaves_o)
        assert np.array_equal(pb.merkle_tree.digests, dig_o)
    # mmr_plonky2_verifier prove + verify against the committed vectors (tests/golden/prove_vectors.json)
    import hashlib
    import json
    from circuit_cases import assign
    for c in json.load(open(os.path.join(ROOT, "tests", "golden", "prove_vectors.json")))["cases"]:
        case = (c["leaf"], np.array(c["siblings"], np.uint64).reshape(-1, 4), np.array(c["lefts"], np.uint8),
                np.array(c["peaks"], np.uint64).reshape(-1, 4), np.array(c["root"], np.uint64))
        cd, leaf_t, proof_ts, peak_ts = pkg.verify_mmr_proof_circuit(len(case[1]), len(case[3]))
        pw = pkg.PartialWitness()
        assign(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs, case, pw.set_target)
        proof = cd.prove(pw)
        assert hashlib.sha256(proof.astype("<u8").tobytes()).hexdigest() == c["proof_sha256"], "proof differs: " + c["name"]
        assert cd.verify(proof)
    print("knobs ok:", {k: v for k, v in os.environ.items() if k.startswith("P2MT_")})


if __name__ == "__main__":
    main()
