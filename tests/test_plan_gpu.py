"""GPU parity of the one-launch tree build (csrc/p2mt_plan.hip: stage 1 and every level above it as dependency-ordered work items of
one grid, hand-offs through per-chunk counters) against the oracle's `for leaf { add_leaf }`
(/root/reference/src/mmr/merkle_mountain_ranges.rs:89-120).  The path is off by default (it measured no faster than the separate
launches, profiles/r05_one_launch_build.txt); these tests turn it on through p2mt_debug_plan_knobs and compare every node."""
import numpy as np
import pytest

import __graft_entry__ as ge
from conftest import splitmix_leaves

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.init(0)
    return p


@pytest.fixture()
def plan(pkg):
    lib = pkg._native.lib()

    def on(order, min_log=16, tq=16, tw=12):
        pkg._native.check(lib.p2mt_debug_plan_knobs(1, min_log, order, tq, tw))
    yield on
    pkg._native.check(lib.p2mt_debug_plan_knobs(0, 18, 0, 16, 12))


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("log_n,tq,tw", [(16, 16, 12), (18, 16, 12), (18, 10, 6), (20, 14, 9)])
def test_one_launch_build_every_node(pkg, oracle, plan, order, log_n, tq, tw):
    """every element, the root and a proof: one-launch build == oracle, for both ticket orders and several Q / W thresholds (which
    levels run one hash per lane, four lanes per hash, one wavefront per hash)"""
    n = 1 << log_n
    leaves = splitmix_leaves(n, 0x5EED0500 + log_n)
    plan(order, 16, tq, tw)
    m = pkg.MMR.from_leaves(leaves)
    om = oracle.mmr(leaves)
    assert np.array_equal(m.elements, om.elements)
    root = m.bagging_the_peaks()
    assert np.array_equal(root, om.bagging_the_peaks())
    pr = m.get_proof_normal_index(n - 3)
    assert pr.verify(int(leaves[n - 3]), root)


def test_one_launch_blocks_inside_a_ragged_extend(pkg, oracle, plan):
    """an extend whose range holds aligned 2^16- / 2^17-leaf blocks between ragged edges, on a non-empty MMR: the blocks take the
    one-launch path, the edges and the carry chains above the blocks the level launches; all nodes == oracle"""
    plan(1, 16)
    first, more = 3 * (1 << 15) + 77, (1 << 18) + 12345
    leaves = splitmix_leaves(first + more, 0x5EED0555)
    m = pkg.MMR.from_leaves(leaves[:first])
    m.extend(leaves[first:])
    om = oracle.mmr(leaves)
    assert len(m) == len(om.elements)
    assert np.array_equal(m.elements, om.elements)
    assert np.array_equal(m.bagging_the_peaks(), om.bagging_the_peaks())
    assert np.array_equal(m.get_peaks(), om.get_peaks())


def test_profile_rows_are_consistent(pkg, plan):
    """the per-item device-clock rows: one row per work item, start <= ready <= end, every consumer ready after its producers' end"""
    lib = pkg._native.lib()
    plan(0, 16)
    n = 1 << 18
    leaves = splitmix_leaves(n, 0x5EED0777)
    pkg._native.check(lib.p2mt_debug_plan_profile(1))
    try:
        m = pkg.MMR.from_leaves(leaves)
        m.bagging_the_peaks()
    finally:
        pkg._native.check(lib.p2mt_debug_plan_profile(0))
    rows = np.zeros((1 << 15, 8), np.uint64)
    cnt = lib.p2mt_debug_plan_profile_read(pkg._native.ptr(rows), rows.shape[0])
    assert cnt > 256
    r = rows[:cnt].astype(np.int64)
    assert (r[:, 3] <= r[:, 5]).all()
    up = r[r[:, 0] != 0]
    assert (up[:, 3] <= up[:, 4]).all() and (up[:, 4] <= up[:, 5]).all()
    # level by level: the first item of a level cannot be ready before some item of the level below has ended
    for h in sorted(set(up[:, 1].tolist())):
        below = r[r[:, 1] == h - 1]
        assert up[up[:, 1] == h][:, 4].min() >= below[:, 5].min()
