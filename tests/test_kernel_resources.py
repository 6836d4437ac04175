"""CPU-only: register / spill / scratch budgets of the shipped throughput kernels, read from the built code objects.

Why a test: in the stage-1 tree kernels the spilled values were the launch's extra HBM-side traffic (r04: FETCH_SIZE 589 MB per
2^24-leaf launch for 134 MB of leaves; round 5 found the lane's loop-invariant indices spilled and reloaded in every one of the 15 steps,
plus each 64-byte leaf sector fetched four times -- 191 MB once both were gone, profiles/r05_stage1_traffic.txt).  A change that makes
the allocator spill them again costs no time (the kernel is issue-bound) and so would go unnoticed: it fails here."""
import glob
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards as ih  # noqa: E402

CSRC = os.path.join(ROOT, "plonky2-merkle-trees_amd", "csrc")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
PAT = re.compile(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)"
                 r".*?\.vgpr_spill_count:\s+(\d+)", re.S)

# kernel name fragment -> (max VGPRs, max spilled VGPRs, max scratch bytes per lane).  128 VGPRs = four waves per SIMD at 256 lanes.
BUDGET = {
    "k_mmr_subtreeILj4ELi256ELi5ELi4": (128, 7, 24),   # the dominant launch of the headline build (r04: 9 spilled, 40 B)
    "k_mmr_subtreeILj3ELi256ELi5ELi4": (128, 7, 24),
    "k_mmr_subtreeILj2ELi256ELi5ELi4": (128, 7, 24),
    "k_merkle_subtreeILj4": (128, 7, 24),              # (r04: 7, 32 B)
    "k_merkle_subtreeILj3": (128, 7, 24),
    "k_merkle_subtreeILj2": (128, 7, 24),
    "k_mmr_levelILi2ELi5": (128, 4, 20),               # one hash per lane: a handful of reloads per 10 k instructions
    "k_merkle_levelILi2ELi5": (128, 4, 20),
    "k_two_to_one_batchILi2ELi5": (128, 4, 20),
    "k_hash_columnsILi2ELi5": (128, 18, 44),
    "k_fri_pow_queueILi2ELi5": (128, 20, 72),           # (spills sit in the prologue, the shared first round and the rare per-lane redo; none inside the permutation: test_no_scratch_access_inside_a_permutation)
    "k_tree_top": (160, 0, 0),                         # the latency layouts of the one-launch build: no scratch on a dependent chain
    "k_mmr_level_quad": (136, 0, 0),
    "k_merkle_level_quad": (136, 0, 0),
    "k_hash_columns_quad": (168, 0, 0),
    "k_quotient": (148, 0, 0),
}


@pytest.fixture(scope="module")
def kernels():
    import __graft_entry__ as ge
    ge.load_package()
    if not os.path.exists(READELF) or not os.path.exists(ih.OBJDUMP):
        pytest.skip("llvm-readelf / llvm-objdump not in this image")
    out = {}
    for path in sorted(glob.glob(os.path.join(CSRC, "*.o"))):
        for blob in ih.extract_gfx950(path):
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob)
                f.flush()
                notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
            for name, priv, ss, vg, vs in PAT.findall(notes):
                out[name] = {"scratch": int(priv), "sgpr_spill": int(ss), "vgpr": int(vg), "vgpr_spill": int(vs), "obj": os.path.basename(path)}
    assert len(out) > 50
    return out


def test_budgets(kernels):
    seen = set()
    for frag, (max_vgpr, max_spill, max_scratch) in BUDGET.items():
        hit = [(n, k) for n, k in kernels.items() if frag in n]
        assert hit, "no kernel matches %s: renamed? update the budget table" % frag
        for n, k in hit:
            seen.add(n)
            assert k["vgpr"] <= max_vgpr, (n, k)
            assert k["vgpr_spill"] <= max_spill, (n, k)
            assert k["scratch"] <= max_scratch, (n, k)
    assert len(seen) >= len(BUDGET)


def test_no_scratch_access_inside_a_permutation(kernels):
    """the stage-1 kernels may keep a few values in scratch around the rare exact redo, but no basic block that holds a matrix-pipe MDS
    layer -- the bodies of the permutation, where the 10 k instructions of a hash are -- touches scratch"""
    obj = os.path.join(CSRC, "p2mt_mmr.o")
    checked = 0
    for blob in ih.extract_gfx950(obj):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            dis = subprocess.run([ih.OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur, lines = None, []
        funcs = {}
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                cur = m.group(1)
                funcs[cur] = []
            elif cur is not None:
                funcs[cur].append(ln)
        for name, body in funcs.items():
            if "k_mmr_subtreeILj4ELi256ELi5ELi4" not in name:
                continue
            # split at branch instructions: a run of straight-line code with an MFMA in it must hold no scratch_ access
            run, runs = [], []
            for ln in body:
                run.append(ln)
                if re.search(r"\bs_cbranch|\bs_branch|\bs_setpc|\bs_endpgm", ln):
                    runs.append(run)
                    run = []
            runs.append(run)
            hot = [r for r in runs if any("v_mfma" in x for x in r)]
            assert len(hot) >= 4, (name, len(hot))
            for r in hot:
                # between the first and the last matrix-pipe instruction of the run: the body of the permutation proper
                idx = [i for i, x in enumerate(r) if "v_mfma" in x]
                inner = r[idx[0]:idx[-1] + 1]
                assert not any("scratch_" in x for x in inner), (name, [x for x in inner if "scratch_" in x][:3])
                checked += 1
            assert sum("scratch_" in x for x in body) <= 16, name  # (r04: 31 static scratch accesses, five of them reloads in every step)
    assert checked >= 4
