#!/usr/bin/env python3
"""Register / spill / scratch figures of every gfx950 kernel in the built objects (from the code objects' metadata notes).

  python tools/kernel_resources.py [--all] plonky2-merkle-trees_amd/csrc/*.o

Without --all only kernels that are worth a look are printed: more than 128 VGPRs (three waves per SIMD or fewer at 256 threads per
workgroup), SGPRs spilled to VGPR lanes, or a private segment (scratch)."""
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import isa_hazards as ih  # noqa: E402

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
PAT = re.compile(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)"
                 r".*?\.vgpr_spill_count:\s+(\d+)", re.S)


def main():
    show_all = "--all" in sys.argv
    for path in [a for a in sys.argv[1:] if a != "--all"]:
        for blob in ih.extract_gfx950(path):
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob)
                f.flush()
                notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
            for name, priv, ss, vg, vs in PAT.findall(notes):
                if show_all or int(vg) > 128 or int(ss) > 0 or int(priv) > 0:
                    print("%-18s %-100s vgpr %3s  sgpr_spill %3s  vgpr_spill %3s  scratch %4s B" % (path.split("/")[-1], name[:100], vg, ss, vs, priv))


if __name__ == "__main__":
    main()
