#!/bin/bash
# Round 5: everything tools/profile_round.sh collects (2^24-leaf MMR build: kernel trace + PMC passes; per-kernel timelines of one prove
# and one recursion proof) plus the timeline of ONE verification with the transcript on the host (run ON the GPU box from the repo root
# through gpurun):   tools/profile_round5.sh <tag>
set -e
TAG="${1:-r05}"
ROOT="$PWD"
export TMPDIR=/tmp
tools/profile_round.sh "$TAG"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/${TAG}_vt" -o v -- python3 "$ROOT/tools/verify_probe.py" 30 > "$ROOT/gpurun_out/${TAG}_verify_probe.log" 2>&1
cd "$ROOT"
python3 tools/verify_probe.py 200 > "gpurun_out/${TAG}_verify_plain.log" 2>&1
python3 tools/prove_probe.py 100 > "gpurun_out/${TAG}_prove_plain.log" 2>&1
DB=$(find "gpurun_out/${TAG}_vt" -name "*_results.db" | head -1)
python3 tools/rocpd_timeline.py "$DB" k_verify_leaf_digests > "gpurun_out/${TAG}_verify_timeline.txt"
rm -rf "gpurun_out/${TAG}_vt"
echo "verify done" >> "gpurun_out/${TAG}_progress.log"
