#!/bin/bash
# per-kernel timeline of ONE mmr_plonky2_verifier_1_recursion proof (run ON the GPU box from the repo root):  tools/profile_recursion.sh <tag>
set -e
TAG="${1:-rec}"
ROOT="$PWD"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/${TAG}_rec" -o rec -- python3 "$ROOT/bench.py" --workload recursion --steps 5 --warmup 2 --no-cpu-baseline > "$ROOT/gpurun_out/${TAG}_rec.log" 2>&1
cd "$ROOT"
DBR=$(find "gpurun_out/${TAG}_rec" -name "*_results.db" | head -1)
python3 tools/rocpd_timeline.py "$DBR" k_witness_flow > "gpurun_out/${TAG}_recursion.txt"
rm -rf "gpurun_out/${TAG}_rec"
