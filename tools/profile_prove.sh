#!/bin/bash
# per-kernel timeline of ONE mmr_plonky2_verifier prove (config 3) (run ON the GPU box from the repo root):  tools/profile_prove.sh <tag>
set -e
TAG="${1:-p6}"
ROOT="$PWD"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/${TAG}_p6" -o prove -- python3 "$ROOT/bench.py" --workload prove --steps 20 --warmup 3 --no-cpu-baseline --threads 1 > "$ROOT/gpurun_out/${TAG}_p6.log" 2>&1
cd "$ROOT"
DB6=$(find "gpurun_out/${TAG}_p6" -name "*_results.db" | head -1)
python3 tools/rocpd_timeline.py "$DB6" k_witness_lds > "gpurun_out/${TAG}_prove_d6.txt"
rm -rf "gpurun_out/${TAG}_p6"
