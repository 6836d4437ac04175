#!/bin/bash
# All rocprofv3 evidence of a round in one call (run ON the GPU box from the repo root through gpurun):
#   tools/profile_round.sh <tag>          -> gpurun_out/<tag>_mmr/summary.txt, gpurun_out/<tag>_prove_d6.txt, gpurun_out/<tag>_recursion.txt
# 1. tools/profile_mmr.sh: kernel trace + PMC passes of the default bench (2^24-leaf MMR build);
# 2. per-kernel timeline of ONE mmr_plonky2_verifier prove and of ONE mmr_plonky2_verifier_1_recursion proof (rocpd database ->
#    tools/rocpd_timeline.py), each next to the un-profiled ms/proof of the same build.
set -e
TAG="${1:-r03}"
ROOT="$PWD"
export TMPDIR=/tmp
tools/profile_mmr.sh "${TAG}_mmr"
echo "mmr done" >> "gpurun_out/${TAG}_progress.log"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/${TAG}_p6" -o prove -- python3 "$ROOT/bench.py" --workload prove --steps 20 --warmup 3 --no-cpu-baseline --threads 1 > "$ROOT/gpurun_out/${TAG}_p6.log" 2>&1
echo "prove trace done" >> "$ROOT/gpurun_out/${TAG}_progress.log"
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/${TAG}_rec" -o rec -- python3 "$ROOT/bench.py" --workload recursion --steps 5 --warmup 2 --no-cpu-baseline > "$ROOT/gpurun_out/${TAG}_rec.log" 2>&1
echo "recursion trace done" >> "$ROOT/gpurun_out/${TAG}_progress.log"
cd "$ROOT"
python3 bench.py --workload prove --steps 30 --warmup 5 --no-cpu-baseline --threads 1 > "gpurun_out/${TAG}_p6_plain.log" 2>&1
python3 bench.py --workload recursion --steps 10 --warmup 2 --no-cpu-baseline > "gpurun_out/${TAG}_rec_plain.log" 2>&1
DB6=$(find "gpurun_out/${TAG}_p6" -name "*_results.db" | head -1)
DBR=$(find "gpurun_out/${TAG}_rec" -name "*_results.db" | head -1)
python3 tools/rocpd_timeline.py "$DB6" k_witness_lds > "gpurun_out/${TAG}_prove_d6.txt"
python3 tools/rocpd_timeline.py "$DBR" k_witness_flow > "gpurun_out/${TAG}_recursion.txt"
rm -rf "gpurun_out/${TAG}_p6" "gpurun_out/${TAG}_rec"   # the databases are large; the summaries are what is kept
