#!/bin/bash
# Round-4 additions to tools/profile_round.sh (run ON the GPU box from the repo root through gpurun):
#   tools/profile_round4.sh <tag>   ->  gpurun_out/<tag>_lde/, <tag>_ntt/ (trace + PMC passes of the transform kernels at the >= 1 GB points),
#   gpurun_out/<tag>_verify_timeline.txt, <tag>_prove_batch.csv, <tag>_recursion_batch.csv
set -e
TAG="${1:-r04}"
ROOT="$PWD"
export TMPDIR=/tmp
tools/profile_cmd.sh "${TAG}_lde" tools/ntt_lde_probe.py --what lde --reps 3
tools/profile_cmd.sh "${TAG}_ntt" tools/ntt_lde_probe.py --what ntt --reps 3
echo "transforms done" >> "gpurun_out/${TAG}_progress.log"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/${TAG}_vt" -o v -- python3 "$ROOT/tools/verify_probe.py" 30 > "$ROOT/gpurun_out/${TAG}_verify_probe.log" 2>&1
cd "$ROOT"
python3 tools/verify_probe.py 100 > "gpurun_out/${TAG}_verify_plain.log" 2>&1
DB=$(find "gpurun_out/${TAG}_vt" -name "*_results.db" | head -1)
python3 tools/rocpd_timeline.py "$DB" k_verify_items > "gpurun_out/${TAG}_verify_timeline.txt"
rm -rf "gpurun_out/${TAG}_vt"
echo "verify done" >> "gpurun_out/${TAG}_progress.log"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_pb" -- python3 "$ROOT/tools/prove_batch_probe.py" 4 1 256 2 1 > "$ROOT/gpurun_out/${TAG}_pb.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_rb" -- python3 "$ROOT/tools/recursion_batch_probe.py" 32 2 1 > "$ROOT/gpurun_out/${TAG}_rb.log" 2>&1
cd "$ROOT"
cp "$(find gpurun_out/${TAG}_pb -name '*_kernel_stats.csv' | head -1)" "gpurun_out/${TAG}_prove_batch.csv"
cp "$(find gpurun_out/${TAG}_rb -name '*_kernel_stats.csv' | head -1)" "gpurun_out/${TAG}_recursion_batch.csv"
rm -rf "gpurun_out/${TAG}_pb" "gpurun_out/${TAG}_rb"
python3 tools/prove_batch_probe.py 4 3 256 3 1 > "gpurun_out/${TAG}_pb_plain.log" 2>&1
python3 tools/recursion_batch_probe.py 32 3 2 > "gpurun_out/${TAG}_rb_plain.log" 2>&1
echo "batch done" >> "gpurun_out/${TAG}_progress.log"
