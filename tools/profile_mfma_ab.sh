#!/bin/bash
# rocprofv3 evidence for the matrix-pipe MDS A/B: run ON the GPU box from the repo root through gpurun:
#   tools/profile_mfma_ab.sh <out_dir_under_gpurun_out> [variants, default "2,0 2,2"]
# Per variant (p2mt_set_variant: at the end of round 3 2,0 = the default, dense MDS layers as one 32x32x32 i8 MFMA per limb; 2,5 = VALU
# MDS; 2,6 = VALU MDS with the previous multiply; 2,2 = MDS layers on v_mfma_i32_4x4x4_16b_i8): one kernel-trace pass and one PMC pass
# (VALU / MFMA instruction counts, matrix-pipe busy and co-execution cycles, GRBM_GUI_ACTIVE from the same dispatches).
set -e
OUT="$PWD/gpurun_out/${1:-prof_mfma_ab}"
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp
VARIANTS="${2:-2,0 2,2}"
for V in $VARIANTS; do
  T="v${V/,/_}"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$T/runc" -- python3 "$BENCH" --variant $V --steps 5 --warmup 2 --no-cpu-baseline --no-prove > "$OUT/trace_$T.log" 2>&1
  echo "trace $V done" >> "$OUT/progress.log"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_$T/runc" -- python3 "$BENCH" --variant $V --steps 2 --warmup 1 --no-cpu-baseline --no-prove > "$OUT/pmc_$T.log" 2>&1
  echo "pmc $V done" >> "$OUT/progress.log"
done
cd - > /dev/null
for V in ${VARIANTS//,/_}; do
  mkdir -p "$OUT/sum_$V/trace" && cp -r "$OUT/trace_v$V"/* "$OUT/sum_$V/trace/" && cp -r "$OUT/pmc_v$V" "$OUT/sum_$V/pmc_sq"
  python3 tools/summarize_rocprof.py "$OUT/sum_$V" 3 > "$OUT/summary_v$V.txt"
done
