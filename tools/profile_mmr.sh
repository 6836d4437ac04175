#!/bin/bash
# rocprofv3 evidence for the headline bench (run ON the GPU box, from the repo root, through gpurun):
#   tools/profile_mmr.sh <out_dir_under_gpurun_out>
# kernel trace + three PMC passes (one counter set per pass, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; SQ counters in their own).  The program itself follows `--` (python3 bench.py ...), no wrapper.
set -e
OUT="$PWD/gpurun_out/${1:-prof_mmr}"
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="$PWD/bench.py"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace/runc" -- python3 "$BENCH" --steps 5 --warmup 2 --no-cpu-baseline --no-prove > "$OUT/trace.log" 2>&1
echo "trace done" >> "$OUT/progress.log"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d "$OUT/pmc_sq/runc" -- python3 "$BENCH" --steps 2 --warmup 1 --no-cpu-baseline --no-prove > "$OUT/pmc_sq.log" 2>&1
echo "sq done" >> "$OUT/progress.log"
# VALU-busy: numerator (SQ_ACTIVE_INST_VALU, quad-cycles summed over the SIMDs) and denominator (GRBM_GUI_ACTIVE, cycles summed over the
# 8 XCDs) from the SAME dispatches (GRBM slots are independent of the SQ ones)
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_grbm/runc" -- python3 "$BENCH" --steps 2 --warmup 1 --no-cpu-baseline --no-prove > "$OUT/pmc_grbm.log" 2>&1
echo "grbm done" >> "$OUT/progress.log"
# the matrix pipe (the dense MDS layers run as v_mfma_i32_32x32x32_i8 since the end of round 3): instructions, busy and co-execution cycles
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma/runc" -- python3 "$BENCH" --steps 2 --warmup 1 --no-cpu-baseline --no-prove > "$OUT/pmc_mfma.log" 2>&1
echo "mfma done" >> "$OUT/progress.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch/runc" -- python3 "$BENCH" --steps 2 --warmup 1 --no-cpu-baseline --no-prove > "$OUT/pmc_fetch.log" 2>&1
echo "fetch done" >> "$OUT/progress.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write/runc" -- python3 "$BENCH" --steps 2 --warmup 1 --no-cpu-baseline --no-prove > "$OUT/pmc_write.log" 2>&1
echo "write done" >> "$OUT/progress.log"
cd - > /dev/null
python3 tools/summarize_rocprof.py "$OUT" 3 > "$OUT/summary.txt"
grep '^{' "$OUT/trace.log" | tail -1 > "$OUT/bench_line.json" || true
