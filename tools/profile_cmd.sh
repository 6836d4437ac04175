#!/bin/bash
# rocprofv3 evidence for any python command of this repo (run ON the GPU box, from the repo root, through gpurun):
#   tools/profile_cmd.sh <out_dir_under_gpurun_out> <script.py> [args...]
# kernel trace + PMC passes, one counter set per pass as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE
# do not fit one pass; GRBM slots are independent of the SQ ones).  The program itself follows `--` (python3 script ...), no wrapper.
set -e
OUT="$PWD/gpurun_out/$1"
shift
SCRIPT="$PWD/$1"
shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace/runc" -- python3 "$SCRIPT" "$@" > "$OUT/trace.log" 2>&1
echo "trace done" >> "$OUT/progress.log"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq/runc" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_sq.log" 2>&1
echo "sq done" >> "$OUT/progress.log"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_lds/runc" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_lds.log" 2>&1
echo "lds done" >> "$OUT/progress.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch/runc" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_fetch.log" 2>&1
echo "fetch done" >> "$OUT/progress.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write/runc" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_write.log" 2>&1
echo "write done" >> "$OUT/progress.log"
cd - > /dev/null
python3 tools/summarize_rocprof.py "$OUT" 1 > "$OUT/summary.txt"
grep '^{' "$OUT/trace.log" > "$OUT/probe_lines.json" || true
