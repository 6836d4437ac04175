#!/bin/bash
# per-kernel times of the 2^20-point transform for the tile-width modes (P2MT_LDE12 = 1 default / 3 wide / 4 narrow)
cd /tmp && export TMPDIR=/tmp
for m in 1 3 4; do
  export P2MT_LDE12=$m
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ntt_mode$m -o ntt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ntt_lde_probe.py --what ntt --reps 10 > $GRAFT_REPO_ROOT/gpurun_out/ntt_mode$m.log 2>&1 || exit 1
  echo "mode $m"; grep ms_median $GRAFT_REPO_ROOT/gpurun_out/ntt_mode$m.log | cut -c1-200
  grep k_ntt20_pass $GRAFT_REPO_ROOT/gpurun_out/ntt_mode$m/*kernel_stats.csv | cut -c1-260
done
