#!/bin/bash
# Build time of a 2^L-leaf MMR for L = 18..24 with the shipped kernels (run ON the GPU box from the repo root): one line per size from
# bench.py's JSON (ms_per_step, median, stage-1 launch, stage-1 kernel, G hashes/s).  DESIGN.md 6 predicts strong scaling from these.
for L in ${@:-18 19 20 21 22 23 24}; do
  python3 bench.py --log-leaves $L --steps 20 --warmup 3 --no-cpu-baseline --no-prove 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        print('log_leaves %d ms_per_step %.4f median %.4f stage1_ms %.4f %s Ghash/s %.4f' % ($L, d['ms_per_step'], d['ms_per_step_median'], d['roofline']['launch_ms'], d['config']['stage1'], d['value']/1e9))
"
done
