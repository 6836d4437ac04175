#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
run() { # name, env..., args
  name=$1; shift
  env_kv=$1; shift
  ( export $env_kv; rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/diag_$name -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prove "$@" > $ROOT/gpurun_out/diag_$name.log 2>&1 )
  f=$(find $ROOT/gpurun_out/diag_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name ($env_kv $@)"; grep "k_mmr_level<\|k_mmr_subtree" $f | awk -F'","|",|,"' '{print substr($1,1,60)}' > /dev/null
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_mmr_level<' in r['Name'] or 'k_mmr_subtree' in r['Name'] or 'quad' in r['Name']:
        print('   %-40s calls=%s avg=%.1f min=%.1f max=%.1f us' % (r['Name'][:40].replace('void (anonymous namespace)::',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
  rm -rf $ROOT/gpurun_out/diag_$name
}
run default24 X=1
run sub5_24 P2MT_SUBTREE=5
run default23 X=1 --log-leaves 23
run default22 X=1 --log-leaves 22
