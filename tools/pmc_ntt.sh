# HBM traffic of the two passes of the 2^20-point transform (FETCH_SIZE / WRITE_SIZE in separate passes; counter unit KB, raw)
export TMPDIR=/tmp; R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_ntt_$c -- python3 $R/tools/ntt_lde_probe.py --what ntt,lde --reps 3 > $R/gpurun_out/pmc_ntt_$c.log 2>&1 || exit 1
done
cd $R; python3 - <<PY
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/pmc_ntt_%s/**/*counter_collection.csv"%c,recursive=True)[0]
    acc=collections.defaultdict(lambda:[0,0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]==c:
            k=r["Kernel_Name"].replace("void (anonymous namespace)::","")[:48]; acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
    for k,v in sorted(acc.items()):
        if "ntt20_pass" in k or "lde12" in k: print(c, k, "calls", v[1], "per dispatch MB (raw KB/1000)", round(v[0]/v[1]/1000,1))
PY
