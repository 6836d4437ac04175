export TMPDIR=/tmp; R=$PWD; TAG=$1
cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_f/runc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prove > $R/gpurun_out/${TAG}_f.log 2>&1
cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_w/runc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prove > $R/gpurun_out/${TAG}_w.log 2>&1
cd $R; python3 - <<PY
import csv,glob,collections
for d,c in (("${TAG}_f","FETCH_SIZE"),("${TAG}_w","WRITE_SIZE")):
    f=glob.glob("gpurun_out/%s/runc/**/*counter_collection.csv"%d,recursive=True)[0]
    acc=collections.defaultdict(lambda:[0,0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]==c:
            k=r["Kernel_Name"][:60]; acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
    for k,v in acc.items():
        if "subtree<4" in k: print(c, k, "per dispatch MB", v[0]/v[1]/1000)
PY
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prove 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['launch_ms'])"; done
