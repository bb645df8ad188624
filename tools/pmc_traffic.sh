#!/bin/bash
# HBM-side traffic per launch of every kernel of the step (FETCH_SIZE and WRITE_SIZE, one rocprofv3 --pmc pass each, one-stream
# bench run of the CURRENT build):  bash tools/pmc_traffic.sh [name]      -> gpurun_out/traffic_<name>/summary.txt
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
name=${1:-now}; out=gpurun_out/traffic_$name; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o r -- python3 bench.py --no-cpu-baseline --no-single --groups 1 --steps 3 --warmup 1 > $out/$c.json 2> $out/$c.err
done
python3 - <<PY | tee $out/summary.txt
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for r in csv.DictReader(open("$out/%s/r_counter_collection.csv" % c)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("navgpu::","").split("<")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
# FETCH_SIZE / WRITE_SIZE count kilobytes; on gfx950 FETCH_SIZE under-counts by 2 (MI355X_MICROARCH.md, HBM / rocprofv3 section)
for k,v in sorted(acc.items()):
    f=sum(v["FETCH_SIZE"])/max(1,len(v["FETCH_SIZE"])); w=sum(v["WRITE_SIZE"])/max(1,len(v["WRITE_SIZE"]))
    print("$name %-22s fetch x2 %8.2f MB  write %8.2f MB per launch (%d launches)" % (k, 2*f*1024/1e6, w*1024/1e6, len(v["FETCH_SIZE"])))
PY
