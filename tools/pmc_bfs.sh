#!/bin/bash
# PMC passes over k_bfs* (run on the GPU box): tools/pmc_bfs.sh <tag> [env assignments are inherited]
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1
mkdir -p gpurun_out/pmc_$tag
for grp in "SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_$tag/$n -o r --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single > gpurun_out/pmc_$tag/$n.log 2>&1 || { tail -5 gpurun_out/pmc_$tag/$n.log; }
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_$tag/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "bfs" not in k and "score" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
