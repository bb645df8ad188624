#!/bin/bash
# Instruction counts per launch of the scoring kernel for build variants:  bash tools/pmc_quick.sh "name:EXTRA flags" ...
# (one rocprofv3 --pmc pass per variant on a one-stream bench run; ends on the default build)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in "$@"; do
  name="${v%%:*}"; extra="${v#*:}"
  make -s -C navigation_amd/csrc clean >/dev/null; make -s -j8 -C navigation_amd/csrc EXTRA="$extra" 2>&1 | grep -E "error|Stop"
  out=gpurun_out/pmcq_$name; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $out -o r -- python3 bench.py --no-cpu-baseline --no-single --groups 1 --steps 3 --warmup 1 > $out/bench.json 2> $out/err.txt
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$out/r_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("navgpu::","").split("<")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("k_score_sweep","k_score_tab","k_score_prep_tab","k_bfs_rows","k_bfs_rows2","k_inflate_bits","k_merge"):
    if k in acc:
        print("$name", k, {c: round(sum(v)/len(v)/1e6,2) for c,v in acc[k].items()}, "M per launch", flush=True)
PY
done
make -s -C navigation_amd/csrc clean >/dev/null; make -s -j8 -C navigation_amd/csrc 2>&1 | grep -E "error|Stop"
echo "default build restored"
