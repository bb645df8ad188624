#!/bin/bash
# Instruction-cache behaviour of the step's kernels, one stream (--groups 1) and the default stream groups:
#   bash tools/pmc_icache.sh      (on the GPU box; writes gpurun_out/icache/*.txt)
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/icache; mkdir -p $out
for g in 1 4; do
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/g$g -o r -- \
    python3 bench.py --no-cpu-baseline --no-single --groups $g --steps 5 --warmup 2 > $out/g$g.json 2> $out/g$g.err || { tail -3 $out/g$g.err; continue; }
  python3 - "$out/g$g/r_counter_collection.csv" "$g" <<'PY' | tee $out/groups$g.txt
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("navgpu::", "").split("<")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQC_ICACHE_REQ":
        n[k] += 1
print("groups", sys.argv[2])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQC_ICACHE_REQ", 0))[:9]:
    req = max(v.get("SQC_ICACHE_REQ", 0), 1)
    print("%-18s launches %4d  req/launch %10.0f  miss %.4f  dup-miss %.4f" % (k, n[k], req / max(n[k], 1), v.get("SQC_ICACHE_MISSES", 0) / req, v.get("SQC_ICACHE_MISSES_DUPLICATE", 0) / req))
PY
done
