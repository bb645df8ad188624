#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box:  bash tools/collect_profiles.sh round1_c
# (kernel-trace/stats pass and the two PMC passes are separate runs; outputs land in gpurun_out/profiles_<tag>/)
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1; out=gpurun_out/profiles_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single > $out/${tag}_bench_under_rocprof.json 2> $out/stats.err
cp $out/stats/r_kernel_stats.csv $out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single > $out/$c.json 2> $out/$c.err
  lc=$(echo $c | tr A-Z a-z)
  cp $out/$c/r_counter_collection.csv $out/${tag}_pmc_$lc.csv
done
python3 - "$out" "$tag" <<'PY'
import csv, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
# profiled regions of the library (navgpu_kernel_name) <- the device kernels they launch
GROUPS = {"k_obstacle": ("k_obstacle",), "k_merge": ("k_merge",), "k_inflate": ("k_inflate", "k_inflate_bits"),
          "k_bfs": ("k_bfs", "k_bfs_wave", "k_bfs_global", "k_free_bits"), "k_score": ("k_score_tab", "k_score_gen", "k_score_prep_tab", "k_score_prep_gen"),
          "k_select": ("k_select",)}
MAIN = {"k_score": ("k_score_tab", "k_score_gen"), "k_bfs": ("k_bfs", "k_bfs_wave", "k_bfs_global")}
raw = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{out}/{tag}_pmc_{c.lower()}.csv")):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("navgpu::", "").split("<")[0]
        raw[k][c].append(float(r["Counter_Value"]) * 1024.0)  # counters are in KB
kern = {}
for g, members in GROUPS.items():
    launches = max((len(raw[m]["WRITE_SIZE"]) for m in MAIN.get(g, members) if m in raw), default=0)
    if not launches: continue
    f = sum(sum(raw[m]["FETCH_SIZE"]) for m in members if m in raw) / launches
    w = sum(sum(raw[m]["WRITE_SIZE"]) for m in members if m in raw) / launches
    kern[g] = dict(fetch_bytes_raw=f, write_bytes=w, hbm_bytes_raw=f + w, hbm_bytes_fetch_x2=2 * f + w, launches=launches,
                   device_kernels=[m for m in members if m in raw])
json.dump(dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean per launch of the profiled region, KB x 1024",
               caveat="gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x and is uncalibrated for gathers",
               kernels=kern), open(f"{out}/{tag}_hbm_traffic.json", "w"), indent=1)
print(json.dumps(kern, indent=1))
PY
