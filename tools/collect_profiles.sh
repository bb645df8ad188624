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
# instruction mix (two more PMC passes; SQ counters only): VALU / SALU wave-instructions and the busy clock
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/$n -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single > $out/$n.json 2> $out/$n.err
  cp $out/$n/r_counter_collection.csv $out/${tag}_pmc_$(echo $n | tr A-Z a-z).csv
done
python3 - "$out" "$tag" <<'PY'
import csv, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
# profiled regions of the library (navgpu_kernel_name) <- the device kernels they launch
GROUPS = {"k_obstacle": ("k_obstacle",), "k_merge": ("k_merge",), "k_inflate": ("k_inflate", "k_inflate_bits"),
          "k_bfs": ("k_bfs", "k_bfs_wave", "k_bfs_big", "k_bfs_global", "k_free_bits"), "k_score": ("k_score_tab", "k_score_gen", "k_score_prep_tab", "k_score_prep_gen"),
          "k_select": ("k_select",)}
MAIN = {"k_score": ("k_score_tab", "k_score_gen"), "k_bfs": ("k_bfs", "k_bfs_wave", "k_bfs_big", "k_bfs_global")}
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
# ---- instruction mix per profiled region: mean per launch, summed over the 8 XCDs
mix = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ("sq_insts_valu", "sq_active_inst_valu"):
    for r in csv.DictReader(open(f"{out}/{tag}_pmc_{f}.csv")):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("navgpu::", "").split("<")[0]
        mix[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
regions = {}
for g, members in GROUPS.items():
    main = [m for m in MAIN.get(g, members) if m in mix]
    if not main: continue
    launches = max(len(mix[m]["SQ_INSTS_VALU"]) for m in main)
    tot = lambda c: sum(sum(mix[m][c]) for m in members if m in mix) / max(launches, 1)
    clk = tot("GRBM_GUI_ACTIVE") / 8.0  # rocprofv3 sums the 8 XCDs
    v = dict(insts_valu=tot("SQ_INSTS_VALU"), insts_salu=tot("SQ_INSTS_SALU"), insts_lds=tot("SQ_INSTS_LDS"), insts_smem=tot("SQ_INSTS_SMEM"),
             insts_vmem_rd=tot("SQ_INSTS_VMEM_RD"), insts_vmem_wr=tot("SQ_INSTS_VMEM_WR"), active_inst_valu_quadcycles=tot("SQ_ACTIVE_INST_VALU"),
             active_inst_sca_quadcycles=tot("SQ_ACTIVE_INST_SCA"), wave_cycles_quadcycles=tot("SQ_WAVE_CYCLES"), waves=tot("SQ_WAVES"),
             gpu_clocks=clk, launches=launches, device_kernels=[m for m in members if m in mix])
    # a SIMD holds one VALU instruction at a time: busy fraction = VALU-active cycles / (1024 SIMDs x kernel clocks)
    v["valu_busy"] = 4.0 * v["active_inst_valu_quadcycles"] / (1024.0 * clk) if clk else None
    v["bound"] = "valu" if (v["valu_busy"] or 0) >= 0.5 else "latency"
    regions[g] = v
json.dump(dict(source="rocprofv3 --pmc (SQ instruction counters, separate passes from the TCC ones); mean per launch, summed over XCDs",
               valu_busy="4 x SQ_ACTIVE_INST_VALU (quad-cycles -> cycles) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)", regions=regions),
          open(f"{out}/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: (round(v["valu_busy"], 3), round(v["insts_valu"] / 1e6, 1), round(v["insts_salu"] / 1e6, 1)) for k, v in regions.items()}))
PY
