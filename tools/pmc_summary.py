"""Summaries of one tools/collect_profiles.sh run:  python3 tools/pmc_summary.py <out_dir> <tag>

Writes <tag>_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE per launch of every profiled region) and <tag>_pmc_summary.json:
per region the instruction counts and a MEASURED issue roofline

  valu_issue_util = SQ_INSTS_VALU x t_issue / (1024 SIMDs x kernel time)     t_issue: tools/microbench/valu_rate on this box
                    (gfx950 is SIMD-32: a wave64 vector instruction issues over 2 cycles, MI355X_MICROARCH.md "Wave scheduling")
  salu_util       = SQ_INSTS_SALU / (256 scalar units x kernel clocks)        clocks = GRBM_GUI_ACTIVE / 8 XCDs
  wait_any, wait_inst_any, active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint shares of a
                    resident wave's life: parked at s_waitcnt / barrier, stalled at issue, issuing)
  lds_bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE

  cu_issue_util   = all wave-instructions (VALU + SALU + LDS + SMEM + VMEM) x t_issue / (1024 SIMDs x kernel time)

`bound` is derived from those: "valu-issue" / "salu-issue" when that pipe is >= 60 % busy, "issue" when all instructions together are, otherwise the largest of the three
wave-cycle shares ("latency" = parked waves, "dependency" = issue stalls, "issue" = issuing but no single pipe saturated)."""
import collections
import csv
import json
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
# profiled regions of the library (navgpu_kernel_name) <- the device kernels they launch
GROUPS = {"k_obstacle": ("k_obstacle",), "k_merge": ("k_merge",), "k_inflate": ("k_inflate", "k_inflate_bits", "k_inflate_pq"),
          "k_bfs": ("k_bfs", "k_bfs_rows", "k_bfs_rows2", "k_bfs_wave", "k_bfs_global", "k_free_bits"),
          "k_score": ("k_score_sweep", "k_score_tab", "k_score_gen", "k_score_prep_tab", "k_score_prep_gen"), "k_select": ("k_select",),
          "k_samples": ("k_samples",)}
MAIN = {"k_score": ("k_score_sweep", "k_score_tab", "k_score_gen"), "k_bfs": ("k_bfs", "k_bfs_rows", "k_bfs_rows2", "k_bfs_wave", "k_bfs_global")}
N_SIMD, N_CU, N_XCD = 1024, 256, 8


def kname(s):
    return s.split("(")[0].replace("void ", "").replace("navgpu::", "").split("<")[0]


def read_pmc(path, acc):
    if not os.path.exists(path):
        return
    for r in csv.DictReader(open(path)):
        acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))


# ---- HBM traffic
raw = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("fetch_size", "write_size"):
    read_pmc(f"{out}/{tag}_pmc_{c}.csv", raw)
kern = {}
for g, members in GROUPS.items():
    launches = max((len(raw[m]["WRITE_SIZE"]) for m in MAIN.get(g, members) if m in raw), default=0)
    if not launches:
        continue
    f = sum(sum(raw[m]["FETCH_SIZE"]) for m in members if m in raw) * 1024.0 / launches  # counters are in KB
    w = sum(sum(raw[m]["WRITE_SIZE"]) for m in members if m in raw) * 1024.0 / launches
    kern[g] = dict(fetch_bytes_raw=f, write_bytes=w, hbm_bytes_raw=f + w, hbm_bytes_fetch_x2=2 * f + w, launches=launches,
                   device_kernels=[m for m in members if m in raw])
json.dump(dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean per launch of the profiled region, KB x 1024",
               caveat="gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x and is uncalibrated for gathers",
               kernels=kern), open(f"{out}/{tag}_hbm_traffic.json", "w"), indent=1)

# ---- kernel times (the --stats pass) and the VALU issue time of this box
avg_ns = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f"{out}/{tag}_kernel_stats.csv")):
    k = kname(r["Name"])
    avg_ns[k][0] += float(r["TotalDurationNs"])
    avg_ns[k][1] += int(r["Calls"])
t_issue_ns, t_src = 1.02, "profiles/round2_valu_rate_microbench.txt (no microbench output in this run)"
try:
    txt = open(f"{out}/{tag}_valu_rate_microbench.txt").read()
    t_issue_ns = float(re.search(r"([0-9.]+) ns per wave-instruction per SIMD", txt).group(1))
    t_src = f"{tag}_valu_rate_microbench.txt (v_fma_f32, 8 waves per SIMD, same box)"
except Exception:
    pass

# ---- instruction mix and waits per region: mean per launch, summed over the 8 XCDs
mix = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ("sq_insts_valu", "sq_active_inst_valu", "sq_wait_any"):
    read_pmc(f"{out}/{tag}_pmc_{f}.csv", mix)
regions = {}
for g, members in GROUPS.items():
    main = [m for m in MAIN.get(g, members) if m in mix and mix[m].get("SQ_INSTS_VALU")]
    if not main:
        continue
    launches = max(len(mix[m]["SQ_INSTS_VALU"]) for m in main)

    def tot(c, per=None):
        # mean per launch of the region; every pass has its own launch count
        n = max((len(mix[m][c]) for m in main), default=0)
        return sum(sum(mix[m][c]) for m in members if m in mix) / n if n else None

    clk = (tot("GRBM_GUI_ACTIVE") or 0.0) / N_XCD
    t_ns = sum(avg_ns[m][0] for m in members if m in avg_ns) / max(max((avg_ns[m][1] for m in main if m in avg_ns), default=0), 1)
    v = dict(kernel_time_ns=t_ns, insts_valu=tot("SQ_INSTS_VALU"), insts_salu=tot("SQ_INSTS_SALU"), insts_lds=tot("SQ_INSTS_LDS"),
             insts_smem=tot("SQ_INSTS_SMEM"), insts_vmem_rd=tot("SQ_INSTS_VMEM_RD"), insts_vmem_wr=tot("SQ_INSTS_VMEM_WR"),
             waves=tot("SQ_WAVES"), gpu_clocks=clk, effective_clock_GHz=(clk / t_ns if t_ns else None), launches=launches,
             device_kernels=[m for m in members if m in mix])
    v["valu_issue_util"] = v["insts_valu"] * t_issue_ns / (N_SIMD * t_ns) if t_ns else None
    v["salu_util"] = v["insts_salu"] / (N_CU * clk) if clk and v["insts_salu"] is not None else None
    # every wave-instruction of whatever kind against the rate a SIMD issues at (round 4: a CU retires about as many instructions per
    # clock whatever their mix - tools/microbench/valu_rate's scalar and mixed rows - and k_score_sweep takes the same time compiled
    # for 4 or 6 waves per SIMD: what bounds it is the NUMBER of instructions)
    n_all = sum(v[k] or 0.0 for k in ("insts_valu", "insts_salu", "insts_lds", "insts_smem", "insts_vmem_rd", "insts_vmem_wr"))
    v["insts_all"] = n_all
    v["cu_issue_util"] = n_all * t_issue_ns / (N_SIMD * t_ns) if t_ns else None
    wc = tot("SQ_WAVE_CYCLES")
    for name, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY"),
                    ("wait_inst_lds", "SQ_WAIT_INST_LDS")):
        x = tot(c)
        v[name] = x / wc if (x is not None and wc) else None
    lb, li = tot("SQ_LDS_BANK_CONFLICT"), tot("SQ_LDS_IDX_ACTIVE")
    v["lds_bank_conflict"] = lb / li if (lb is not None and li) else None
    v["wave_cycles_quadcycles"] = wc
    if (v["valu_issue_util"] or 0) >= 0.6:
        v["bound"] = "valu-issue"
    elif (v["cu_issue_util"] or 0) >= 0.6:
        v["bound"] = "issue"
    elif (v["salu_util"] or 0) >= 0.6:
        v["bound"] = "salu-issue"
    else:
        shares = {"latency": v["wait_any"] or 0, "dependency": v["wait_inst_any"] or 0, "issue": v["active"] or 0}
        v["bound"] = max(shares, key=shares.get)
    regions[g] = v
json.dump(dict(source="rocprofv3 --pmc (three SQ counter passes, separate from the TCC ones and from --stats); mean per launch, summed over XCDs",
               t_issue_ns=t_issue_ns, t_issue_source=t_src,
               valu_issue_util="SQ_INSTS_VALU x t_issue / (1024 SIMDs x kernel time from the --stats pass)",
               salu_util="SQ_INSTS_SALU / (256 CUs x GRBM_GUI_ACTIVE / 8)",
               cu_issue_util="(SQ_INSTS_VALU + SALU + LDS + SMEM + VMEM_RD + VMEM_WR) x t_issue / (1024 SIMDs x kernel time)",
               wave_cycle_shares="SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots)",
               regions=regions), open(f"{out}/{tag}_pmc_summary.json", "w"), indent=1)
rnd = lambda x: None if x is None else round(x, 3)
for k, v in regions.items():
    print(k, dict(ms=rnd(v["kernel_time_ns"] / 1e6), valu_M=rnd(v["insts_valu"] / 1e6), salu_M=rnd((v["insts_salu"] or 0) / 1e6),
                  valu_issue_util=rnd(v["valu_issue_util"]), salu_util=rnd(v["salu_util"]), cu_issue_util=rnd(v["cu_issue_util"]), wait_any=rnd(v["wait_any"]),
                  wait_inst_any=rnd(v["wait_inst_any"]), active=rnd(v["active"]), lds_conflict=rnd(v["lds_bank_conflict"]), bound=v["bound"]))
print({k: (round(v["hbm_bytes_fetch_x2"] / 1e6, 1), "MB") for k, v in kern.items()})
