import json, sys
d=json.load(open(sys.argv[1]))
print("value %.4g ms_per_step %.4f" % (d["value"], d["ms_per_step"]))
print("kernel_ms", d["kernel_ms"])
r2=d.get("roofline_second")
if r2: print("roofline_second", {k:r2[k] for k in ("kernel","achieved","frac","avg_launch_ms","bound","frac_issue") if k in r2})
r=d["roofline"]; print("roofline (alone)", {k:r[k] for k in ("kernel","achieved","frac","avg_launch_ms","bound","frac_issue") if k in r}, "in_schedule", {k: r["in_schedule"][k] for k in ("achieved","frac","avg_launch_ms")}, "profile", {k: r.get("profile",{}).get(k) for k in ("tag","matches_running_build")})
for k in ("pcie_inclusive","resident_inputs","value_over_resident","cycle_latency","cycle_latency_resident_inputs","one_stream","inflation_cells_per_s"):
    v=d.get(k)
    if isinstance(v, dict): v={a:b for a,b in v.items() if a!="note"}
    print(k, v)
for k in ("whole_grid_wavefronts","inflation_reference_order","single_robot","configs4_one_gpu_share"):
    v=d.get(k)
    if v: print(k, {a:b for a,b in v.items() if a in ("ms_per_step","k_bfs_ms","ms_per_update_per_cycle_windows","ms_per_cycle","kernel_ms")})
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("gpu_over_cpu"))
