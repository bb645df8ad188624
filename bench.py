#!/usr/bin/env python3
"""bench.py — fleet throughput of the MI355X-native costmap + DWA hot path.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 launched by torch.distributed.run,
one rank per GPU).  One *step* = one pass of the hot path over one batch: for every robot
instance of the rank's fleet a LayeredCostmap::updateMap (LaserScan clearing + marking + merge +
inflation) followed by a DWAPlanner::findBestPath (3 MapGrid wavefronts + rollout + six critics +
selection).  Inputs are staged in HBM before the timed region.

Workload (config.workload): BASELINE.json's metric is a whole-node throughput, quoted on the
fleet configurations; configs[3] (2048 instances over 8 GPUs) does not fit one GPU, so at N=1 the
workload is the largest single-GPU configuration, configs[2]: 256 batched robot instances on one
MI355X, 400x400 costmaps, 32x32x16 samples, 20 sim steps, LaserScan update each cycle.  At N GPUs
every rank runs the same 256 instances (weak scaling; N=8 is configs[3]).  The single-robot
configs[1] latency is reported beside it in "single_robot".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
T_STEPS, P_PERIM, G_GRIDS = 20, 32, 4
BYTES_PER_TRAJ = 20 + T_STEPS * (P_PERIM + 1) + T_STEPS * G_GRIDS * 4  # = 1000 B
BYTES_PER_BFS_CELL = 5       # 1 B costmap read + 4 B distance write, per grid
BYTES_PER_INFL_CELL = 2
BYTES_PER_MERGE_CELL = 3


def build_fleet(nav, n_inst, n_cells, seed0, device=0, vs=(32, 32, 16), footprint="square"):
    from navigation_amd import _lib as N, synth
    fp = synth.FOOTPRINT5 if footprint == "poly5" else synth.FOOTPRINT
    fl = nav.Fleet(n_inst, n_cells, n_cells, synth.RES, layers=N.LAYER_STATIC | N.LAYER_OBSTACLE | N.LAYER_INFLATION,
                   max_points=720, max_observations=1, max_plan=200, max_footprint=8, max_sim_steps=24, device=device)
    fl.configure_obstacle()
    fl.set_footprint(fp)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(fp))
    cfg = synth.fleet_config(*vs)
    fl.configure_planner(cfg)
    insts = [synth.make_instance(n_cells, seed0 + i) for i in range(n_inst)]
    for i, ins in enumerate(insts):
        fl.add_static_map(np.where(ins["cells"] == 254, 100, 0).astype(np.int8), first=i, count=1)
    poses = np.array([[float(v) for v in ins["pos"]] for ins in insts])
    obs = []
    for i, ins in enumerate(insts):
        pts = synth.laser_scan(ins, 0)
        obs.append(dict(instance=i, points=pts, origin=(float(ins["pos"][0]), float(ins["pos"][1]), 0.3),
                        obstacle_range=2.5, raytrace_range=3.0))
    fl.stage_observations(poses, obs)
    fl.stage_planner(np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]),
                     np.stack([i["plan"] for i in insts]))
    fl.set_plan()
    fl._bench_host_inputs = (poses, obs, np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]),
                             np.stack([i["plan"] for i in insts]))
    return fl, insts, cfg


def raw_inputs(fl):
    """Pre-marshalled ctypes/numpy buffers of one cycle's inputs (so the PCIe-inclusive leg times the
    C-ABI staging calls, not Python list handling)."""
    from navigation_amd._lib import Observation, RobotState, OBS_MARKING, OBS_CLEARING
    poses, obs, pos, vel, plans = fl._bench_host_inputs
    arr = (Observation * len(obs))()
    pts, off = [], 0
    for k, o in enumerate(obs):
        p = np.ascontiguousarray(o["points"], np.float32)
        arr[k] = Observation(o["instance"], off, len(p), OBS_MARKING | OBS_CLEARING, o["origin"][0], o["origin"][1],
                             o["origin"][2], o["obstacle_range"], o["raytrace_range"])
        pts.append(p)
        off += len(p)
    allp = np.ascontiguousarray(np.concatenate(pts), np.float32)
    n = len(pos)
    states = (RobotState * n)()
    k = plans.shape[1]
    for i in range(n):
        states[i].pos[:] = [float(v) for v in pos[i]]
        states[i].vel[:] = [float(v) for v in vel[i]]
        states[i].plan_first = i * k
        states[i].plan_count = k
    packed = np.ascontiguousarray(plans, np.float64).reshape(-1, 2)
    return poses, arr, len(obs), allp, states, n, packed


def hbm_traffic_from_profiles(kernel):
    """HBM bytes per launch of `kernel` as measured by rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this
    exact workload (bench.py cannot collect PMC counters itself); newest profiles/*_hbm_traffic.json."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        # MI355X_MICROARCH.md (HBM): gfx950 FETCH_SIZE tallies 128-B read requests at 64 B -> doubled; WRITE_SIZE is exact
        return d["kernels"][kernel]["hbm_bytes_fetch_x2"], os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def legacy_leg(nav, insts, n_cells, masters, with_cpu, device):
    """TrajectoryPlanner::findBestPath (BaseLocalPlanner.cfg defaults: 20 x 20 samples, two wavefronts)
    for a fleet of 32 robots; PCIe- and host-selection-inclusive by construction.  CPU: the oracle, one thread."""
    from navigation_amd import _lib as N, synth
    ns = min(len(masters), 32)
    cfg = N.TpConfig()
    fl = nav.Fleet(ns, n_cells, n_cells, synth.RES, layers=N.LAYER_OBSTACLE, max_sim_steps=96, max_plan=256, device=device)
    fl.set_footprint(synth.FOOTPRINT)
    fl.upload(N.GRID_MASTER, np.ascontiguousarray(masters[:ns]))
    fl.configure_trajectory_planner(cfg)
    for k in range(ns):
        fl.tp_update_plan(k, insts[k]["plan"])
    pos = np.stack([i["pos"] for i in insts[:ns]]).astype(np.float32)
    vel = np.stack([i["vel"] for i in insts[:ns]]).astype(np.float32)
    r = fl.tp_find_best_path(pos, vel)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fl.tp_find_best_path(pos, vel)
    dt = time.perf_counter() - t0
    calls = sum(x.n_samples for x in r)
    out = {"workload": f"{ns} robots, {n_cells}x{n_cells}, vx_samples 20 x vtheta_samples 20, sim_time 1.7 / 0.025",
           "ms_per_cycle": dt / reps * 1e3, "trajectories_per_s": calls * reps / dt, "generate_calls_per_cycle": calls}
    fl.close()
    if with_cpu:
        from oracle import pyoracle as orc
        k_cpu = min(ns, 4)
        oracles = [orc.TrajectoryPlanner(masters[k], synth.RES, cfg, synth.FOOTPRINT) for k in range(k_cpu)]
        for k, o in enumerate(oracles):
            o.update_plan(insts[k]["plan"])
        t0 = time.perf_counter()
        n_calls = 0
        for k, o in enumerate(oracles):
            res, _, _ = o.find_best_path(pos[k], vel[k], N.TpResult, N.TpSample)
            n_calls += res.n_samples
        dc = time.perf_counter() - t0
        out["cpu_port"] = {"trajectories_per_s": n_calls / dc, "cores": 1, "sample": f"{k_cpu} robots x 1 cycle"}
    return out


def configs4_leg(nav, device, n_robots=64, n_cells=1000, steps=10):
    """BASELINE configs[4] on one GPU: 1000 x 1000 costmaps, voxel_grid obstacle layer (10 z-voxels) + inflation, the
    5-vertex polygon footprint, 64 x 64 x 32 samples, 20 steps; scans and plans resident.  A supplementary figure (the
    headline is configs[2]); 64 robots keep the leg to a few seconds, the rate per robot does not depend on the count."""
    from navigation_amd import _lib as N, synth
    fl = nav.Fleet(n_robots, n_cells, n_cells, synth.RES, layers=N.LAYER_VOXEL | N.LAYER_INFLATION, track_unknown=False, max_points=1440,
                   max_observations=1, max_sim_steps=24, max_plan=256, max_footprint=8, device=device)
    fl.configure_obstacle(z_voxels=10, origin_z=0.0, z_resolution=0.2, unknown_threshold=15, mark_threshold=0, max_obstacle_height=2.0)
    fl.set_footprint(synth.FOOTPRINT5)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(synth.FOOTPRINT5))
    fl.configure_planner(nav.DwaConfig(vx_samples=64, vy_samples=64, vth_samples=32, sim_time=2.0, sim_granularity=0.1, discretize_by_time=1))
    insts = [synth.make_instance(n_cells, 700 + i) for i in range(n_robots)]
    fl.upload(N.GRID_MASTER, np.stack([i["cells"] for i in insts]))
    poses = np.array([[float(v) for v in i["pos"]] for i in insts])
    obs = [dict(instance=k, points=synth.laser_scan(i, 0, z=0.3, z_jitter=1.5), origin=(poses[k][0], poses[k][1], 0.3), obstacle_range=2.5,
                raytrace_range=3.0) for k, i in enumerate(insts)]
    fl.stage_observations(poses, obs)
    fl.stage_planner(np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]), np.stack([i["plan"] for i in insts]))
    fl.set_plan()
    for _ in range(2):
        step(fl)
    fl.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(fl)
    fl.sync()
    dt = (time.perf_counter() - t0) / steps
    fl.profile(True)
    fl.profile_reset()
    for _ in range(3):
        step(fl)
    fl.sync()
    pr = fl.profile_read()
    fl.profile(False)
    scored = sum(r.n_scored for r in fl.results())
    fl.close()
    return {"workload": f"{n_robots} robots, {n_cells}x{n_cells}, voxel layer (10 z-voxels) + inflation, 5-vertex footprint, 64x64x32 samples, 20 steps",
            "ms_per_step": dt * 1e3, "trajectories_per_s": scored / dt, "per_instance_trajectories_per_s": scored / dt / n_robots,
            "kernel_ms": {k: round(v[0] / v[1], 4) for k, v in pr.items() if v[1]}}


def step(fl, poses=None, k=0):
    """One pass of the hot path over the fleet.  With `poses` (a PoseSchedule) the cycle first stages its own pose and
    velocity (24 B per robot, navgpu_planner_stage_poses) - everything else (costmaps, scans, plans) is resident in HBM."""
    if poses is not None:
        fl.stage_poses(poses.pos[k % len(poses.pos)], poses.vel)
    fl.update_map()
    fl.planner_cycle()


class PoseSchedule:
    """Seeded per-cycle poses: the robot of cycle k stands at base + N(0, 2 cm) and is turned by N(0, 0.05 rad), so that
    successive cycles differ (wavefront lengths, reach boxes, which samples collide) and last cycle's level counts are
    no perfect predictor for the longest-first dispatch of this one."""

    def __init__(self, pos, vel, n_cycles, seed):
        rs = np.random.RandomState(seed)
        d = rs.normal(size=(n_cycles,) + pos.shape) * np.array([0.02, 0.02, 0.05])
        self.pos = [np.ascontiguousarray(pos + d[k], np.float32) for k in range(n_cycles)]
        self.vel = np.ascontiguousarray(vel, np.float32)


def host_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256
    logical CPUs and a quota of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    eff = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return eff, n, quota


def pmc_summary_from_profiles():
    """Newest profiles/*_pmc_summary.json (tools/collect_profiles.sh): VALU-busy fraction etc. per device kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a fresh child process tree BEFORE anything
    in this process touches torch.cuda / HIP, relay its output and exit code (never exec from a GPU process)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(insts_sample, cfg, n_cells, masters):
    """Oracle ("port") timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import pyoracle as orc
    from navigation_amd import synth
    L = orc.lib()
    cores, n_logical, quota = host_cores()
    n_inst = len(insts_sample)
    ocfg = orc.DwaConfig(**cfg.as_dict())
    pos = np.ascontiguousarray(np.stack([i["pos"] for i in insts_sample]), np.float32)
    vel = np.ascontiguousarray(np.stack([i["vel"] for i in insts_sample]), np.float32)
    plans = np.ascontiguousarray(np.stack([i["plan"] for i in insts_sample]), np.float64)
    origins = np.zeros((n_inst, 2))
    cells = np.ascontiguousarray(masters, np.uint8)
    fp = np.ascontiguousarray(synth.FOOTPRINT, np.float64)
    cycles = 6  # 32 robots x 6 cycles ~ 20 s of CPU work spread over the host cores
    scored = C.c_uint64()
    dt = L.orc_bench_dwa(n_cells, n_cells, synth.RES, cells, n_inst, C.byref(ocfg), pos, vel, plans, plans.shape[1], origins, fp,
                         len(fp), cycles, cores, C.byref(scored))
    raw = np.ascontiguousarray(np.stack([i["cells"] for i in insts_sample]), np.uint8)
    reps = 8
    dti = L.orc_bench_inflate(raw, n_inst, n_cells, n_cells, synth.RES, synth.INFLATION_RADIUS, synth.COST_SCALING,
                              synth.inscribed_radius(synth.FOOTPRINT), reps, cores)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=scored.value / dt, unit="trajectories/s", cores=cores, kind="port",
                per_core=scored.value / dt / cores, host_cpu=model, host_nproc=n_logical, host_cpu_quota=quota,
                cores_note="threads = CPUs this process may use: affinity mask capped by the cgroup CPU quota (cpu.max)",
                inflation_cells_per_s_per_core=n_inst * reps * n_cells * n_cells / dti / cores,
                sample=f"{n_inst} instances x {cycles} planner cycles (4 MapGrid BFS + rollout + 6 critics, reference early-out), "
                       f"one instance per thread",
                inflation_cells_per_s=n_inst * reps * n_cells * n_cells / dti,
                inflation_sample=f"{n_inst} full-window 400x400 InflationLayer::updateCosts (reference PQ walk) x {reps}",
                seconds=round(dt + dti, 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--instances", type=int, default=256, help="robot instances per GPU")
    ap.add_argument("--total-instances", type=int, default=0,
                    help="strong scaling (SURVEY 8e: 2048 robots over the node): this many robots in all, split over the ranks by "
                         "navigation_amd.sharding.shard_range; 0 = --instances per GPU (weak scaling, the contract workload)")
    ap.add_argument("--size", type=int, default=400, help="costmap cells per side")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true")
    ap.add_argument("--vsamples", default="32,32,16", help="vx,vy,vtheta samples (exploration; the contract workload is 32,32,16)")
    ap.add_argument("--footprint", default="square", choices=["square", "poly5"], help="poly5: costmap_params.yaml's 5-vertex polygon")
    ap.add_argument("--fixed-poses", action="store_true", help="every cycle sees the same poses (A/B only; the default perturbs them per cycle)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-process rehearsal on a box with ONE GPU: every rank uses device 0, collectives over gloo")
    args = ap.parse_args()

    # ---- N ranks.  Nothing above or below this block has touched torch / HIP yet.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch one rank per GPU", file=sys.stderr)
        sys.exit(2)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    import navigation_amd as nav
    from navigation_amd import _lib as N, synth

    nav.lib()  # fails loudly if the HIP extension is missing
    n_inst, n_cells = args.instances, args.size
    seed0 = rank * n_inst
    if args.total_instances:
        from navigation_amd.sharding import shard_range
        seed0, n_inst = shard_range(args.total_instances, rank, world)
    vs = tuple(int(v) for v in args.vsamples.split(","))
    fl, insts, cfg = build_fleet(nav, n_inst, n_cells, seed0=seed0, device=local_rank, vs=vs, footprint=args.footprint)
    _, _, pos_h, vel_h, plans_h = fl._bench_host_inputs
    poses = None if args.fixed_poses else PoseSchedule(pos_h, vel_h, 64, seed=4242 + rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    kk = 0  # running cycle number: every cycle of the run takes the next pose of the schedule
    for _ in range(args.warmup):
        step(fl, poses, kk)
        kk += 1
    fl.sync()
    # every kernel bracketed by HIP events over a few untimed steps: finds the dominant kernel and gives the others'
    # durations; the timed region then brackets the dominant kernel only (each pair of events costs the stream a few
    # microseconds: 47 us per step with all six regions bracketed)
    fl.profile_select(None)
    fl.profile(True)
    fl.profile_reset()
    pre_steps = max(3, min(10, args.steps))
    for _ in range(pre_steps):
        step(fl, poses, kk)
        kk += 1
    prof_all = fl.profile_read()
    dom_pre = max(prof_all, key=lambda k: prof_all[k][0])
    fl.profile_select([dom_pre])
    fl.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(fl, poses, kk)
        kk += 1
    fl.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = dict(prof_all)
    prof[dom_pre] = fl.profile_read()[dom_pre]  # the dominant kernel: measured live over the timed region
    fl.profile(False)
    fl.profile_select(None)

    res = fl.results()
    scored = sum(r.n_scored for r in res)
    boxes = fl.bounds()
    win_cells = int(((boxes[:, 1] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 2])).sum())
    # what a bounded wavefront has to move: its region's distances; the region = the robot's box + 2 cells,
    # whole 32-cell words
    wb = fl.wavefront_boxes().astype(np.int64)
    reg_rows = np.minimum(wb[:, 3] + 2, n_cells - 1) - np.maximum(wb[:, 2] - 2, 0) + 1
    reg_words = np.minimum(wb[:, 1] + 2, n_cells - 1) // 32 - np.maximum(wb[:, 0] - 2, 0) // 32 + 1
    region_cells = int((reg_rows * reg_words * 32).sum())

    # the only collective: throughput counters over RCCL (navigation_amd/sharding.py, gloo-tested on CPU)
    from navigation_amd.sharding import reduce_counters
    elapsed_max, (total_scored, total_win) = reduce_counters(dist, elapsed, [scored, win_cells],
                                                             device="cpu" if args.rehearse_on_one_gpu else "cuda")

    out = None
    if rank == 0:
        ms_per_step = elapsed_max / args.steps * 1e3
        traj_per_s = total_scored * args.steps / elapsed_max
        dom = dom_pre
        avg_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()}
        bitmap = n_cells * ((n_cells + 31) // 32) * 4
        alg_bytes = {
            "k_score": BYTES_PER_TRAJ * scored,
            # bounded search: costmap bytes once per robot -> traversable bitmap (k_free_bits), the bitmap read twice per
            # grid, 4 B per cell of the region written.  (SURVEY 8d's 5 B x every cell is the WHOLE-grid search's figure
            # and is used for the whole_grid_wavefronts leg only.)
            "k_bfs": n_inst * (n_cells * n_cells + bitmap + 3 * 2 * bitmap) + 3 * 4 * region_cells,
            "k_inflate": BYTES_PER_INFL_CELL * win_cells,
            "k_merge": BYTES_PER_MERGE_CELL * win_cells,
            "k_obstacle": 0.0, "k_select": 0.0,
        }
        contract = (n_inst, n_cells) == (256, 400) and vs == (32, 32, 16) and args.footprint == "square"
        achieved = alg_bytes[dom] / (avg_ms[dom] * 1e-3) / 1e9 if avg_ms[dom] > 0 else 0.0
        traffic, traffic_src = hbm_traffic_from_profiles(dom) if contract else (None, None)
        pmc, pmc_src = pmc_summary_from_profiles() if contract else (None, None)
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "achieved_is": "ALGORITHMIC bytes (SURVEY 8d: 1000 B per scored trajectory) / launch time - a figure of merit, not HBM utilisation",
                "algorithmic_bytes_per_launch": alg_bytes[dom], "avg_launch_ms": avg_ms[dom],
                "frac_vs_measured_copy_peak_6290": achieved / 6290.0}
        if traffic and avg_ms[dom] > 0:
            roof["hbm_measured"] = {"GBps": traffic / (avg_ms[dom] * 1e-3) / 1e9, "frac_of_peak": traffic / (avg_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "bytes_per_launch": traffic, "note": "PMC FETCH_SIZE x2 + WRITE_SIZE (gfx950 correction), separate passes"}
        if pmc and dom in pmc.get("regions", {}):
            v = pmc["regions"][dom]
            roof["bound"] = v.get("bound", "valu")
            roof["valu_busy"] = v.get("valu_busy")
            roof["valu"] = dict(v, source=pmc_src)
        out = {
            "metric": "scored trajectories/sec (whole node) + costmap inflation cells/sec, 400x400 map",
            "value": traj_per_s, "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.total_instances else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"configs[2]: {n_inst} batched robot instances per MI355X, {n_cells}x{n_cells} costmaps "
                                   f"+ inflation, 32x32x16 velocity samples, 20 sim steps, LaserScan (720 beams) update "
                                   f"each cycle; " + (f"{args.total_instances} instances in all, split over the ranks (strong scaling)" if args.total_instances
                                                      else f"N GPUs = N x {n_inst} instances (N=8 is configs[3])"),
                       "instances_per_gpu": n_inst, "costmap": f"{n_cells}x{n_cells}@0.05", "vsamples": "x".join(str(v) for v in vs),
                       "sim_steps": T_STEPS, "critics": "oscillation+obstacle+goal_front+alignment+path+goal",
                       "parallelism": f"fleet-shard x{world}"},
            "value_is": "costmaps, scans and plans resident in HBM; every timed cycle stages its own (perturbed) pose + velocity, "
                        "24 B per robot H2D, and its results land in pinned host memory; the fully PCIe-inclusive rate "
                        "(scans + plans + poses re-staged every cycle) is pcie_inclusive",
            "poses": "fixed" if poses is None else "per-cycle N(0, 2 cm / 0.05 rad) around the base pose, seeded",
            "per_instance_trajectories_per_s": traj_per_s / (n_inst * world),
            "inflation_cells_per_s": total_win * args.steps / elapsed_max,
            "inflation_window_cells_per_step": total_win,
            "trajectories_per_step": total_scored,
            "kernel_ms": {k: round(avg_ms[k], 4) for k in avg_ms},
            "kernel_ms_source": f"HIP events on the library's stream: {dom} over the {args.steps} timed steps, the others over {pre_steps} untimed steps before them",
            "roofline": roof,
        }
        # every kernel against the same roofline, and the whole step as SURVEY 8(d) defines it
        per_kernel = {}
        for k in avg_ms:
            if avg_ms[k] > 0 and alg_bytes.get(k, 0) > 0:
                gbs = alg_bytes[k] / (avg_ms[k] * 1e-3) / 1e9
                tk, _ = hbm_traffic_from_profiles(k) if contract else (None, None)
                per_kernel[k] = {"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "avg_launch_ms": avg_ms[k], "traffic": tk,
                                 "algorithmic_bytes_per_launch": alg_bytes[k]}
                if pmc and k in pmc.get("regions", {}):
                    per_kernel[k]["valu_busy"] = pmc["regions"][k].get("valu_busy")
        out["roofline_all"] = per_kernel
        step_bytes = sum(alg_bytes.values())
        out["roofline_step"] = {"algorithmic_bytes_per_step": step_bytes, "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9,
                                "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s"}
    # ---- extra legs on rank 0 at N=1 only
    if rank == 0 and world == 1:
        # per-cycle latency of the same step, one cycle in flight at a time (median / p99 over >= 50 cycles)
        lat = []
        for _ in range(max(60, args.steps)):
            tc = time.perf_counter()
            step(fl, poses, kk)
            kk += 1
            fl.sync()
            lat.append((time.perf_counter() - tc) * 1e3)
        lat.sort()
        out["cycle_latency"] = {"cycles": len(lat), "ms_median": lat[len(lat) // 2], "ms_p99": lat[min(len(lat) - 1, int(0.99 * len(lat)))],
                                "ms_min": lat[0], "ms_max": lat[-1], "note": "stage pose -> updateMap -> findBestPath -> results visible, synchronised every cycle"}
        # the same step with every MapGrid wavefront run over the whole costmap, as the reference does (the default stops a
        # search once the box its robot's samples can reach is settled; planner results are identical, tests/test_gpu_parity.py)
        if not args.no_single:  # (the profiling runs of tools/collect_profiles.sh leave it out: one kind of launch per kernel)
            lv_bounded = fl.wavefront_levels().mean(axis=0)
            fl.set_bounded_map_grids(False)
            fl.stage_planner(pos_h, vel_h, plans_h)
            for _ in range(3):
                step(fl, poses, kk)
                kk += 1
            fl.sync()
            fl.profile(True)
            fl.profile_reset()
            k2 = 20
            t1 = time.perf_counter()
            for _ in range(k2):
                step(fl, poses, kk)
                kk += 1
            fl.sync()
            dt = time.perf_counter() - t1
            pk = fl.profile_read()["k_bfs"]
            fl.profile(False)
            lv_whole = fl.wavefront_levels().mean(axis=0)
            wg_ms = pk[0] / max(pk[1], 1)
            wg_bytes = BYTES_PER_BFS_CELL * 3 * n_cells * n_cells * n_inst
            out["whole_grid_wavefronts"] = {"ms_per_step": dt / k2 * 1e3, "trajectories_per_s": scored * k2 / dt, "k_bfs_ms": wg_ms,
                                            "levels_path_goal_front": [float(v) for v in lv_whole],
                                            "roofline": {"algorithmic_bytes_per_launch": wg_bytes, "achieved": wg_bytes / (wg_ms * 1e-3) / 1e9,
                                                         "frac": wg_bytes / (wg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s",
                                                         "note": "SURVEY 8d: 5 B per cell and grid"}}
            out["bounded_wavefronts"] = {"enabled": True, "levels_path_goal_front": [float(v) for v in lv_bounded],
                                         "region_cells_per_robot": region_cells / n_inst,
                                         "note": "value / ms_per_step are measured with bounded wavefronts (library default)"}
            fl.set_bounded_map_grids(True)
            fl.stage_planner(pos_h, vel_h, plans_h)
            step(fl)
            fl.sync()
        # full-window inflation throughput (the BASELINE.md probe shape)
        raw = np.stack([i["cells"] for i in insts])
        fl.upload(N.GRID_MASTER, raw)
        full = [[0, 0, n_cells, n_cells]] * n_inst
        fl.inflate(boxes=full)
        fl.sync()
        fl.profile(True)
        fl.profile_reset()
        reps = 10
        t1 = time.perf_counter()
        for _ in range(reps):
            fl.inflate(boxes=full)
        fl.sync()
        dt = time.perf_counter() - t1
        pk = fl.profile_read()["k_inflate"]
        fl.profile(False)
        out["inflation_full_window"] = {"cells_per_s": reps * n_inst * n_cells * n_cells / dt,
                                        "kernel_ms": pk[0] / max(pk[1], 1),
                                        "achieved_GBps": BYTES_PER_INFL_CELL * n_inst * n_cells * n_cells / (pk[0] / max(pk[1], 1) * 1e-3) / 1e9}
        # PCIe-inclusive rate (SURVEY 8d's timing protocol): every cycle re-stages ALL its inputs from pageable host
        # memory (scan clouds, plans, poses: H2D) and reads the results (D2H) through the C-ABI.
        poses_h, obs_arr, n_obs, pts_h, states_h, n_st, plans_pk = raw_inputs(fl)
        sched = []
        for k in range(8):  # a few pre-marshalled pose sets so that these cycles differ too
            st_k = (type(states_h[0]) * n_st)()
            C.memmove(st_k, states_h, C.sizeof(states_h))
            if poses is not None:
                for i in range(n_st):
                    st_k[i].pos[:] = [float(v) for v in poses.pos[k][i]]
            sched.append(st_k)
        fl.stage_observations_raw(poses_h, obs_arr, n_obs, pts_h)
        fl.stage_planner_raw(states_h, n_st, plans_pk)
        step(fl)
        fl.sync()
        kp = 300  # enough cycles for a p99 that is not the maximum
        import gc
        gc.collect()
        gc.disable()
        t1 = time.perf_counter()
        from navigation_amd._lib import PlanResult
        rbuf = (PlanResult * n_st)()  # reused: no per-cycle Python allocation (see Fleet.results_into)
        cyc = []
        n_sc = 0
        for k in range(kp):
            tc = time.perf_counter()
            fl.stage_observations_raw(poses_h, obs_arr, n_obs, pts_h)
            fl.stage_planner_raw(sched[k % len(sched)], n_st, plans_pk)
            step(fl)
            rr = fl.results_into(rbuf)
            cyc.append(time.perf_counter() - tc)
        dp = time.perf_counter() - t1
        gc.enable()
        n_sc = sum(r.n_scored for r in rr)
        h2d = poses_h.nbytes + pts_h.nbytes + plans_pk.nbytes + n_obs * 56 + n_st * 32
        worst = int(np.argmax(cyc))
        worst_ms = cyc[worst] * 1e3
        cyc.sort()
        out["pcie_inclusive"] = {"trajectories_per_s": n_sc * kp / dp, "ms_per_step": dp / kp * 1e3,
                                 "cycle_ms_median": cyc[len(cyc) // 2] * 1e3, "cycle_ms_p99": cyc[min(len(cyc) - 1, int(0.99 * len(cyc)))] * 1e3,
                                 "cycle_ms_max": worst_ms, "worst_cycle_index": worst, "cycles": kp, "h2d_bytes_per_step": h2d, "d2h_bytes_per_step": n_st * 72,
                                 "note": "caller buffers are pageable; the library stages them through pinned mirrors"}
        fl.upload(N.GRID_MASTER, raw)
        fl.inflate(boxes=full)
        masters = fl.master(0, min(n_inst, 32))
        if not args.no_single:
            f1, i1, c1 = build_fleet(nav, 1, n_cells, seed0=0, device=local_rank)
            p1 = PoseSchedule(f1._bench_host_inputs[2], f1._bench_host_inputs[3], 64, seed=7)
            for k in range(3):
                step(f1, p1, k)
            f1.sync()
            t1 = time.perf_counter()
            k1 = 50
            for k in range(k1):
                step(f1, p1, 3 + k)
            f1.sync()
            d1 = time.perf_counter() - t1
            r1 = f1.results()[0]
            out["single_robot"] = {"workload": "configs[1]: one robot, 400x400 + inflation, 32x32x16, 20 steps",
                                   "ms_per_cycle": d1 / k1 * 1e3, "trajectories_per_s": r1.n_scored * k1 / d1,
                                   "n_scored": r1.n_scored}
            f1.close()
        if not args.no_single:
            out["legacy_trajectory_planner"] = legacy_leg(nav, insts, n_cells, masters, not args.no_cpu_baseline, local_rank)
            out["configs4_one_gpu_share"] = configs4_leg(nav, local_rank)
        if not args.no_cpu_baseline:
            ns = min(n_inst, 32)
            out["cpu_baseline"] = cpu_baseline(insts[:ns], cfg, n_cells, masters[:ns])
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    fl.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
