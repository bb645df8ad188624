#!/usr/bin/env python3
"""bench.py - fleet throughput of the MI355X-native costmap + DWA hot path.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 launched by torch.distributed.run, one rank per GPU).
One *step* = one control cycle of every robot of the rank's fleet, by SURVEY 8(d)'s protocol: the cycle's NEW LaserScan
cloud, its pose / velocity and its plan are handed over from host memory (H2D), then LayeredCostmap::updateMap
(clearing + marking + merge + inflation) and DWAPlanner::findBestPath (3 MapGrid wavefronts + rollout + six critics +
selection), and the results come back to host memory (D2H).  The costmaps themselves stay resident in HBM.

Workload (config.workload): BASELINE.json's metric is a whole-node throughput, quoted on the fleet configurations;
configs[3] (2048 instances over 8 GPUs) does not fit one GPU, so at N=1 the workload is the largest single-GPU
configuration, configs[2]: 256 batched robot instances on one MI355X, 400x400 costmaps, 32x32x16 samples, 20 sim steps,
a new LaserScan every cycle.  At N GPUs every rank runs the same 256 instances (weak scaling; N=8 is configs[3]).

The fleet runs as --groups G independent groups of 256 / G robots, each with its own HIP stream (a navgpu fleet = one
stream; robots are independent, so any partition gives the same results).  While the host hands group g's next cycle
over, the other groups' cycles are queued on the GPU: the PCIe transfers hide behind their kernels, and the
latency-bound wavefront kernels of one group overlap the issue-bound scoring kernels of another.

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

# A HIP runtime setting of this PROCESS (read when the runtime starts, i.e. before anything below touches the GPU): host-to-device copies
# of up to 2 MiB go through the runtime's copy kernels instead of the SDMA engines.  Every hand-over of the step is such a copy (a group's
# packed scan block is 0.6 MB), and through SDMA one hipMemcpyAsync in ~2 500 stalls for ~8 ms inside the runtime - about every 150 fleet
# cycles (tools/probe_cycle_outliers.py).  Not set when the caller has set it; the line says what was in force (config.env).
os.environ.setdefault("GPU_FORCE_BLIT_COPY_SIZE", "2048")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
T_STEPS, P_PERIM, G_GRIDS = 20, 32, 4
BYTES_PER_TRAJ = 20 + T_STEPS * (P_PERIM + 1) + T_STEPS * G_GRIDS * 4  # = 1000 B
BYTES_PER_BFS_CELL = 5       # 1 B costmap read + 4 B distance write, per grid
BYTES_PER_INFL_CELL = 2
BYTES_PER_MERGE_CELL = 3
SCAN_CYCLES = 8              # distinct pre-marshalled LaserScan clouds per robot (the discs move from one to the next)


def build_fleet(nav, n_inst, n_cells, seed0, device=0, vs=(32, 32, 16), footprint="square", insts=None):
    from navigation_amd import _lib as N, synth
    fp = synth.FOOTPRINT5 if footprint == "poly5" else synth.FOOTPRINT
    fl = nav.Fleet(n_inst, n_cells, n_cells, synth.RES, layers=N.LAYER_STATIC | N.LAYER_OBSTACLE | N.LAYER_INFLATION,
                   max_points=720, max_observations=1, max_plan=200, max_footprint=8, max_sim_steps=24, device=device)
    fl.configure_obstacle()
    fl.set_footprint(fp)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(fp))
    cfg = synth.fleet_config(*vs)
    fl.configure_planner(cfg)
    if insts is None:
        insts = [synth.make_instance(n_cells, seed0 + i) for i in range(n_inst)]
    for i, ins in enumerate(insts):
        fl.add_static_map(np.where(ins["cells"] == 254, 100, 0).astype(np.int8), first=i, count=1)
    poses = np.array([[float(v) for v in ins["pos"]] for ins in insts])
    obs = []
    for i, ins in enumerate(insts):
        pts = cached_scan(synth, ins, 0)
        obs.append(dict(instance=i, points=pts, origin=(float(ins["pos"][0]), float(ins["pos"][1]), 0.3),
                        obstacle_range=2.5, raytrace_range=3.0))
    fl.stage_observations(poses, obs)
    fl.stage_planner(np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]),
                     np.stack([i["plan"] for i in insts]))
    fl.set_plan()
    fl._bench_host_inputs = (poses, obs, np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]),
                             np.stack([i["plan"] for i in insts]))
    fl._bench_insts = insts
    return fl, insts, cfg


def step(fl, poses=None, k=0):
    """One pass of the hot path over a fleet with RESIDENT scans and plans (legs and probes; the timed region uses
    Group.cycle).  With `poses` (a PoseSchedule) the cycle first stages its own pose and velocity."""
    if poses is not None:
        fl.stage_poses(poses.pos[k % len(poses.pos)], poses.vel)
    fl.update_map()
    fl.planner_cycle()


class PoseSchedule:
    """Seeded per-cycle poses: the robot of cycle k stands at base + N(0, 2 cm) and is turned by N(0, 0.05 rad), so that
    successive cycles differ (wavefront lengths, reach boxes, which samples collide)."""

    def __init__(self, pos, vel, n_cycles, seed):
        rs = np.random.RandomState(seed)
        d = rs.normal(size=(n_cycles,) + pos.shape) * np.array([0.02, 0.02, 0.05])
        self.pos = [np.ascontiguousarray(pos + d[k], np.float32) for k in range(n_cycles)]
        self.vel = np.ascontiguousarray(vel, np.float32)


def cached_scan(synth, ins, cycle):
    """synth.laser_scan(ins, cycle) as float32, kept with the robot's own record (the one-stream pre-pass re-uses the groups' clouds)."""
    scans = ins.setdefault("_scans", {})
    if cycle not in scans:
        scans[cycle] = np.ascontiguousarray(synth.laser_scan(ins, cycle), np.float32)
    return scans[cycle]


class Group:
    """One stream's share of the fleet with everything a cycle hands over pre-marshalled (the timed loop calls the
    C-ABI staging entry points on ready buffers: no Python list handling inside it)."""

    def __init__(self, nav, fl, insts, seed, stub=False):
        from navigation_amd._lib import OBS_CLEARING, OBS_MARKING, Observation, PlanResult, RobotState
        from navigation_amd import synth
        self.fl, self.n = fl, len(insts)
        poses_h, _, pos, vel, plans = fl._bench_host_inputs
        self.poses_h = np.ascontiguousarray(poses_h, np.float64)
        self.sched = PoseSchedule(pos, vel, 64, seed)
        self.plans_pk = np.ascontiguousarray(plans, np.float64).reshape(-1, 2)
        k_plan = plans.shape[1]
        # the scans of SCAN_CYCLES consecutive control cycles (3 moving discs + the static cells, synth.laser_scan)
        self.scans = []
        for c in range(SCAN_CYCLES):
            arr = (Observation * self.n)()
            pts, off = [], 0
            for i, ins in enumerate(insts):
                p = np.zeros((4, 3), np.float32) if stub else cached_scan(synth, ins, c)
                arr[i] = Observation(i, off, len(p), OBS_MARKING | OBS_CLEARING, float(ins["pos"][0]), float(ins["pos"][1]), 0.3, 2.5, 3.0)
                pts.append(p)
                off += len(p)
            self.scans.append((arr, np.ascontiguousarray(np.concatenate(pts), np.float32)))
        self.states = []
        for c in range(16):
            st = (RobotState * self.n)()
            for i in range(self.n):
                st[i].pos[:] = [float(v) for v in self.sched.pos[c][i]]
                st[i].vel[:] = [float(v) for v in vel[i]]
                st[i].plan_first = i * k_plan
                st[i].plan_count = k_plan
            self.states.append(st)
        self.rbuf = (PlanResult * self.n)()
        self.rview = np.frombuffer(self.rbuf, dtype=np.dtype(PlanResult))
        self.scored = 0      # trajectories scored by the cycles whose results have been read
        self.pending = False  # a cycle is queued whose results have not been read yet
        self.depth = 1       # cycles in flight on the stream (set_depth)
        self.h2d_bytes = self.poses_h.nbytes + self.scans[0][1].nbytes + self.plans_pk.nbytes + self.n * (56 + 32)

    def collect(self):
        """Results of the queued cycle -> host (waits for the group's stream)."""
        if self.pending:
            self.fl.results_into(self.rbuf)
            self.scored += int(self.rview["n_scored"].sum())
            self.pending = False

    def set_depth(self, depth):
        """Cycles in flight on the group's stream (navgpu_planner_set_cycles_in_flight): with 2, cycle k is handed over and
        queued BEFORE the results of cycle k - 1 are read, so the stream goes straight from one cycle into the next."""
        self.collect()
        self.fl.set_cycles_in_flight(depth)
        self.depth = depth

    def cycle(self, k, restage=True):
        """One control cycle: results of the previous one, hand over cycle k's inputs, queue updateMap + findBestPath."""
        if self.depth < 2:
            self.collect()
        if restage:
            arr, pts = self.scans[k % SCAN_CYCLES]
            self.fl.stage_observations_raw(self.poses_h, arr, self.n, pts)
            self.fl.stage_planner_raw(self.states[k % len(self.states)], self.n, self.plans_pk)
        else:
            self.fl.stage_poses(self.sched.pos[k % len(self.sched.pos)], self.sched.vel)
        self.fl.update_map()
        self.fl.planner_cycle()
        if self.depth >= 2 and self.pending:  # cycle k - 1's results, while cycle k runs
            self.fl.results_previous_into(self.rbuf)
            self.scored += int(self.rview["n_scored"].sum())
        self.pending = True


def run_cycles(groups, k0, n, restage=True):
    for k in range(k0, k0 + n):
        for g in groups:
            g.cycle(k, restage)
    for g in groups:
        g.collect()
    return k0 + n


def current_profile():
    """The rocprofv3 profile the line quotes: profiles/CURRENT.json (written by tools/collect_profiles.sh -> tools/profile_tag.py: tag, git
    head, a hash of the library's sources, compile flags) - an explicit tag, not the lexically last file - and whether the sources of
    the build that is RUNNING are the ones it was taken from."""
    try:
        cur = json.load(open(os.path.join(ROOT, "profiles", "CURRENT.json")))
    except Exception:
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from profile_tag import sources_sha16
        now = sources_sha16()
    except Exception:
        now = None
    cur["running_sources_sha16"] = now
    cur["matches_running_build"] = bool(now) and now == cur.get("sources_sha16")
    return cur


def hbm_traffic_from_profiles(kernel, cur):
    """HBM bytes per launch of `kernel` as measured by rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this workload (one full-fleet launch
    per kernel: tools/collect_profiles.sh runs --groups 1), from the CURRENT profile's <tag>_hbm_traffic.json."""
    if not cur:
        return None, None
    f = os.path.join(ROOT, "profiles", f"{cur['tag']}_hbm_traffic.json")
    try:
        d = json.load(open(f))
        # MI355X_MICROARCH.md (HBM): gfx950 FETCH_SIZE tallies 128-B read requests at 64 B -> doubled; WRITE_SIZE is exact
        return d["kernels"][kernel]["hbm_bytes_fetch_x2"], os.path.relpath(f, ROOT)
    except Exception:
        return None, None


def pmc_summary_from_profiles(cur):
    """The CURRENT profile's <tag>_pmc_summary.json (tools/collect_profiles.sh + tools/pmc_summary.py): the measured issue roofline."""
    if not cur:
        return None, None
    f = os.path.join(ROOT, "profiles", f"{cur['tag']}_pmc_summary.json")
    try:
        return json.load(open(f)), os.path.relpath(f, ROOT)
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


def navfn_leg(nav, fleet, n_robots, device):
    """SURVEY 8 row f-4 (navfn::NavFn, once per replan): one global plan per robot across its own master grid, handed over on
    the device - the tiled wavefront mode (navgpu_navfn_plan_wavefront) and the reference-order mode (navgpu_navfn_plan, bit-exact,
    one lane per plan) on the same plans."""
    n = int(fleet.nx)
    nf = nav.NavFn(n, n, n_robots, device=device)
    nf.set_costmap_from_fleet(fleet, count=n_robots)
    goals = np.tile(np.array([[n // 2, n // 2]], np.int32), (n_robots, 1))
    starts = np.array([[12 + (7 * i) % 24, n - 13 - (11 * i) % 24] if i % 2 else [n - 13 - (5 * i) % 24, 12 + (3 * i) % 24] for i in range(n_robots)], np.int32)
    out = {}
    for name, fn in (("wavefront", lambda: nf.plan_wavefront(goals, starts)), ("reference_order", lambda: nf.plan(goals, starts))):
        fn()
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            res = fn()
            t.append((time.perf_counter() - t0) * 1e3)
        out[name] = {"ms_per_batch": min(t), "plans": n_robots, "found": int(sum(r.found for r in res)),
                     "mean_path_points": float(np.mean([r.path_length for r in res]))}
    nf.close()
    out["note"] = (f"{n_robots} plans side by side on {n} x {n} master grids (corner of the map -> its centre), costmaps handed over on the device; "
                   "wavefront: updateCell's rule relaxed to its fixed point by 32 x 32 LDS tiles, potentials <= the reference's, paths of equal cost "
                   "(tests/test_navfn.py); reference_order: the reference's priority buffers replayed bit for bit, one lane per plan")
    return out


def inflation_reference_order_leg(nav, insts, n_cells, device):
    """SURVEY a10 in the reference's own order (priority_queue_order = 1: InflationLayer::updateCosts' priority-queue walk,
    byte-identical to the reference) on the contract workload: static + obstacle + inflation layers, a new scan per cycle,
    the per-cycle update windows.  ms per costmap update of the whole fleet."""
    from navigation_amd import _lib as N, synth
    n_inst = len(insts)
    fl = nav.Fleet(n_inst, n_cells, n_cells, synth.RES, layers=N.LAYER_STATIC | N.LAYER_OBSTACLE | N.LAYER_INFLATION, max_points=720,
                   max_observations=1, device=device)
    fl.configure_obstacle()
    fl.set_footprint(synth.FOOTPRINT)
    fl.configure_inflation(synth.INFLATION_RADIUS, synth.COST_SCALING, synth.inscribed_radius(synth.FOOTPRINT), priority_queue_order=True)
    for i, ins in enumerate(insts):
        fl.add_static_map(np.where(ins["cells"] == 254, 100, 0).astype(np.int8), first=i, count=1)
    poses = np.array([[float(v) for v in ins["pos"]] for ins in insts])

    def stage(k):
        fl.stage_observations(poses, [dict(instance=i, points=cached_scan(synth, ins, k), origin=(float(ins["pos"][0]), float(ins["pos"][1]), 0.3),
                                           obstacle_range=2.5, raytrace_range=3.0) for i, ins in enumerate(insts)])
    stage(0)
    t0 = time.perf_counter()
    fl.update_map()
    fl.sync()
    first_ms = (time.perf_counter() - t0) * 1e3
    stage(1)
    fl.update_map()  # (the inflation layer's second update still covers the whole map: last cycle's box)
    fl.sync()
    ms, cells = [], 0
    for k in range(2, 5):
        stage(k)
        fl.sync()
        t0 = time.perf_counter()
        fl.update_map()
        fl.sync()
        ms.append((time.perf_counter() - t0) * 1e3)
        b = fl.bounds()
        cells = int(((b[:, 1] - b[:, 0]) * (b[:, 3] - b[:, 2])).sum())
    fl.close()
    ms.sort()
    return {"workload": f"{n_inst} robots, {n_cells}x{n_cells}, static + obstacle + inflation, new scan per cycle, priority_queue_order = 1",
            "ms_per_update_per_cycle_windows": ms[len(ms) // 2], "window_cells_per_update": cells,
            "cells_per_s": cells / (ms[len(ms) // 2] * 1e-3), "ms_first_update_whole_maps": first_ms,
            "note": "byte-identical to the reference's std::priority_queue walk (tests: test_inflate_reference_priority_queue_order); one lane per "
                    "robot walks the heap, 256 robots side by side; the default mode (exact transform, >= the reference everywhere) is what `value` uses"}


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


def oracle_build_flags():
    """How the CPU port was compiled (oracle/Makefile): -O2 without -march=native or FMA contraction - its doubles have to round like the
    reference's default x86-64 build - where SURVEY 8(d) suggested -O3 -march=native."""
    try:
        for line in open(os.path.join(ROOT, "oracle", "Makefile")):
            if line.startswith("CXXFLAGS"):
                return "g++ " + line.split("=", 1)[1].strip() + " (no -march=native, no FMA contraction: the doubles must round like the reference's default x86-64 build)"
    except Exception:
        pass
    return None


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
    return dict(value=scored.value / dt, unit="trajectories/s", cores=cores, kind="port", build=oracle_build_flags(),
                per_core=scored.value / dt / cores, host_cpu=model, host_nproc=n_logical, host_cpu_quota=quota,
                cores_note="threads = CPUs this process may use: affinity mask capped by the cgroup CPU quota (cpu.max)",
                inflation_cells_per_s_per_core=n_inst * reps * n_cells * n_cells / dti / cores,
                sample=f"{n_inst} instances x {cycles} planner cycles (4 MapGrid BFS + rollout + 6 critics, reference early-out), "
                       f"one instance per thread",
                inflation_cells_per_s=n_inst * reps * n_cells * n_cells / dti,
                inflation_sample=f"{n_inst} full-window 400x400 InflationLayer::updateCosts (reference PQ walk) x {reps}",
                seconds=round(dt + dti, 2))


class _StubFleet:
    """Host-only stand-in for navigation_amd.Fleet (--stub-fleet: tests/test_distributed_counters.py drives bench.py's N > 1
    code path - rank bookkeeping, counter all-reduce, the JSON line - under gloo on a box without a GPU).  It computes
    nothing: a cycle is a short sleep and every robot "scores" 17 000 trajectories."""

    def __init__(self, n):
        self.n = n

    def _noop(self, *a, **k):
        return None

    stage_observations_raw = stage_planner_raw = stage_poses = update_map = sync = profile = profile_reset = profile_select = close = _noop
    set_bounded_map_grids = stage_planner = inflate = upload = _noop

    def planner_cycle(self):
        time.sleep(0.0005)

    def results_into(self, buf):
        for r in buf:
            r.n_scored = 17000
        return buf

    def profile_read(self):
        return {k: (0.05, 1) for k in ("k_obstacle", "k_merge", "k_inflate", "k_bfs", "k_score", "k_select")}

    def bounds(self):
        return np.tile(np.array([[100, 240, 100, 240]], np.int32), (self.n, 1))

    def wavefront_boxes(self):
        return np.tile(np.array([[170, 230, 170, 230]], np.int32), (self.n, 1))


def stub_group_fleet(n, n_cells, seed0):
    from navigation_amd import synth
    insts = [dict(pos=np.array([10.0, 10.0, 0.1 * i], np.float32), vel=np.array([0.2, 0.0, 0.0], np.float32),
                  plan=np.stack([10.0 + 0.04 * np.arange(200), np.full(200, 10.0)], 1)) for i in range(n)]
    fl = _StubFleet(n)
    fl._bench_host_inputs = (np.array([[10.0, 10.0, 0.0]] * n), None, np.stack([i["pos"] for i in insts]), np.stack([i["vel"] for i in insts]),
                             np.stack([i["plan"] for i in insts]))
    fl._bench_insts = insts
    return fl, insts, synth.fleet_config()


def device_identity(torch, local_rank, stub):
    """What tells two ranks on ONE device apart from two ranks on two: PCI domain:bus:device of the rank's GPU."""
    if stub:
        return f"stub:{local_rank}"
    try:
        p = torch.cuda.get_device_properties(local_rank)
        if hasattr(p, "pci_bus_id"):
            return f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}"
        return str(getattr(p, "uuid", local_rank))
    except Exception:
        return f"cuda:{local_rank}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--instances", type=int, default=256, help="robot instances per GPU")
    ap.add_argument("--groups", type=int, default=4, help="independent groups (= HIP streams) the rank's fleet runs as; 1 = one stream, serial")
    ap.add_argument("--cycles-in-flight", type=int, default=2, choices=[1, 2],
                    help="2 (default since round 4's last build: a group's chain of launches is 0.59 ms and its hand-over 0.13 ms, so a stream "
                         "that waits for its results idles a fifth of the time): a group's cycle k + 1 is handed over and queued while its cycle "
                         "k runs, the results of k are read after that - every cycle's inputs are still new, every cycle's results are still read "
                         "inside the timed region; 1: results of cycle k are read before cycle k + 1 is handed over")
    ap.add_argument("--total-instances", type=int, default=0,
                    help="strong scaling (SURVEY 8e: 2048 robots over the node): this many robots in all, split over the ranks by "
                         "navigation_amd.sharding.shard_range; 0 = --instances per GPU (weak scaling, the contract workload)")
    ap.add_argument("--size", type=int, default=400, help="costmap cells per side")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="leave the supplementary legs out (profiling runs)")
    ap.add_argument("--vsamples", default="32,32,16", help="vx,vy,vtheta samples (exploration; the contract workload is 32,32,16)")
    ap.add_argument("--footprint", default="square", choices=["square", "poly5"], help="poly5: costmap_params.yaml's 5-vertex polygon")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-process rehearsal on a box with ONE GPU: every rank uses device 0, collectives over gloo")
    ap.add_argument("--stub-fleet", action="store_true", help="host-only stand-in for the fleet (CPU tests of the N > 1 code path; computes nothing)")
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
    stub = args.stub_fleet
    host_side = stub or args.rehearse_on_one_gpu  # collectives over gloo on CPU tensors
    dist = None
    if args.rehearse_on_one_gpu or stub:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        if not stub:
            torch.cuda.set_device(local_rank)
        if host_side:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif not stub:
        torch.cuda.set_device(local_rank)
    import navigation_amd as nav
    from navigation_amd import _lib as N, synth

    if not stub:
        nav.lib()  # fails loudly if the HIP extension is missing
    n_inst, n_cells = args.instances, args.size
    seed0 = rank * n_inst
    if args.total_instances:
        from navigation_amd.sharding import shard_range
        seed0, n_inst = shard_range(args.total_instances, rank, world)
    vs = tuple(int(v) for v in args.vsamples.split(","))
    G = max(1, min(args.groups, n_inst))
    from navigation_amd.sharding import shard_range as split
    groups, insts, cfg = [], [], None
    for gi in range(G):
        g0, gn = split(n_inst, gi, G)
        if stub:
            fl, gi_insts, cfg = stub_group_fleet(gn, n_cells, seed0 + g0)
        else:
            fl, gi_insts, cfg = build_fleet(nav, gn, n_cells, seed0=seed0 + g0, device=local_rank, vs=vs, footprint=args.footprint)
        groups.append(Group(nav, fl, gi_insts, seed=4242 + rank * 64 + gi, stub=stub))
        insts += gi_insts

    depth = 1 if stub else args.cycles_in_flight
    if depth > 1:
        for g in groups:
            g.set_depth(depth)

    def sync_all():
        for g in groups:
            g.fl.sync()

    def barrier():
        if not stub:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    # ---- who runs where (an N-rank run on fewer than N devices is a rehearsal and must say so)
    ident = device_identity(torch, local_rank, stub)
    ranks_seen = [(rank, local_rank, ident)]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, local_rank, ident))
        ranks_seen = sorted(gathered)
    ids = [r[2] for r in ranks_seen]
    if len(set(ids)) != len(ids) and not (args.rehearse_on_one_gpu or stub):
        if rank == 0:
            print(f"bench.py: {world} ranks share devices {ids}: pass --rehearse-on-one-gpu for a plumbing rehearsal", file=sys.stderr)
        if dist is not None:
            dist.destroy_process_group()
        sys.exit(3)

    kk = run_cycles(groups, 0, args.warmup)
    # ---- every kernel bracketed by HIP events over a few untimed cycles of the WHOLE fleet on ONE stream (one launch of 256
    # robots per kernel and cycle, nothing overlapping): finds the dominant kernel and gives every kernel's duration alone -
    # the figures earlier rounds reported and the ones a per-kernel roofline can be read against.  (A group's launches cannot
    # stand in for it: a latency-bound kernel takes as long for 64 robots as for 256.)
    pre_steps = max(3, min(6, args.steps))
    one_stream = None
    if G > 1:
        if stub:
            f1, i1, _ = stub_group_fleet(n_inst, n_cells, seed0)
        else:
            f1, i1, _ = build_fleet(nav, n_inst, n_cells, seed0=seed0, device=local_rank, vs=vs, footprint=args.footprint, insts=insts)
        g1 = Group(nav, f1, i1, seed=4242 + rank * 64, stub=stub)
    else:
        f1, g1 = groups[0].fl, groups[0]
    k1 = run_cycles([g1], 0, 3)
    f1.profile_select(None)
    f1.profile(True)
    f1.profile_reset()
    k1 = run_cycles([g1], k1, pre_steps)
    serial = {name: (ms / cnt if cnt else 0.0) for name, (ms, cnt) in f1.profile_read().items()}
    f1.profile(False)
    if G > 1:
        f1.sync()
        t1 = time.perf_counter()
        k1 = run_cycles([g1], k1, 20)
        f1.sync()
        d1 = time.perf_counter() - t1
        t1 = time.perf_counter()
        k1 = run_cycles([g1], k1, 20, restage=False)
        f1.sync()
        d2 = time.perf_counter() - t1
        one_stream = {"ms_per_step": d1 / 20 * 1e3, "ms_per_step_resident_inputs": d2 / 20 * 1e3,
                      "note": "--groups 1: the whole fleet on one stream, every cycle's hand-over serial with its kernels"}
        if not args.no_single and not stub:  # the whole-grid wavefront launch of the WHOLE fleet, alone (whole_grid_wavefronts leg)
            f1.set_bounded_map_grids(False)
            f1.profile_select(["k_bfs"])
            k1 = run_cycles([g1], k1, 3)
            f1.profile(True)
            f1.profile_reset()
            k1 = run_cycles([g1], k1, 5)
            ms, cnt = f1.profile_read()["k_bfs"]
            one_stream["whole_grid_k_bfs_ms"] = ms / max(cnt, 1)
            f1.profile(False)
        f1.close()
    dom = max(serial, key=lambda k: serial[k])
    dom_by = "the longest launch of this run's one-stream pass"
    # The scoring launch and the wavefront launch take about as long as each other since round 4's last build, and which is longer
    # changes from box to box: when the longest launch of the CURRENT profile (profiles/<tag>_pmc_summary.json) is within 15 % of this
    # run's longest, the line describes that one - the same kernel from run to run - and carries the other as roofline_second.
    try:
        _pmc, _ = pmc_summary_from_profiles(current_profile())
        _reg = {k: v.get("kernel_time_ns", 0) for k, v in (_pmc or {}).get("regions", {}).items() if k in serial}
        if _reg:
            prof_dom = max(_reg, key=lambda k: _reg[k])
            if prof_dom != dom and serial[prof_dom] >= 0.85 * serial[dom]:
                dom, dom_by = prof_dom, "the CURRENT profile's longest launch (within 15 % of this run's longest: a tie, broken the same way every run)"
    except Exception:
        pass
    for g in groups:
        g.fl.profile_select([dom])  # each bracketed launch costs its stream a few microseconds: the timed region brackets one kernel
        g.fl.profile(True)
        g.fl.profile_reset()
        g.scored = 0
    barrier()
    t0 = time.perf_counter()
    kk = run_cycles(groups, kk, args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    barrier()
    scored_run = sum(g.scored for g in groups)  # every cycle's own count (the perturbed poses change it from cycle to cycle)
    dom_ms, dom_n = 0.0, 0
    for g in groups:
        ms, cnt = g.fl.profile_read()[dom]
        dom_ms += ms
        dom_n += cnt
        g.fl.profile(False)
        g.fl.profile_select(None)
    boxes = np.concatenate([g.fl.bounds() for g in groups])
    win_cells = int(((boxes[:, 1] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 2])).sum())
    wb = np.concatenate([g.fl.wavefront_boxes() for g in groups]).astype(np.int64)
    reg_rows = np.minimum(wb[:, 3] + 2, n_cells - 1) - np.maximum(wb[:, 2] - 2, 0) + 1
    reg_words = np.minimum(wb[:, 1] + 2, n_cells - 1) // 32 - np.maximum(wb[:, 0] - 2, 0) // 32 + 1
    region_cells = int((reg_rows * reg_words * 32).sum())

    # the only collective: throughput counters over RCCL (navigation_amd/sharding.py, gloo-tested on CPU)
    from navigation_amd.sharding import reduce_counters
    elapsed_max, (total_scored, total_win) = reduce_counters(dist, elapsed, [scored_run, win_cells * args.steps], device="cpu" if host_side else "cuda")
    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, elapsed / args.steps * 1e3)
        per_rank_ms = gathered

    out = None
    if rank == 0:
        ms_per_step = elapsed_max / args.steps * 1e3
        traj_per_s = total_scored / elapsed_max
        scored_step = scored_run / args.steps
        bitmap = n_cells * ((n_cells + 31) // 32) * 4
        alg_bytes = {  # per STEP (all groups' launches of the kernel)
            "k_score": BYTES_PER_TRAJ * scored_step,
            # bounded search: costmap bytes once per robot -> traversable bitmap, the bitmap read twice per grid, 4 B per cell of
            # the region written.  (SURVEY 8d's 5 B x every cell is the WHOLE-grid search's figure: whole_grid_wavefronts leg.)
            "k_bfs": n_inst * (n_cells * n_cells + bitmap + 3 * 2 * bitmap) + 3 * 4 * region_cells,
            "k_inflate": BYTES_PER_INFL_CELL * win_cells,
            "k_merge": BYTES_PER_MERGE_CELL * win_cells,
            "k_obstacle": 0.0, "k_select": 0.0,
        }
        contract = (n_inst, n_cells) == (256, 400) and vs == (32, 32, 16) and args.footprint == "square"
        launch_ms = dom_ms / dom_n if dom_n else 0.0            # one group's launch, measured live over the timed region
        launch_bytes = alg_bytes[dom] / G
        achieved = launch_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        achieved_serial = alg_bytes[dom] / (serial[dom] * 1e-3) / 1e9 if serial[dom] > 0 else 0.0
        cur = current_profile() if contract else None
        traffic, traffic_src = hbm_traffic_from_profiles(dom, cur)
        pmc, pmc_src = pmc_summary_from_profiles(cur)
        # roofline.frac: the dominant kernel's full-fleet launch ALONE (one stream, nothing overlapping; HIP events inside this run) -
        # the per-kernel figure, comparable from round to round and with profiles/<tag>_kernel_stats.csv.  in_schedule: the same
        # kernel's launches inside the timed region, where a launch covers 1 / G of the fleet and shares the GPU with the other
        # groups' kernels.
        roof = {"bound": "hbm", "kernel": dom, "dominant_by": dom_by, "achieved": achieved_serial, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_serial / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "achieved_is": "ALGORITHMIC bytes (SURVEY 8d: 1000 B per scored trajectory) of one launch over the whole fleet / its duration alone on the GPU "
                               f"(HIP events on the launch's stream, {pre_steps} untimed cycles on one stream inside this run).  A figure of merit, not HBM "
                               "utilisation: the window, its screens and the heading tables live in LDS, and a fraction near or above 1 says the "
                               "kernel does not move those bytes.  The ceiling it is graded against is frac_issue",
                "algorithmic_bytes_per_launch": alg_bytes[dom], "avg_launch_ms": serial[dom],
                "in_schedule": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "avg_launch_ms": launch_ms, "launches_timed": dom_n,
                                "algorithmic_bytes_per_launch": launch_bytes,
                                "note": f"the launches of the timed region: each covers 1 / {G} of the fleet and runs beside the other groups' wavefront / scoring kernels"},
                "frac_vs_measured_copy_peak_6290": achieved_serial / 6290.0}
        if cur:
            roof["profile"] = {"tag": cur.get("tag"), "git_head": cur.get("git_head"), "sources_sha16": cur.get("sources_sha16"),
                               "extra_flags": cur.get("extra_flags"), "running_sources_sha16": cur.get("running_sources_sha16"),
                               "matches_running_build": cur.get("matches_running_build"),
                               "note": "profiles/CURRENT.json: the rocprofv3 profile traffic / issue are quoted from; false = the library's sources have changed since it was taken"}
        if traffic and serial[dom] > 0:
            roof["hbm_measured"] = {"GBps": traffic / (serial[dom] * 1e-3) / 1e9, "frac_of_peak": traffic / (serial[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "bytes_per_launch": traffic, "note": "PMC FETCH_SIZE x2 + WRITE_SIZE (gfx950 correction), separate passes, --groups 1"}
        if pmc and dom in pmc.get("regions", {}):
            v = pmc["regions"][dom]
            roof["bound"] = v.get("bound", "latency")
            roof["frac_issue"] = v.get("cu_issue_util", v.get("valu_issue_util"))
            roof["issue"] = {"cu_issue_util": v.get("cu_issue_util"), "valu_util": v.get("valu_issue_util"), "salu_util": v.get("salu_util"), "wait_any": v.get("wait_any"),
                             "wait_inst_any": v.get("wait_inst_any"), "active": v.get("active"), "lds_bank_conflict": v.get("lds_bank_conflict"),
                             "insts_valu": v.get("insts_valu"), "insts_salu": v.get("insts_salu"), "insts_lds": v.get("insts_lds"), "kernel_time_ns": v.get("kernel_time_ns"),
                             "t_issue_ns": pmc.get("t_issue_ns"), "source": pmc_src,
                             "note": "measured issue roofline of the full-fleet launch (--groups 1), tools/pmc_summary.py: cu_issue_util = (VALU + SALU + LDS + memory "
                                     "wave-instructions) x the box's own issue time per instruction and CU / (256 CUs x kernel time) - a CU issues about as many "
                                     "instructions per clock whatever their kind (tools/microbench/valu_rate); valu_util / salu_util are the two pipes alone"}
        # The runner-up, when it is within 15 % of the dominant launch (since round 4's last build the scoring launch and the wavefront
        # launch take about as long as each other, and which of them is "dominant" changes from box to box): the same per-kernel figures,
        # alone on the GPU, so that the line carries both whichever came first.
        second = None
        others = sorted((k for k in serial if k != dom and alg_bytes.get(k, 0) > 0), key=lambda k: -serial[k])
        if others and serial[others[0]] >= 0.85 * serial[dom]:  # (it may be the longer of the two: see dominant_by)
            k2 = others[0]
            a2 = alg_bytes[k2] / (serial[k2] * 1e-3) / 1e9
            t2, t2_src = hbm_traffic_from_profiles(k2, cur)
            second = {"bound": "hbm", "kernel": k2, "achieved": a2, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a2 / HBM_PEAK_GBS, "traffic": t2,
                      "traffic_source": t2_src, "algorithmic_bytes_per_launch": alg_bytes[k2], "avg_launch_ms": serial[k2],
                      "note": "the second-longest launch of the step, alone on the GPU like roofline.frac (no in_schedule bracket: one kernel is bracketed per run)"}
            if pmc and k2 in pmc.get("regions", {}):
                v2 = pmc["regions"][k2]
                second["bound"] = v2.get("bound", "latency")
                second["frac_issue"] = v2.get("cu_issue_util", v2.get("valu_issue_util"))
                second["issue"] = {"cu_issue_util": v2.get("cu_issue_util"), "wait_any": v2.get("wait_any"), "active": v2.get("active"),
                                   "lds_bank_conflict": v2.get("lds_bank_conflict"), "insts_valu": v2.get("insts_valu"), "insts_salu": v2.get("insts_salu"),
                                   "insts_lds": v2.get("insts_lds"), "kernel_time_ns": v2.get("kernel_time_ns"), "source": pmc_src}
        costmap_ms = serial.get("k_obstacle", 0) + serial.get("k_merge", 0) + serial.get("k_inflate", 0)
        out = {
            "metric": "scored trajectories/sec (whole node) + costmap inflation cells/sec, 400x400 map",
            "value": traj_per_s, "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.total_instances else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"configs[2]: {n_inst} batched robot instances per MI355X, {n_cells}x{n_cells} costmaps + inflation, 32x32x16 "
                                   f"velocity samples, 20 sim steps, a NEW LaserScan (720 beams, 3 moving discs) every cycle; "
                                   + (f"{args.total_instances} instances in all, split over the ranks (strong scaling)" if args.total_instances
                                      else f"N GPUs = N x {n_inst} instances (N=8 is configs[3])"),
                       "instances_per_gpu": n_inst, "groups_per_gpu": G, "cycles_in_flight_per_group": depth, "costmap": f"{n_cells}x{n_cells}@0.05", "vsamples": "x".join(str(v) for v in vs),
                       "sim_steps": T_STEPS, "critics": "oscillation+obstacle+goal_front+alignment+path+goal",
                       "parallelism": f"fleet-shard x{world}, {G} stream groups per GPU x {depth} cycle(s) in flight each",
                       "env": {"GPU_FORCE_BLIT_COPY_SIZE": os.environ.get("GPU_FORCE_BLIT_COPY_SIZE")}},
            "value_is": "SURVEY 8(d)'s protocol: every timed cycle hands over its own new scan cloud, pose / velocity and plan from host memory "
                        f"(H2D, {sum(g.h2d_bytes for g in groups)} B per step) and reads its results back (D2H); costmaps resident.  The transfers of one "
                        "group hide behind the other groups' kernels; resident_inputs is the same loop without them",
            "poses": "per-cycle N(0, 2 cm / 0.05 rad) around the base pose, seeded",
            "per_instance_trajectories_per_s": traj_per_s / (n_inst * world),
            "inflation_cells_per_s": (total_win / args.steps / world) / (costmap_ms * 1e-3) * world if costmap_ms > 0 else None,
            "inflation_cells_per_s_is": "SURVEY 8(d) M2: update-window cells per second of obstacle-merge + inflation time (k_obstacle + k_merge + k_inflate, whole-fleet launches alone)",
            "inflation_window_cells_per_step": total_win / args.steps,
            "trajectories_per_step": total_scored / args.steps,
            "kernel_ms": {k: round(serial[k], 4) for k in serial},
            "kernel_ms_source": f"HIP events, {pre_steps} untimed cycles of the whole fleet on ONE stream (one launch of {n_inst} robots per kernel and cycle, nothing "
                                f"overlapping); in the timed region the {G} groups' launches run side by side: sum {sum(serial.values()):.3f} ms vs step {ms_per_step:.3f} ms",
            "roofline": roof,
            "roofline_second": second,
            "ranks_seen": [{"rank": r[0], "local_rank": r[1], "device": r[2]} for r in ranks_seen],
            "per_rank_ms_per_step": {"min": min(per_rank_ms), "max": max(per_rank_ms), "all": [round(v, 4) for v in per_rank_ms]},
        }
        if args.rehearse_on_one_gpu or stub:
            out["rehearsal"] = "all ranks on one device / host-only stand-in: a plumbing check, not a scaling figure"
        per_kernel = {}
        for k in serial:
            if serial[k] > 0 and alg_bytes.get(k, 0) > 0:
                gbs = alg_bytes[k] / (serial[k] * 1e-3) / 1e9
                tk, _ = hbm_traffic_from_profiles(k, cur)
                per_kernel[k] = {"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "avg_launch_ms_alone": serial[k], "traffic": tk,
                                 "algorithmic_bytes_per_launch": alg_bytes[k]}
                if pmc and k in pmc.get("regions", {}):
                    per_kernel[k]["frac_issue"] = pmc["regions"][k].get("cu_issue_util", pmc["regions"][k].get("valu_issue_util"))
                    per_kernel[k]["bound"] = pmc["regions"][k].get("bound")
        out["roofline_all"] = per_kernel
        step_bytes = sum(alg_bytes.values())
        out["roofline_step"] = {"algorithmic_bytes_per_step": step_bytes, "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9,
                                "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s"}
    # ---- extra legs on rank 0 at N=1 only
    if rank == 0 and world == 1 and not stub:
        # PCIe-inclusive cycle times of the same loop: the host's view of one step (all groups queued, results of the previous one read)
        import gc
        gc.collect()
        gc.disable()
        cyc = []
        for _ in range(300):
            tc = time.perf_counter()
            for g in groups:
                g.cycle(kk)
            kk += 1
            cyc.append(time.perf_counter() - tc)
        for g in groups:
            g.collect()
        gc.enable()
        cyc.sort()
        out["pcie_inclusive"] = {"ms_per_step": sum(cyc) / len(cyc) * 1e3, "cycle_ms_median": cyc[len(cyc) // 2] * 1e3,
                                 "cycle_ms_p99": cyc[min(len(cyc) - 1, int(0.99 * len(cyc)))] * 1e3, "cycle_ms_max": cyc[-1] * 1e3, "cycle_ms_second_max": cyc[-2] * 1e3,
                                 "max_note": "with the HIP runtime's default copy path (SDMA) one hipMemcpyAsync in ~2 500 stalls ~8 ms inside the runtime, about "
                                             "every 150 fleet cycles; this process runs with GPU_FORCE_BLIT_COPY_SIZE (config.env): copies up to that many KiB "
                                             "are kernels, and the stall is gone (tools/probe_cycle_outliers.py)",
                                 "cycles": len(cyc),
                                 "h2d_bytes_per_step": sum(g.h2d_bytes for g in groups), "d2h_bytes_per_step": n_inst * 72,
                                 "note": "= the protocol of `value` (the timed region), host-side time per step over 300 more cycles; caller buffers are pageable, "
                                         "the library stages them through pinned mirrors"}
        # the same loop with scans and plans resident (only pose + velocity, 24 B per robot, staged per cycle)
        for g in groups:
            g.scored = 0
        sync_all()
        t1 = time.perf_counter()
        kk = run_cycles(groups, kk, args.steps, restage=False)
        sync_all()
        dt = time.perf_counter() - t1
        out["resident_inputs"] = {"ms_per_step": dt / args.steps * 1e3, "trajectories_per_s": sum(g.scored for g in groups) / dt,
                                  "note": "scans and plans stay in HBM, every cycle stages its pose and velocity only"}
        out["value_over_resident"] = out["resident_inputs"]["ms_per_step"] / ms_per_step
        # latency of one control cycle of the whole fleet, nothing else in flight (median / p99)
        lat = []
        for _ in range(max(60, args.steps)):
            tc = time.perf_counter()
            for g in groups:
                g.cycle(kk)
            for g in groups:
                g.collect()
            kk += 1
            lat.append((time.perf_counter() - tc) * 1e3)
        lat.sort()
        out["cycle_latency"] = {"cycles": len(lat), "ms_median": lat[len(lat) // 2], "ms_p99": lat[min(len(lat) - 1, int(0.99 * len(lat)))],
                                "ms_min": lat[0], "ms_max": lat[-1],
                                "note": "hand over scan + pose + plan -> updateMap -> findBestPath -> results on the host, synchronised every cycle"}
        if one_stream:
            out["one_stream"] = one_stream
        # the same latency with scans and plans resident (pose + velocity staged only): round 2's cycle_latency protocol
        lat = []
        for _ in range(max(60, args.steps)):
            tc = time.perf_counter()
            for g in groups:
                g.cycle(kk, restage=False)
            for g in groups:
                g.collect()
            kk += 1
            lat.append((time.perf_counter() - tc) * 1e3)
        lat.sort()
        out["cycle_latency_resident_inputs"] = {"cycles": len(lat), "ms_median": lat[len(lat) // 2], "ms_p99": lat[min(len(lat) - 1, int(0.99 * len(lat)))]}
        # the same step with every MapGrid wavefront run over the whole costmap, as the reference does (the default stops a
        # search once the box its robot's samples can reach is settled; planner results are identical, tests/test_gpu_parity.py)
        if not args.no_single:
            lv_bounded = np.concatenate([g.fl.wavefront_levels() for g in groups]).mean(axis=0)
            for g in groups:
                g.fl.set_bounded_map_grids(False)
                g.fl.profile_select(["k_bfs"])
            kk = run_cycles(groups, kk, 3)
            for g in groups:
                g.fl.profile(True)
                g.fl.profile_reset()
            k2 = 10
            wg_ms = 0.0
            for _ in range(k2):  # one group at a time: the wavefront kernel alone on the GPU
                for g in groups:
                    g.cycle(kk)
                    g.collect()
                kk += 1
            for g in groups:
                ms, cnt = g.fl.profile_read()["k_bfs"]
                wg_ms += ms / max(cnt, 1)
                g.fl.profile(False)
                g.fl.profile_select(None)
            sync_all()
            t1 = time.perf_counter()
            kk = run_cycles(groups, kk, 20)
            sync_all()
            dt = time.perf_counter() - t1
            lv_whole = np.concatenate([g.fl.wavefront_levels() for g in groups]).mean(axis=0)
            wg_bytes = BYTES_PER_BFS_CELL * 3 * n_cells * n_cells * n_inst
            # (a latency-bound launch takes as long for one group's 64 robots as for the fleet's 256: the per-launch figure is
            # the one-stream fleet's, measured before the timed region; the groups' own launches are kept beside it)
            wg_one = (one_stream or {}).pop("whole_grid_k_bfs_ms", None) if G > 1 else wg_ms
            wg_ref = wg_one if wg_one else wg_ms
            out["whole_grid_wavefronts"] = {"ms_per_step": dt / 20 * 1e3, "k_bfs_ms": wg_ref, "k_bfs_ms_group_launch": wg_ms / G,
                                            "levels_path_goal_front": [float(v) for v in lv_whole],
                                            "roofline": {"algorithmic_bytes_per_step": wg_bytes, "achieved": wg_bytes / (wg_ref * 1e-3) / 1e9,
                                                         "frac": wg_bytes / (wg_ref * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s",
                                                         "note": "SURVEY 8d: 5 B per cell and grid; k_bfs_ms = ONE launch over the whole fleet, alone on the GPU "
                                                                 "(k_bfs_ms_group_launch: one group's launch, alone; ms_per_step: the groups overlapped)"}}
            out["bounded_wavefronts"] = {"enabled": True, "levels_path_goal_front": [float(v) for v in lv_bounded],
                                         "region_cells_per_robot": region_cells / n_inst,
                                         "note": "value / ms_per_step are measured with bounded wavefronts (library default)"}
            for g in groups:
                g.fl.set_bounded_map_grids(True)
            kk = run_cycles(groups, kk, 1)
        # full-window inflation throughput (the BASELINE.md probe shape)
        full_ms, reps = 0.0, 10
        t1 = time.perf_counter()
        for g in groups:
            raw = np.stack([i["cells"] for i in g.fl._bench_insts])
            g.fl.upload(N.GRID_MASTER, raw)
            full = [[0, 0, n_cells, n_cells]] * g.n
            g.fl.inflate(boxes=full)
            g.fl.sync()
            g.fl.profile(True)
            g.fl.profile_reset()
            for _ in range(reps):
                g.fl.inflate(boxes=full)
            g.fl.sync()
            pk = g.fl.profile_read()["k_inflate"]
            g.fl.profile(False)
            full_ms += pk[0] / max(pk[1], 1)
        out["inflation_full_window"] = {"cells_per_s": n_inst * n_cells * n_cells / (full_ms * 1e-3), "kernel_ms": full_ms,
                                        "achieved_GBps": BYTES_PER_INFL_CELL * n_inst * n_cells * n_cells / (full_ms * 1e-3) / 1e9}
        masters = groups[0].fl.master(0, min(groups[0].n, 32))
        if not args.no_single:
            out["inflation_reference_order"] = inflation_reference_order_leg(nav, insts, n_cells, local_rank)
            out["navfn_global_plans"] = navfn_leg(nav, groups[0].fl, min(groups[0].n, 32), local_rank)
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
            out["legacy_trajectory_planner"] = legacy_leg(nav, insts, n_cells, masters, not args.no_cpu_baseline, local_rank)
            out["configs4_one_gpu_share"] = configs4_leg(nav, local_rank)
        if not args.no_cpu_baseline:
            ns = min(len(masters), 32)
            out["cpu_baseline"] = cpu_baseline(insts[:ns], cfg, n_cells, masters[:ns])
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    for g in groups:
        g.fl.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
