import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [("baseline tables", {}), ("no tables", {"NAVGPU_DEBUG_NO_TABLES": "1"}),
         ("no tables, +40KB lds (3 blocks/CU)", {"NAVGPU_DEBUG_NO_TABLES": "1", "NAVGPU_DEBUG_SCORE_LDS": "40000"}),
         ("no tables, +70KB lds (2 blocks/CU)", {"NAVGPU_DEBUG_NO_TABLES": "1", "NAVGPU_DEBUG_SCORE_LDS": "70000"}),
         ("no tables, no dist gathers", {"NAVGPU_DEBUG_NO_TABLES": "1", "NAVGPU_DEBUG_SCORE": "1"}),
         ("no tables, no footprint walk", {"NAVGPU_DEBUG_NO_TABLES": "1", "NAVGPU_DEBUG_SCORE": "2"}),
         ("no tables, neither", {"NAVGPU_DEBUG_NO_TABLES": "1", "NAVGPU_DEBUG_SCORE": "3"}),
         ("tables, no footprint walk", {"NAVGPU_DEBUG_SCORE": "2"})]
for name, env in cases:
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-single"],
                         env=e, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{name:40s} k_score {d['kernel_ms']['k_score']:.3f} ms  k_bfs {d['kernel_ms']['k_bfs']:.3f}  step {d['ms_per_step']:.3f}")
    except Exception as ex:
        print(name, "FAILED", out.stderr[-300:])
