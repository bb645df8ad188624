import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [("baseline", {}), ("no planes (16)", {"NAVGPU_DEBUG_BFS": "16"}), ("fixed levels no flag (32)", {"NAVGPU_DEBUG_BFS": "32"}),
         ("no l/r LDS reads (64)", {"NAVGPU_DEBUG_BFS": "64"}), ("16+32", {"NAVGPU_DEBUG_BFS": "48"}), ("16+32+64", {"NAVGPU_DEBUG_BFS": "112"})]
for name, env in cases:
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-single", "--instances", "85"],
                         env=e, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{name:32s} k_bfs {d['kernel_ms']['k_bfs']:.3f} ms")
    except Exception as ex:
        print(name, "FAILED", out.stderr[-300:])
