"""configs[4] on one GPU, this leg of bench.py only:  python3 tools/probe_configs4.py [robots]
(A/B: NAVGPU_DEBUG_BFS_NO_ROWS=1 puts the wavefronts on k_bfs_global)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import navigation_amd as nav  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = bench.configs4_leg(nav, 0, n_robots=n)
print(json.dumps(out))
