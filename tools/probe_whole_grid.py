"""The contract fleet (256 robots, 400 x 400) with every MapGrid wavefront run over the whole costmap, one stream:
python3 tools/probe_whole_grid.py [cycles]   (under rocprofv3 --pmc WRITE_SIZE: the HBM write traffic of the launch)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import navigation_amd as nav  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
fl, insts, cfg = bench.build_fleet(nav, 256, 400, 0)
g = bench.Group(nav, fl, insts, seed=4242)
fl.set_bounded_map_grids(False)
k = bench.run_cycles([g], 0, 2)
fl.profile_select(["k_bfs"])
fl.profile(True)
fl.profile_reset()
k = bench.run_cycles([g], k, n)
ms, cnt = fl.profile_read()["k_bfs"]
print("whole-grid k_bfs ms per launch", ms / max(cnt, 1), "levels", fl.wavefront_levels().mean(axis=0))
fl.close()
