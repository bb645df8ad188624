"""Where does k_score_prep_tab spend its time?  Needs an experiment build:
   make -C navigation_amd/csrc clean all EXTRA=-DNAVGPU_PREP_TIMING   (rebuild without EXTRA afterwards)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import navigation_amd as nav
fn = C.CDLL(nav.lib_path()).navgpu_debug_prep_stats
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # stream groups (1: the prep launch alone on the GPU; 4: in the company of the other groups' kernels)
from navigation_amd.sharding import shard_range as split
groups = []
for gi in range(G):
    g0, gn = split(256, gi, G)
    fl, insts, cfg = bench.build_fleet(nav, gn, 400, seed0=g0)
    groups.append(bench.Group(nav, fl, insts, seed=4242 + gi))
kk = bench.run_cycles(groups, 0, 3)
out = (C.c_ulonglong * 16)()
fn(out, 1)
kk = bench.run_cycles(groups, kk, 10)
for g in groups:
    g.fl.sync()
fn(out, 0)
groups_n = G
n = max(out[15], 1)
names = ["stage footprint / samples / window bytes", "raw bits", "dilation", "MapGrid screens", "heading sequences", "trig + rotated footprints",
         "image store", "free distance, scalars, reject bytes"]
tot = 0.0
for i, nm in enumerate(names):
    us = out[i] / n * 10.0 / 1e3  # wall_clock64: 100 MHz
    tot += us
    print("%-44s %7.2f us" % (nm, us))
print("%-44s %7.2f us  (%d workgroups sampled, %d stream group(s))" % ("whole workgroup", tot, n, groups_n))
