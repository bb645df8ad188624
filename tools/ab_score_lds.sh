#!/bin/bash
# A/B on the GPU box: k_score_tab's LDS budget / waves per SIMD (rebuilds planner_kernels.o per variant)
cd "$GRAFT_REPO_ROOT"
for v in "26 6" "14 6" "14 8" "12 8" "20 7"; do
  set -- $v
  touch navigation_amd/csrc/planner_kernels.hip navigation_amd/csrc/navgpu_host.cpp
  make -s -C navigation_amd/csrc EXTRA="-DNAVGPU_SCORE_TAB_LDS_KB=$1 -DNAVGPU_SCORE_TAB_WAVES=$2" 2>&1 | grep -E "error|Stop"
  python bench.py --no-cpu-baseline --no-single --groups 1 --steps 30 > gpurun_out/ab_$1_$2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$1_$2.json"))
print("LDS_KB $1 WAVES $2: k_score %.4f ms, step %.4f ms" % (d["kernel_ms"]["k_score"], d["ms_per_step"]), flush=True)
PY
done
