#!/bin/bash
# A/B on the GPU box:  bash tools/ab_variants.sh "name:EXTRA flags" ...   rebuilds libnavgpu.so per variant (the whole library: every
# object sees EXTRA), prints the per-kernel times of a one-stream pass and the 4-group step time, and ends on the default build.
cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; extra=$2
  make -s -C navigation_amd/csrc clean >/dev/null; make -s -j8 -C navigation_amd/csrc EXTRA="$extra" 2>&1 | grep -E "error|Stop"
  python bench.py --no-cpu-baseline --no-single --groups 1 --steps 30 > gpurun_out/ab_$name.1.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-single --steps 50 > gpurun_out/ab_$name.4.json 2>/dev/null
  python - <<PY
import json
a=json.load(open("gpurun_out/ab_$name.1.json")); b=json.load(open("gpurun_out/ab_$name.4.json"))
print("$name [$extra]: alone k_score %.4f k_bfs %.4f ms | 1 stream %.4f | 4 groups %.4f ms per step" % (a["kernel_ms"]["k_score"], a["kernel_ms"]["k_bfs"], a["ms_per_step"], b["ms_per_step"]), flush=True)
PY
}
for v in "$@"; do run "${v%%:*}" "${v#*:}"; done
make -s -C navigation_amd/csrc clean >/dev/null; make -s -j8 -C navigation_amd/csrc 2>&1 | grep -E "error|Stop"
echo "default build restored"
