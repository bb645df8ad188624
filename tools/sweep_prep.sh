#!/bin/bash
# Exploration (run on the GPU box): threads of the image-building workgroup, timed in the default stream-group schedule.
cd "$GRAFT_REPO_ROOT/navigation_amd/csrc"
run() {
  touch planner_kernels.hip
  make -s EXTRA="$2" 2>/dev/null || { echo "$1: build failed"; return; }
  for i in 1 2; do (cd ../.. && timeout -k 10 200 python bench.py --no-cpu-baseline --no-single --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step', round(d['ms_per_step'],4), 'alone: bfs', d['kernel_ms']['k_bfs'], 'score', d['kernel_ms']['k_score'])"); done
}
run prep-512 "-DNAVGPU_SCORE_PREP_THREADS=512"
run prep-256 "-DNAVGPU_SCORE_PREP_THREADS=256"
run prep-128 "-DNAVGPU_SCORE_PREP_THREADS=128"
run prep-512 "-DNAVGPU_SCORE_PREP_THREADS=512"
run prep-128 "-DNAVGPU_SCORE_PREP_THREADS=128"
run prep-256 "-DNAVGPU_SCORE_PREP_THREADS=256"
