#!/bin/bash
# Exploration (run on the GPU box): k_score_tab compiled for other occupancies, timed in the default stream-group schedule.
cd "$GRAFT_REPO_ROOT/navigation_amd/csrc"
run() {
  touch planner_kernels.hip
  make -s EXTRA="$2" 2>/dev/null || { echo "$1: build failed"; return; }
  (cd ../.. && timeout -k 10 200 python bench.py --no-cpu-baseline --no-single --steps 100 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step', round(d['ms_per_step'],4), 'alone: bfs', d['kernel_ms']['k_bfs'], 'score', d['kernel_ms']['k_score'], 'one_stream', round(d.get('one_stream',{}).get('ms_per_step',0),4))")
}
run base ""
run score-waves-8 "-DNAVGPU_SCORE_TAB_WAVES=8"
run score-waves-5 "-DNAVGPU_SCORE_TAB_WAVES=5"
run score-waves-4 "-DNAVGPU_SCORE_TAB_WAVES=4"
run base-again ""
