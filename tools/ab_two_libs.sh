#!/bin/bash
# A/B of two builds of libnavgpu.so on ONE box:  bash tools/ab_two_libs.sh old.so new.so [rounds]
# (kernel times alone and the default stream-group step, alternating)
cd "$GRAFT_REPO_ROOT"
old=$1; new=$2; rounds=${3:-3}
for i in $(seq $rounds); do
  for v in old new; do
    eval lib=\$$v
    cp "$lib" navigation_amd/libnavgpu.so
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-single --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms_per_step', round(d['ms_per_step'],4), 'alone: bfs', d['kernel_ms']['k_bfs'], 'score', d['kernel_ms']['k_score'], 'one_stream', round(d.get('one_stream',{}).get('ms_per_step',0),4))"
  done
done
cp "$new" navigation_amd/libnavgpu.so
