#!/bin/bash
# Exploration: rebuild planner_kernels.hip with extra compiler flags and time the bench (run on the GPU box).
cd "$GRAFT_REPO_ROOT/navigation_amd/csrc"
make -s 2>/dev/null
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-unused-value"
run() {
  /opt/rocm/bin/hipcc -x hip $BASE $2 -c planner_kernels.hip -o planner_kernels.o 2>/dev/null || { echo "$1: build failed"; return; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnavgpu.so navgpu_host.o navgpu_local_planner.o navgpu_tp.o costmap_kernels.o planner_kernels.o tp_kernels.o
  (cd ../.. && timeout -k 10 200 python bench.py --no-cpu-baseline --no-single | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), d['kernel_ms']['k_bfs'], d['kernel_ms']['k_score'])")
}
run base ""
run max-ilp "-mllvm -amdgpu-sched-strategy=max-ilp"
run max-memory-clause "-mllvm -amdgpu-sched-strategy=max-memory-clause"
run iterative-minreg "-mllvm -amdgpu-sched-strategy=iterative-minreg"
run iterative-ilp "-mllvm -amdgpu-sched-strategy=iterative-ilp"
run no-postra "-mllvm -enable-post-misched=0"
run O2 "-O2"
