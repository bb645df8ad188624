#!/bin/bash
# Exploration (run on the GPU box): stream groups x hardware queues (GPU_MAX_HW_QUEUES) for the default bench schedule.
cd "$GRAFT_REPO_ROOT"
for q in 4 8; do for g in 4 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --no-single --no-cpu-baseline --groups $g > gpurun_out/q${q}_g${g}.json 2> gpurun_out/q${q}_g${g}.err
  python - gpurun_out/q${q}_g${g}.json $q $g <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("hwq", sys.argv[2], "groups", sys.argv[3], "ms_per_step %.4f" % d["ms_per_step"], "resident %.4f" % d["resident_inputs"]["ms_per_step"], "lat %.3f" % d["cycle_latency"]["ms_median"], flush=True)
except Exception as e:
    print("hwq", sys.argv[2], "groups", sys.argv[3], "failed", e, flush=True)
PY
done; done
