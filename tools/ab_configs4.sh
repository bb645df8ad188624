#!/bin/bash
# A/B on the GPU box, BASELINE configs[4] on one GPU (64 robots, 1000 x 1000, voxel layer, 5-vertex footprint, 64 x 64 x 32 samples):
#   bash tools/ab_configs4.sh "name:EXTRA flags" ...     (rebuilds the library per variant, ends on the default build)
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  name="${v%%:*}"; extra="${v#*:}"
  make -s -C navigation_amd/csrc clean >/dev/null; make -s -j8 -C navigation_amd/csrc EXTRA="$extra" 2>&1 | grep -E "error|Stop"
  python tools/probe_configs4.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name [$extra]: configs[4] share', round(d['ms_per_step'],4), 'ms per step, kernels', d['kernel_ms'])"
done
make -s -C navigation_amd/csrc clean >/dev/null; make -s -j8 -C navigation_amd/csrc 2>&1 | grep -E "error|Stop"
echo "default build restored"
