#!/bin/bash
# After `bash tools/collect_profiles.sh <tag>` on the GPU box (its outputs come back under gpurun_out/profiles_<tag>/): copy the summaries
# into profiles/, make the tag CURRENT (profiles/CURRENT.json, with this checkout's git head) and drop the previous tag's copies:
#   bash tools/install_profile.sh <tag> [previous tag to remove]
set -e
cd "$(dirname "$0")/.."
tag=$1; prev=$2
for f in bench_groups_under_rocprof.json bench_under_rocprof.json hbm_traffic.json kernel_stats.csv kernel_stats_groups.csv overlap_groups.txt pmc_summary.json timeline_gaps.txt valu_rate_microbench.txt; do
  cp gpurun_out/profiles_$tag/${tag}_$f profiles/
done
if [ -n "$prev" ]; then
  for f in profiles/${prev}_*; do
    n=profiles/${tag}_${f#profiles/${prev}_}
    [ -e "$n" ] || git mv -q "$f" "$n" 2>/dev/null || mv "$f" "$n"   # (bench lines, rehearsal: re-recorded after this)
  done
  git rm -q --cached profiles/${prev}_* 2>/dev/null || true
  rm -f profiles/${prev}_*
fi
python3 - <<PY
import json, subprocess
d = json.load(open("gpurun_out/profiles_$tag/${tag}_profile_tag.json"))
d["git_head"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
d["note"] = "collected on the GPU box (no .git there): git_head filled in when the summaries were copied into profiles/; the source hash is what ties the profile to a build"
json.dump(d, open("profiles/CURRENT.json", "w"), indent=1)
import sys; sys.path.insert(0, ".")
import bench
print("CURRENT =", d["tag"], "matches the running build:", bench.current_profile()["matches_running_build"])
PY
