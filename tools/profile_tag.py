"""The identity of the build a profile belongs to:  python3 tools/profile_tag.py <tag> [EXTRA flags...]  prints a JSON object with the
tag, the git head (if any), a hash of the library's SOURCES (navigation_amd/csrc + include: what decides the kernels, the same on
every box - the .so itself is rebuilt per box) and the EXTRA compile flags.  tools/collect_profiles.sh writes it beside the
summaries; profiles/CURRENT.json names the profile bench.py quotes, and bench.py flags it when the sources have changed since."""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha16():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "navigation_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "navigation_amd", "csrc", "*.cpp")) +
                   glob.glob(os.path.join(ROOT, "navigation_amd", "csrc", "*.h")) + [os.path.join(ROOT, "navigation_amd", "csrc", "Makefile"),
                                                                                       os.path.join(ROOT, "include", "navgpu.h")])
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "untagged"
    print(json.dumps({"tag": tag, "git_head": git_head(), "sources_sha16": sources_sha16(), "extra_flags": " ".join(sys.argv[2:]),
                      "files": [f"{tag}_kernel_stats.csv", f"{tag}_pmc_summary.json", f"{tag}_hbm_traffic.json"]}))
