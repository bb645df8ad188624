"""Idle time between the kernels of a step, from a rocprofv3 --kernel-trace CSV (r_kernel_trace.csv):
   python tools/timeline_gaps.py <dir>/r_kernel_trace.csv
Prints, for the steady-state steps (k_obstacle .. k_select), the mean duration of every kernel and the mean gap before it."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("navgpu::", "")) for r in rows))
# steps = runs that start at k_obstacle and end at k_select
steps, cur = [], None
for s, e, k in ev:
    if k.startswith("k_obstacle"):
        cur = []
    if cur is not None:
        cur.append((s, e, k))
        if k.startswith("k_select"):
            steps.append(cur)
            cur = None
steps = [st for st in steps if len(st) == collections.Counter(len(x) for x in steps).most_common(1)[0][0]][2:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
span, busy = [], []
for st in steps:
    for i, (s, e, k) in enumerate(st):
        dur[(i, k)].append(e - s)
        if i:
            gap[(i, k)].append(s - st[i - 1][1])
    span.append(st[-1][1] - st[0][0])
    busy.append(sum(e - s for s, e, _ in st))
print(f"{len(steps)} steps; span {sum(span) / len(span) / 1e3:.1f} us, kernels {sum(busy) / len(busy) / 1e3:.1f} us")
for (i, k) in sorted(dur):
    g = gap.get((i, k))
    d = sorted(dur[(i, k)])
    print(f"  {i:2d} {k[:50]:50s} mean {sum(d) / len(d) / 1e3:7.1f}  median {d[len(d) // 2] / 1e3:7.1f}  min {d[0] / 1e3:7.1f}  max {d[-1] / 1e3:7.1f} us   gap before {(sum(g) / len(g) / 1e3 if g else 0):5.1f} us")
# step-to-step: start of k_obstacle to the next one
starts = [st[0][0] for st in steps]
d = [b - a for a, b in zip(starts, starts[1:]) if b - a < 10 * (sum(span) / len(span))]
if d:
    print(f"step period {sum(d) / len(d) / 1e3:.1f} us over {len(d)} consecutive steps")
