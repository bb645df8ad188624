"""How much do the kernels of different streams really overlap?  python tools/overlap_trace.py <r_kernel_trace.csv>
Prints per kernel name the mean duration, and over the trace's busy span the share of time with 0 / 1 / >= 2 kernels running."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
dur = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("navgpu::", "")
    ev.append((s, 1, k))
    ev.append((e, -1, k))
    dur[k].append(e - s)
ev.sort()
# steady state: from the first k_select on (set-up launches before it), over the span in which k_select keeps recurring
sel = sorted(t for t, d, k in ev if d == 1 and k.startswith("k_select"))
t_lo = sel[len(sel) // 5] if sel else ev[0][0]
t_hi = sel[-1] if sel else ev[-1][0]
ev = [e for e in ev if e[0] <= t_hi]
active, last, share = 0, None, collections.Counter()
pair = collections.Counter()
running = collections.Counter()
for t, d, k in ev:
    if last is not None and t > t_lo:
        dt = t - max(last, t_lo)
        if dt > 0:
            share[min(active, 2)] += dt
            names = tuple(sorted(n.split("<")[0] for n, c in running.items() if c > 0))
            pair[names] += dt
    active += d
    running[k] += d
    last = t
tot = sum(share.values())
print("share of time with 0 / 1 / >=2 kernels running:", {k: round(v / tot, 3) for k, v in sorted(share.items())})
for names, v in pair.most_common(12):
    print(f"  {v / tot:6.3f}  {' + '.join(names) if names else '(idle)'}")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v = v[len(v) // 2:]
    print(f"  {k[:60]:60s} n {len(v):5d} mean {sum(v) / len(v) / 1e3:8.1f} us")
