"""GPU idle time inside the timed steps of a rocprofv3 --kernel-trace run (CSV: *_kernel_trace.csv with Start_Timestamp /
End_Timestamp in ns): the union of the busy intervals of ALL queues over the last `steps` steps of the run, the idle total, and the
largest gaps with the kernels before / after them.      python tools/timeline_gaps.py <kernel_trace.csv> <steps> [top]"""
import csv, sys, re
path, steps = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:60]
# step boundaries: the optimiser kernel k_adamw runs twice per step (two optimisers), the last launches of a step
ends = [e for s, e, n, q in rows if "k_adamw" in n]
if len(ends) < 2 * steps + 2:
    print("not enough k_adamw launches", len(ends)); sys.exit(1)
t1 = ends[-1]
t0 = ends[-1 - 2 * steps]
sel = [(s, e, n, q) for s, e, n, q in rows if s >= t0 and e <= t1]
busy, gaps = 0, []
cur_s, cur_e, last_name = sel[0][0], sel[0][1], sel[0][2]
for s, e, n, q in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, short(last_name), short(n)))
        cur_s, cur_e, last_name = s, e, n
    else:
        if e > cur_e:
            cur_e, last_name = e, n
busy += cur_e - cur_s
total = t1 - t0
print(f"{steps} steps: {total / steps / 1e6:.3f} ms per step, busy {busy / steps / 1e6:.3f} ms, idle {(total - busy) / steps / 1e6:.3f} ms ({100 * (total - busy) / total:.1f} %), {len(sel) / steps:.0f} launches per step, {len(gaps) / steps:.0f} gaps per step")
import collections
hist = collections.Counter()
for g, a, b in gaps:
    hist[min(int(g / 1000), 50)] += 1
print("gap histogram (us -> count per step):", {k: round(v / steps, 1) for k, v in sorted(hist.items())})
print("idle by gap size: <3us %.3f ms, 3-10us %.3f ms, >10us %.3f ms per step" % (
    sum(g for g, _, _ in gaps if g < 3000) / steps / 1e6, sum(g for g, _, _ in gaps if 3000 <= g < 10000) / steps / 1e6, sum(g for g, _, _ in gaps if g >= 10000) / steps / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    agg[(a, b)][0] += g; agg[(a, b)][1] += 1
print("largest idle by (kernel before -> kernel after), per step:")
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"  {g / steps / 1e3:8.1f} us  {c / steps:5.1f}x  {a}  ->  {b}")
