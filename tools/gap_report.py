"""Idle gaps of the GPU inside one training step, from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): the kernels are merged
over all streams into busy intervals; gaps above --min us are listed with the kernels before and after them.
usage: python tools/gap_report.py trace.csv [--min 30] [--steps 13]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--min", type=float, default=30.0)
ap.add_argument("--tail", type=float, default=0.25, help="fraction of the trace (its end) that is analysed")
a = ap.parse_args()
rows = []
for r in csv.DictReader(open(a.csv)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - (t1 - t0) * a.tail
rows = [r for r in rows if r[0] >= cut]
busy_end, prev = rows[0][1], rows[0][2]
gaps, busy = [], 0
cur_start = rows[0][0]
for s, e, n in rows[1:]:
    if s > busy_end:
        gaps.append((s - busy_end, prev, n))
        busy += busy_end - cur_start
        cur_start = s
    if e > busy_end:
        busy_end, prev = e, n
busy += busy_end - cur_start
span = rows[-1][1] - rows[0][0]
print(f"analysed {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps ({len(rows)} kernels)")
big = sorted(gaps, reverse=True)
print(f"gaps >= {a.min} us: {sum(1 for g in gaps if g[0] >= a.min * 1e3)}, their sum {sum(g[0] for g in gaps if g[0] >= a.min * 1e3) / 1e6:.2f} ms")
for g, p, n in big[:40]:
    if g < a.min * 1e3:
        break
    print(f"{g / 1e3:8.1f} us  after {p}  before {n}")
