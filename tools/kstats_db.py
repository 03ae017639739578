"""Per-kernel totals of a rocprofv3 --kernel-trace run stored as the tool's SQLite database (its default output format).
    python tools/kstats_db.py <results.db> <steps> [csv_out]
Names: the anonymous-namespace prefix and argument lists are dropped, template arguments kept."""
import collections
import re
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    depth, out = 0, []
    for ch in n:  # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for n, s, e in rows:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    tot = sum(v[1] for v in agg.values())
    busy_span = rows[-1][2] - rows[0][1]
    print(f"# {len(rows)} dispatches, {steps:g} steps: {tot / 1e6 / steps:.3f} ms of kernels and {len(rows) / steps:.0f} launches per step"
          f" (first to last dispatch {busy_span / 1e6:.1f} ms)")
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,MsPerStep,CallsPerStep"]
    for n, (c, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{n}\",{c},{t:.0f},{t / c:.0f},{100 * t / tot:.2f},{mn:.0f},{mx:.0f},{t / 1e6 / steps:.4f},{c / steps:.1f}")
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("\n".join(lines) + "\n")
    for ln in lines[1:61]:
        f = ln.rsplit(",", 8)
        print(f"{float(f[7]):8.3f} ms {float(f[8]):7.1f}/step {float(f[3]) / 1e3:9.1f} us  {f[0][:120]}")


if __name__ == "__main__":
    main()
