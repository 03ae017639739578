"""Summarises a rocprofv3 rocpd database (the default output of ROCm 7.2): per-kernel call count / average duration, and
per-kernel averages of the PMC counters when the run collected any.

    python tools/rocpd_summary.py <results.db> [name-filter] [--csv]
"""
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def main():
    db, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "")
    if os.path.isdir(db):  # the -d directory of a rocprofv3 run: take its newest database
        found = sorted(glob.glob(os.path.join(db, "**", "*.db"), recursive=True), key=os.path.getmtime)
        if not found:
            raise SystemExit(f"no rocpd database under {db}")
        db = found[-1]
    csv = "--csv" in sys.argv
    if "--by-grid" in sys.argv:  # per (kernel, grid size): calls, average / total duration - which layer sizes cost the time
        con = sqlite3.connect(db)
        rows = con.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(duration), sum(duration) from kernels "
                           "group by name, grid_x, grid_y, grid_z order by sum(duration) desc").fetchall()
        tot = sum(r[7] for r in rows) or 1
        for name, gx, gy, gz, wx, n, avg, sm in rows:
            if flt and flt not in name:
                continue
            print(f"{short(name)[:60]:60s} wgs {gx // max(wx, 1):7d}x{gy}x{gz} calls {n:5d} avg {avg / 1e3:9.1f} us total {sm / 1e6:8.2f} ms {100 * sm / tot:5.1f}%")
        return
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration), max(vgpr_count), "
                       "max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    if csv:
        print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, n, avg, mn, mx, sm, vg, ag, lds, gx, wx in rows:
        if flt and flt not in name:
            continue
        if csv:
            print(f'"{short(name)}",{n},{int(sm)},{avg:.1f},{100 * sm / tot:.2f},{mn},{mx}')
        else:
            print(f"{short(name)[:70]:70s} calls {n:5d} avg {avg / 1e3:9.1f} us  min {mn / 1e3:8.1f} max {mx / 1e3:8.1f}  {100 * sm / tot:5.1f}%  "
                  f"vgpr {vg}+{ag} lds {lds} grid {gx}/{wx}")
    try:
        pm = cur.execute("select k.name, p.counter_name, avg(p.value), count(*) from counters_collection p join kernels k on "
                         "p.dispatch_id = k.dispatch_id group by k.name, p.counter_name").fetchall()
    except sqlite3.Error:
        cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
        print("counters_collection columns:", cols)
        pm = []
    by = {}
    for name, c, v, n in pm:
        if flt and flt not in name:
            continue
        by.setdefault(short(name), {})[c] = v
    for name, d in by.items():
        print(name[:90])
        for c, v in sorted(d.items()):
            print(f"    {c:28s} {v:16.1f}")


if __name__ == "__main__":
    main()
