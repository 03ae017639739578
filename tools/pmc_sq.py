"""Reduces a rocprofv3 --pmc run (rocpd database) of the SQ counters into per-kernel fractions of the wave cycles.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
              SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d <dir> -- python3 bench.py --steps 3 --warmup 2 --no-extras
    python tools/pmc_sq.py <dir> [fp16|bf16] > profiles/rNN/pmc_sq_step.json

The record is stamped with the sha256 of the 2D kernel sources (csrc/conv2d.hip + csrc/h16.h), the git commit and the 16-bit
storage format of the run: bench.py quotes MFMA-busy figures only from a record whose fingerprint matches the tree it runs in.

SQ_* counters tick in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles (divided by 4 here), so every figure is a fraction of the
wave cycles of the kernel's waves.  MFMA pipe utilisation = mfma_busy_cycles_per_wave_cycle x resident waves per SIMD.
"""
import glob
import hashlib
import json
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SQ_SOURCES = ("conv2d.hip", "h16.h")  # bench.py SQ_SOURCES


def conv2d_sources_sha256():
    h = hashlib.sha256()
    for f in SQ_SOURCES:
        h.update(open(os.path.join(ROOT, "mm2d3d_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
    except OSError:
        return "unknown"


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def main():
    d = sys.argv[1]
    db = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True), key=os.path.getmtime)[-1] if os.path.isdir(d) else d
    con = sqlite3.connect(db)
    rows = con.execute("select k.name, p.counter_name, sum(p.value) from counters_collection p join kernels k on "
                       "p.dispatch_id = k.dispatch_id group by k.name, p.counter_name").fetchall()
    times = {short(n): (c, t) for n, c, t in con.execute("select name, count(*), sum(duration) from kernels group by name")}
    by = {}
    for name, c, v in rows:
        by.setdefault(short(name), {})[c] = float(v)
    out = {}
    for name, c in sorted(by.items(), key=lambda kv: -times.get(kv[0], (0, 0))[1]):
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0 or times[name][1] < 2e6:
            continue
        lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        out[name] = {
            "calls": times[name][0], "time_ms": round(times[name][1] / 1e6, 2),
            "wait_any": round(c.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst_any": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
            "active_inst_any": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
            "mfma_busy_cycles_per_wave_cycle": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4 / wc, 3),
            "wait_inst_lds": round(c.get("SQ_WAIT_INST_LDS", 0) / wc, 3),
            "lds_bank_conflict_per_lds_active": round(c.get("SQ_LDS_BANK_CONFLICT", 0) / lds, 3) if lds > 0 else 0.0,
        }
    git = os.environ.get("MM_GIT_HEAD") or git_head()  # the GPU box has no .git: the collecting script passes the commit in
    print(json.dumps({"what": __doc__.strip().split("\n\n")[1].replace("\n", " "), "conv2d_sources_sha256": conv2d_sources_sha256(),
                      "git": git, "storage": sys.argv[2] if len(sys.argv) > 2 else "fp16", "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
