"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) of tools/prof3d.py into the HBM
traffic record bench.py reports as roofline.traffic.  Units/corrections as MI355X_MICROARCH.md prescribes: the counters
are in KB; FETCH_SIZE is doubled on gfx950 for 16-B/lane streaming reads.

    python tools/pmc_traffic.py <fetch.csv> <write.csv> <n_profiled_steps> <scenes> <algorithmic_bytes_per_step> [git hash] > profiles/rNN/traffic_3d.json

(run it in the tree the counters were collected on: the record carries a fingerprint of csrc/{spconv,osconv,ostable}.hip)
"""
import collections
import csv
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINE_SOURCES = ("spconv.hip", "osconv.hip", "ostable.hip")


def engine_sources_sha256():
    """Fingerprint of the sparse-conv engine sources the counters were taken on; bench.py refuses a record whose fingerprint
    differs from the tree it runs in (a kernel change that moves traffic must come with a new measurement)."""
    h = hashlib.sha256()
    for f in ENGINE_SOURCES:
        h.update(open(os.path.join(ROOT, "mm2d3d_amd", "csrc", f), "rb").read())
    return h.hexdigest()

ENGINE = ("k_gather_gemm", "k_csr_reduce", "k_dw_direct", "k_dw_tr16", "k_dw_reduce", "k_pack_frag", "k_rows_narrow", "k_generic", "k_osconv", "k_os_pack")


def load(path, counter):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"[<(].*", "", name)
        acc[name] += float(r["Counter_Value"])
    return acc


if __name__ != "__main__":
    raise SystemExit  # imported for engine_sources_sha256 only (never reached: bench.py re-implements the three lines)
fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps, scenes, algo = int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
per = {}
total = 0.0
for k in sorted(set(fetch) | set(write)):
    if not k.startswith(ENGINE):
        continue
    f, w = 2.0 * fetch.get(k, 0.0) * 1024 / steps, write.get(k, 0.0) * 1024 / steps
    per[k] = {"fetch_x2_MB": round(f / 1e6, 1), "write_MB": round(w / 1e6, 1)}
    total += f + w
print(json.dumps({
    "what": f"HBM traffic of the sparse-conv engine kernels, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
            f"tools/prof3d.py --scenes {scenes}: 3D branch fwd+bwd on {scenes} NuScenes-shaped scenes = the joint [source|target] pass of one step",
    "units": "counter values are KB (MI355X_MICROARCH.md HBM section); FETCH_SIZE doubled (gfx950 reports 1/2 of 16-B/lane streaming reads)",
    "per_step_MB": per,
    "bytes_per_step": round(total),
    "algorithmic_bytes_per_step": algo,
    "traffic_over_algorithmic": round(total / algo, 3),
    "engine_sources_sha256": engine_sources_sha256(),
    "git": (subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None)
           if len(sys.argv) < 7 else sys.argv[6],
}, indent=1))
