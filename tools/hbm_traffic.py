#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh into profiles/hbm_traffic.json:
HBM bytes per launch of each path kernel. rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE
tallies 128-byte read requests at 64 bytes, so reads are doubled (WRITE_SIZE needs no correction).
usage: tools/hbm_traffic.py <profile dir> <workload name> [out.json]"""
import collections, csv, glob, json, sys
from pathlib import Path

prof, workload = sys.argv[1], sys.argv[2]
out = Path(sys.argv[3] if len(sys.argv) > 3 else Path(__file__).resolve().parent.parent / "profiles" / "hbm_traffic.json")
KERNELS = ("traverse_kernel", "shade_start_kernel", "shade_extended_kernel", "shade_kernel", "generate_camera_kernel", "generate_kernel",
           "bounce_kernel", "emit_kernel", "resolve_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_fetch", "pmc_write"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (prof, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((k for k in KERNELS if k in r["Kernel_Name"]), None)
            if k and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = json.loads(out.read_text()) if out.exists() else {}
res[workload] = {}
for k, c in acc.items():
    n = max(len(c["FETCH_SIZE"]), 1)
    fetch = 2.0 * 1024.0 * sum(c["FETCH_SIZE"]) / n
    write = 1024.0 * sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
    res[workload][k] = {"launches_profiled": n, "read_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                        "bytes_per_launch": fetch + write, "source": prof}
out.write_text(json.dumps(res, indent=1, sort_keys=True) + "\n")
print(json.dumps(res[workload], indent=1))
