#!/usr/bin/env python3
"""Turns the rocprofv3 passes of tools/profile_bench.sh into profiles/pmc_summary.json: per workload and kernel, the
per-launch means bench.py's roofline terms are formed from.

  hbm_bytes_per_launch         2 * FETCH_SIZE + WRITE_SIZE. rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies
                               128-byte read requests at 64 bytes, so reads are doubled (WRITE_SIZE needs no correction)
  lane_instructions_per_launch SQ_THREAD_CYCLES_VALU: lanes active, summed over the vector instructions executed
  lane_utilisation             SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  valu_busy                    SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / (1024 SIMDs * cycles), cycles = GRBM_GUI_ACTIVE / 8
                               (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs)
  clock_ghz                    cycles / the dispatch's own duration (End_Timestamp - Start_Timestamp)

Only the launches of the run's last render are used: bench.py ends with a render that brackets every launch with HIP
events (one batch at a time, no overlap between batches), and every render of bench.py ends with one tonemap_kernel.
The launches after the last-but-one tone map are therefore the same population as the live launch times bench.py
divides these counters by.
usage: tools/pmc_summary.py <profile dir> <workload>/<rule> [out.json]"""
import collections, csv, glob, json, sys
from pathlib import Path

prof, key = sys.argv[1], sys.argv[2]
out = Path(sys.argv[3] if len(sys.argv) > 3 else Path(__file__).resolve().parent.parent / "profiles" / "pmc_summary.json")
KERNELS = ("traverse_kernel", "shade_start_kernel", "shade_extended_kernel", "shade_kernel", "generate_camera_kernel", "generate_kernel",
           "bounce_kernel", "tail_kernel", "emit_kernel", "resolve_kernel")
vals = collections.defaultdict(lambda: collections.defaultdict(dict))     # kernel -> counter -> {dispatch: value}
dur = collections.defaultdict(dict)                                        # kernel -> {(pass, dispatch): ns}
for f in glob.glob("%s/pmc_*/**/*counter_collection.csv" % prof, recursive=True):
    pass_name = f[len(prof):].strip("/").split("/")[0]
    rows, tonemaps = [], set()
    for r in csv.DictReader(open(f)):
        if "tonemap_kernel" in r["Kernel_Name"]:
            tonemaps.add(int(r["Dispatch_Id"]))
        k = next((k for k in KERNELS if k in r["Kernel_Name"]), None)
        if k:
            rows.append((k, r))
    ends = sorted(tonemaps)
    last_but_one = ends[-2] if len(ends) > 1 else -1
    for k, r in rows:
        if int(r["Dispatch_Id"]) <= last_but_one:
            continue
        d = (pass_name, r["Dispatch_Id"])
        vals[k][r["Counter_Name"]][d] = vals[k][r["Counter_Name"]].get(d, 0.0) + float(r["Counter_Value"])
        if pass_name == "pmc_sq":
            dur[k][d] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])

def mean(k, c):
    v = vals[k].get(c)
    return sum(v.values()) / len(v) if v else None

res = json.loads(out.read_text()) if out.exists() else {}
res[key] = {}
# hash of raytracing_amd/csrc as it was on the GPU box when the passes ran (tools/profile_bench.sh writes it)
stamp = Path(prof) / "source_hash.txt"
source_hash = stamp.read_text().strip() if stamp.exists() else None
for k in vals:
    e = {"source": prof, "source_hash": source_hash, "launches_profiled": max(len(v) for v in vals[k].values())}
    fetch, write = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
    if fetch is not None and write is not None:
        e["hbm_read_bytes_per_launch"] = 2.0 * 1024.0 * fetch
        e["hbm_write_bytes_per_launch"] = 1024.0 * write
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    hit, miss = mean(k, "TCC_HIT_sum"), mean(k, "TCC_MISS_sum")
    if hit is not None and miss is not None and hit + miss > 0:
        e["l2_hit_rate"] = hit / (hit + miss)
    thr, act, insts, gui = mean(k, "SQ_THREAD_CYCLES_VALU"), mean(k, "SQ_ACTIVE_INST_VALU"), mean(k, "SQ_INSTS_VALU"), mean(k, "GRBM_GUI_ACTIVE")
    if thr is not None and act:
        e["lane_instructions_per_launch"] = thr
        e["lane_utilisation"] = thr / (64.0 * act)
        e["valu_instructions_per_launch"] = insts
    if act and gui:
        cycles = gui / 8.0
        e["valu_busy"] = act * 4.0 / (1024.0 * cycles)
        if dur[k]:
            ns = sum(dur[k].values()) / len(dur[k])
            e["launch_ms"] = ns * 1e-6
            e["clock_ghz"] = cycles / ns
    for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_SALU",
              "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
        m = mean(k, c)
        if m is not None:
            e[c] = m
    res[key][k] = e
out.write_text(json.dumps(res, indent=1, sort_keys=True) + "\n")
print(json.dumps(res[key], indent=1, sort_keys=True))
