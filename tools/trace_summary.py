#!/usr/bin/env python3
"""Prints per-launch durations of the last batch of path kernels from a rocprofv3 kernel trace csv."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
names = ("bounce_kernel", "generate_camera_kernel", "generate_kernel", "shade_start_kernel", "shade_extended_kernel", "shade_kernel",
         "traverse_kernel", "emit_kernel", "resolve_kernel")
rows = [r for r in csv.DictReader(open(f)) if any(n in r["Kernel_Name"] for n in names)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one batch = everything up to and including a resolve_kernel
ends = [i for i, r in enumerate(rows) if "resolve_kernel" in r["Kernel_Name"]]
last = rows[ends[-2] + 1:ends[-1] + 1] if len(ends) > 1 else rows
tot = 0
t0 = int(last[0]["Start_Timestamp"])
for i, r in enumerate(last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = next(n for n in names if n in r["Kernel_Name"])
    arg = r["Kernel_Name"][r["Kernel_Name"].index(name) + len(name):].split(">")[0] + ">" if "<" in r["Kernel_Name"] else ""
    print("%2d %-16s %-24s %8.1f us  start %8.1f  vgpr %s lds %s grid %s" % (i, name, arg[:24], d, (int(r["Start_Timestamp"]) - t0) / 1e3, r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size_X"]))
print("batch kernel total %.1f us, span %.1f us" % (tot, (int(last[-1]["End_Timestamp"]) - t0) / 1e3))
