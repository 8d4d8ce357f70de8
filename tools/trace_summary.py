#!/usr/bin/env python3
"""Prints per-launch durations of the last batch of bounce kernels from a rocprofv3 kernel trace csv."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "bounce_kernel" in r["Kernel_Name"]]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
last = rows[-per:]
tot = 0
for i, r in enumerate(last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(i, "P" if "<true" in r["Kernel_Name"] else "B", "%.1f us" % d, "vgpr", r["VGPR_Count"], "lds", r["LDS_Block_Size"], "grid", r["Grid_Size_X"])
print("batch total %.1f us" % tot)
