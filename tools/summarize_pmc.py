#!/usr/bin/env python3
"""Summarises rocprofv3 counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import csv, sys, collections, glob
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            short = name.split("(")[0].split("::")[-1] if "bounce_kernel" not in name else name[name.index("bounce_kernel"):name.index(">") + 1]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k, cs in sorted(acc.items()):
            if "bounce" not in k and "resolve" not in k:
                continue
            print(" ", k, "dispatches", len(next(iter(cs.values()))))
            for c, v in sorted(cs.items()):
                print("     %-22s mean %.4g  sum %.4g" % (c, sum(v) / len(v), sum(v)))
