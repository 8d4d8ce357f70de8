#!/usr/bin/env python3
"""Summarises rocprofv3 counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import csv, sys, collections, glob, re
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            m = re.search(r"(\w+_kernel)", name)
            short = m.group(1) if m else name.split("(")[0].split("::")[-1]
            for k in ("bounce_kernel", "traverse_kernel"):
                if k in name and ">" in name:
                    short = name[name.index(k):name.index(">") + 1]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k, cs in sorted(acc.items()):
            if not any(w in k for w in ("bounce", "shade", "traverse", "emit", "resolve", "generate")):
                continue
            print(" ", k, "dispatches", len(next(iter(cs.values()))))
            for c, v in sorted(cs.items()):
                print("     %-22s mean %.4g  sum %.4g" % (c, sum(v) / len(v), sum(v)))
            if "SQ_THREAD_CYCLES_VALU" in cs and "SQ_ACTIVE_INST_VALU" in cs:
                print("     %-22s %.1f %%" % ("lane utilisation", 100.0 * sum(cs["SQ_THREAD_CYCLES_VALU"]) / (64.0 * sum(cs["SQ_ACTIVE_INST_VALU"]))))
