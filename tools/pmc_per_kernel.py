#!/usr/bin/env python3
"""Mean per dispatch of every counter of the rocprofv3 --pmc passes of tools/profile_bench.sh, per kernel, over every
dispatch of the run (the raw material behind profiles/pmc_summary.json). usage: tools/pmc_per_kernel.py <profile dir>"""
import collections, csv, glob, sys
prof = sys.argv[1]
print("# rocprofv3 --pmc passes (tools/profile_bench.sh), every dispatch of the run, per kernel and counter: mean per dispatch")
for pas in ("pmc_sq", "pmc_mix", "pmc_fetch", "pmc_write"):
    files = glob.glob("%s/%s/**/*counter_collection.csv" % (prof, pas), recursive=True)
    if not files:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].replace("rtmi::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        per[k][r["Counter_Name"]][r["Dispatch_Id"]] = per[k][r["Counter_Name"]].get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    print("## pass %s" % pas)
    for k in sorted(per):
        if "at::" in k or "rocclr" in k:
            continue
        print("%-46s dispatches %4d  " % (k[:46], max(len(v) for v in per[k].values())) +
              "  ".join("%s=%.4g" % (c, sum(v.values()) / len(v)) for c, v in sorted(per[k].items())))
