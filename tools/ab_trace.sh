#!/bin/bash
# Per-kernel time split of library builds: tools/ab_trace.sh "<bench args>" variant1 variant2 ... ("base" = the real library)
args=$1; shift
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset RTMI_LIB; else export RTMI_LIB=raytracing_amd/librtmi_$v.so; fi
  rm -rf /tmp/abt_$v
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abt_$v -- python3 bench.py --no-cpu-baseline --sustain 0 --steps 3 --warmup 1 $args > /tmp/abt_$v.log 2>&1
  echo "== $v"
  python3 - /tmp/abt_$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    for k in ("traverse_kernel", "shade_start_kernel", "shade_kernel", "generate_camera_kernel", "bounce_kernel", "emit_kernel", "resolve_kernel", "tail_kernel"):
        if k in n:
            tag = n[n.index(k):n.index(">") + 1] if "<" in n and k in ("traverse_kernel", "bounce_kernel") else k
            print("   %-40s calls %4s avg %9.1f us total %9.1f us" % (tag, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
            break
PY
  if [ -n "${AB_LAUNCHES:-}" ]; then python3 tools/trace_summary.py /tmp/abt_$v | head -${AB_LAUNCHES}; fi
done
