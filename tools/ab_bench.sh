#!/bin/bash
# A/B of library builds in one GPU session: tools/ab_bench.sh "<bench args>" variant1 variant2 ... ("base" = the real library)
args=$1; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset RTMI_LIB; else export RTMI_LIB=raytracing_amd/librtmi_$v.so; fi
    out=$(timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --sustain 0 $args 2>/dev/null | grep '^{' | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.1f Mseg/s  %.2f ms/step  %s avg %.3f ms" % (d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"]))')
    echo "round $round $v: $out"
  done
done
