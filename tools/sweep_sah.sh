#!/bin/bash
# Sweep of the SAH triangle cost on the bench line: tools/sweep_sah.sh "<bench args>" cost1 cost2 ...
args=$1; shift
for c in "$@"; do
  out=$(RTMI_SAH_TRIANGLE_COST=$c timeout -k 10 200 python3 -u bench.py --no-cpu-baseline --sustain 0 $args 2>/dev/null | grep '^{' | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.1f Mseg/s  %.2f ms/step  traverse avg %.3f ms  nodes %.2f tris %.2f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["nodes_per_unit"], r["triangles_per_unit"]))')
  echo "cost $c: $out"
done
