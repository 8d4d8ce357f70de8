#!/bin/bash
# Whole-step throughput against the number of path slots per batch, for both pipelines.
# usage: tools/sweep_batch.sh "<bench args>" <log2 sizes...>
args=$1; shift
for lg in "$@"; do
  for pipe in split fused; do
    out=$(RTMI_PIPELINE=$pipe timeout -k 10 300 python3 bench.py $args --no-cpu-baseline --sustain 0 --batch-paths $((1 << lg)) 2>&1 | tail -1) || { echo "$pipe 2^$lg failed: $out"; exit 1; }
    python3 - "$pipe" "$lg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
print("%-5s 2^%s: %8.1f Mseg/s  %7.2f ms/step  launches %d" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["launches"]))
PY
  done
done
