#!/bin/bash
# The bench line of every workload of the results table, CPU baseline legs included, saved under gpurun_out/bench_<tag>/:
#   tools/bench_table.sh <tag>
tag=$1
out=gpurun_out/bench_$tag
mkdir -p $out
run() { # name workload rule steps [extra args]
  local name=$1 w=$2 rule=$3 steps=$4; shift 4
  timeout -k 10 900 python3 bench.py --steps $steps --warmup 2 --workload $w --rule $rule "$@" > $out/$name.json 2> $out/$name.err
  python3 - $out/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    r, c = d["roofline"], d.get("cpu_baseline") or {}
    print("%-44s %9.1f Mseg/s %8.2f ms/step  mesh rays %7.1f M/s  %s avg %.3f ms  alg frac %.3f issue %s hbm %s  cpu %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], d["mesh_rays_mps"], r["kernel"], r["avg_launch_ms"], r["frac"],
        r.get("counter_issue_frac"), r.get("counter_hbm_frac"), c.get("value")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run default suzanne_1080p_64spp first_index 20
run suzanne_1080p_64spp_nearest suzanne_1080p_64spp nearest 20
run suzanne_specular_1080p_64spp_first_index suzanne_specular_1080p_64spp first_index 20
run scene1_400x225_100spp_first_index scene1_400x225_100spp first_index 20
run scene2_850x480_100spp_first_index scene2_850x480_100spp first_index 20
run scene3_720p_100spp_first_index scene3_720p_100spp first_index 20
run mesh1m_256_nearest mesh1m_4k_256spp nearest 5
run mesh1m_256_nearest_gpubvh mesh1m_4k_256spp nearest 5 --gpu-bvh --no-cpu-baseline
run mesh1m_16_first_index mesh1m_4k_16spp first_index 5 --no-cpu-baseline
run scene3ext_1024 scene3_extended_4k_1024spp first_index 2 --no-cpu-baseline
