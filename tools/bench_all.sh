#!/bin/bash
# One line per workload / hit rule: tools/bench_all.sh [extra bench args]
run() { # workload rule
  out=$(timeout -k 10 400 python3 bench.py --no-cpu-baseline --sustain 0 --steps 5 --warmup 2 --workload $1 --rule $2 "${@:3}" 2>/dev/null | grep '^{' | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%9.1f Mseg/s %8.2f ms/step  %s avg %.3f ms  nodes %.2f tris %.2f  bvh %.2f s" % (d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["nodes_per_unit"], r["triangles_per_unit"], d["config"]["bvh_build_s"]))')
  printf "%-32s %-12s %s\n" $1 $2 "$out"
}
run suzanne_1080p_64spp first_index "$@"
run suzanne_1080p_64spp nearest "$@"
run suzanne_specular_1080p_64spp first_index "$@"
run scene1_400x225_100spp first_index "$@"
run scene2_850x480_100spp first_index "$@"
run scene3_720p_100spp first_index "$@"
run scene3_extended_4k_16spp first_index "$@"
run mesh1m_4k_16spp nearest "$@"
run mesh1m_4k_16spp first_index "$@"
