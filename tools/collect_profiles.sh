#!/bin/bash
# Turns the passes tools/profile_bench.sh left under gpurun_out/prof_<tag>* into the committed summaries:
#   profiles/pmc_summary.json, profiles/<round>_<name>_kernel_stats.csv, _last_batch.txt, _pmc_per_kernel.txt
# usage: tools/collect_profiles.sh <tag> <round>      e.g. tools/collect_profiles.sh r02i r02
set -e
tag=$1; round=$2
rm -f profiles/pmc_summary.json
while read -r suffix name key; do
  [ "$suffix" = "-" ] && suffix=""
  d=gpurun_out/prof_${tag}${suffix}
  [ -d "$d" ] || continue
  python3 tools/pmc_summary.py "$d" "$key" > /dev/null
  cp "$(find "$d/trace" -name '*kernel_stats.csv')" profiles/${round}_${name}_kernel_stats.csv
  python3 tools/trace_summary.py "$d/trace" > profiles/${round}_${name}_last_batch.txt
  python3 tools/pmc_per_kernel.py "$d" > profiles/${round}_${name}_pmc_per_kernel.txt
done <<LIST
- suzanne suzanne_1080p_64spp/first_index
_mesh1m mesh1m mesh1m_4k_256spp/nearest
_mesh1m_first mesh1m_first_index mesh1m_4k_16spp/first_index
_near suzanne_nearest suzanne_1080p_64spp/nearest
_scene3 scene3 scene3_720p_100spp/first_index
_scene3ext scene3_extended_16spp scene3_extended_4k_16spp/first_index
_scene3ext1024 scene3_extended scene3_extended_4k_1024spp/first_index
LIST
ls profiles/${round}_*
