#!/bin/bash
# Every workload's rocprofv3 passes in one GPU session: tools/profile_all.sh <tag>   (then tools/collect_profiles.sh <tag> <round>)
tag=$1
set -e
bash tools/profile_bench.sh prof_${tag}
bash tools/profile_bench.sh prof_${tag}_near --rule nearest
bash tools/profile_bench.sh prof_${tag}_mesh1m --workload mesh1m_4k_256spp --rule nearest
bash tools/profile_bench.sh prof_${tag}_mesh1m_first --workload mesh1m_4k_16spp --rule first_index
bash tools/profile_bench.sh prof_${tag}_scene3 --workload scene3_720p_100spp
bash tools/profile_bench.sh prof_${tag}_scene3ext --workload scene3_extended_4k_16spp
bash tools/profile_bench.sh prof_${tag}_scene3ext1024 --workload scene3_extended_4k_1024spp
