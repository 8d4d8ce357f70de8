#!/bin/bash
# Collects the rocprofv3 evidence for one bench.py configuration on the GPU box:
#   1. kernel trace + stats, 2-5. PMC passes (each in its own run: SQ issue, SQ instruction mix, TCC fetch, TCC write).
# usage: tools/profile_bench.sh <out_dir_under_gpurun_out> [bench args...]
set -u
out=gpurun_out/$1; shift
mkdir -p "$out"
python3 tools/source_hash.py > "$out/source_hash.txt"    # what was profiled: bench.py compares it with what it times
export TMPDIR=/tmp
run() { # name, rocprof args...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --sustain 0 "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name exit=$rc"
  return $rc
}
BENCH_ARGS=("$@")
run trace --kernel-trace --stats || exit 1
run pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE || exit 1
run pmc_mix --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY || exit 1
run pmc_fetch --pmc FETCH_SIZE TCC_HIT_sum || exit 1
run pmc_write --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum || exit 1
find "$out" -name "*.csv" | head -20
