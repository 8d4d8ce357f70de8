#!/bin/bash
# Diagnostic counter passes for one bench.py configuration (what a kernel waits on): binary64 instruction mix and
# pipe activity; wave-cycle breakdown with address/data path (TA / TD / TCP) pressure.
# usage: tools/profile_diag.sh <out_dir_under_gpurun_out> [bench args...]
set -u
out=gpurun_out/$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --sustain 0 "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name exit=$rc"
  return $rc
}
BENCH_ARGS=("$@")
run pmc_f64 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE || exit 1
run pmc_wait --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || exit 1
