#!/bin/bash
# Collects the rocprofv3 evidence for one round into gpurun_out/<tag>/ :
#   kernel trace + stats, then hardware counters in separate passes (the SQ
#   block takes 8 counters per pass, TCC 4; FETCH_SIZE uses 3, WRITE_SIZE 2).
# usage: benchmarks/collect_profiles.sh <tag> [bench.py args...]
set -u
tag=${1:-prof}; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cmd=(python3 bench.py --steps 3 --warmup 1 --serial --no-cpu-baseline --no-linear-scan-leg --no-live-pmc "$@")

run() {  # name, then rocprofv3 options
  local name=$1; shift
  echo "== $name" | tee -a "$out/log.txt"
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$out/$name" -o "$name" -- "${cmd[@]}" >> "$out/log.txt" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a "$out/log.txt"
  return $rc
}

run trace --kernel-trace --stats || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS || exit 1
run pmc_sq2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA || exit 1
run pmc_sq3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM || exit 1
run pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
run pmc_salu --pmc SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU || echo 'pmc_salu pass failed (counter not available?)'
run pmc_occ --pmc OccupancyPercent || exit 1
run pmc_valubusy --pmc VALUBusy || exit 1
run pmc_valuutil --pmc VALUUtilization || exit 1
echo done | tee -a "$out/log.txt"
