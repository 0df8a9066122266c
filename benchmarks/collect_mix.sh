#!/bin/bash
# instruction mix of the trace kernel: two counter passes into gpurun_out/<tag>/
set -u
tag=${1:-mix}; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cmd=(python3 bench.py --steps 3 --warmup 1 --serial --no-cpu-baseline --no-linear-scan-leg --no-live-pmc "$@")
run() {
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$out/$name" -o "$name" -- "${cmd[@]}" >> "$out/log.txt" 2>&1
  echo "$name rc=$?" | tee -a "$out/log.txt"
}
run pmc_mix1 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 || exit 1
run pmc_mix2 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU || exit 1
