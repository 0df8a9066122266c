#!/bin/bash
# LDS bank-conflict attribution: one rocprofv3 counter pass per library (the product build and the RT_EXP 51..58
# builds of benchmarks/build_exp.sh, each of which issues one kind of LDS access twice) and workload, into
# gpurun_out/<tag>/<workload>/<lib>/ ; benchmarks/lds_attribution.py turns the passes into a table.
# usage: benchmarks/lds_attribution.sh <tag> [workload ...]
set -u
tag=${1:-lds}; shift || true
workloads=("$@")
[ ${#workloads[@]} -eq 0 ] && workloads=(final1080 dense1080)
root=$PWD
export TMPDIR=/tmp
for w in "${workloads[@]}"; do
  spp=16; [ "$w" = dense1080 ] && spp=4
  for lib in base 51 52 53 54 55 56 57 58; do
    out=$root/gpurun_out/$tag/$w/$lib
    mkdir -p "$out"
    if [ "$lib" = base ]; then unset RT_AMD_DEVICE_LIB; else
      [ -f "$root/_out/exp/lib_exp$lib.so" ] || continue
      export RT_AMD_DEVICE_LIB=$root/_out/exp/lib_exp$lib.so
    fi
    timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES \
      --output-format csv -d "$out" -o pmc -- python3 "$root/bench.py" --pmc-child --workload "$w" --spp $spp > "$out/log.txt" 2>&1
    rc=$?
    echo "$w $lib rc=$rc"
    [ $rc -eq 0 ] || exit 1
  done
done
unset RT_AMD_DEVICE_LIB
