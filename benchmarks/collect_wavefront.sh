#!/bin/bash
# The material-sorted wavefront variant (RT_KERNEL_WAVEFRONT) beside the megakernel on one workload: bench lines of
# both, kernel trace + stats and one counter pass (thread efficiency per stage) of the wavefront form, into
# gpurun_out/<tag>/ .   usage: benchmarks/collect_wavefront.sh <tag> <workload> [spp]
set -u
tag=${1:-wf}; w=${2:-final1080}; spp=${3:-0}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
sppflag=(); [ "$spp" != 0 ] && sppflag=(--spp "$spp")
common=(--workload "$w" "${sppflag[@]}" --no-cpu-baseline --no-linear-scan-leg --no-live-pmc)
timeout -k 10 300 python3 bench.py --kernel wavefront --steps 3 --warmup 1 "${common[@]}" > "$out/bench_wavefront_$w.json" 2> "$out/bench_wavefront_$w.err" || exit 1
timeout -k 10 300 python3 bench.py --kernel mega --steps 5 --warmup 1 --serial "${common[@]}" > "$out/bench_mega_serial_$w.json" 2> "$out/bench_mega_serial_$w.err" || exit 1
cmd=(python3 bench.py --kernel wavefront --steps 1 --warmup 0 --serial --no-ramp "${common[@]}")
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$w" -o trace -- "${cmd[@]}" > "$out/trace_$w.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
  --output-format csv -d "$out/pmc_sq1_$w" -o pmc -- "${cmd[@]}" > "$out/pmc_$w.log" 2>&1 || exit 1
echo "wavefront $w collected"
