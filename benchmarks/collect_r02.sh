#!/bin/bash
# One round's evidence into gpurun_out/<tag>/ : bench lines of every workload, kernel trace + counters of the
# headline workload rendered serially (every launch then shows its own duration), instruction mix, phase and
# stage profiles, the same counters for the dense and the glass workload.
# usage: benchmarks/collect_r02.sh <tag>
set -u
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$out/bench_final1080.json" 2> "$out/bench_final1080.err" || exit 1
for w in glass1080 dense1080; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > "$out/bench_$w.json" 2> "$out/bench_$w.err" || exit 1
done
timeout -k 10 300 python3 bench.py --workload final4k --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-linear-scan-leg > "$out/bench_final4k.json" 2> "$out/bench_final4k.err" || exit 1
timeout -k 10 700 bash benchmarks/collect_profiles.sh $tag/final1080 > /dev/null 2>&1 || exit 1
timeout -k 10 300 bash benchmarks/collect_mix.sh $tag/final1080_mix > /dev/null 2>&1 || exit 1
timeout -k 10 700 bash benchmarks/collect_profiles.sh $tag/full_scan --full-scan > /dev/null 2>&1 || exit 1
for w in glass1080 dense1080; do
  timeout -k 10 900 bash benchmarks/collect_profiles.sh $tag/$w --workload $w > /dev/null 2>&1 || exit 1
done
timeout -k 10 200 python3 benchmarks/phase_profile.py --spp 64 > "$out/phase_profile.jsonl" 2> /dev/null
timeout -k 10 300 python3 benchmarks/stage_profile.py > "$out/stage_profile.jsonl" 2> /dev/null
timeout -k 10 100 python3 benchmarks/tail_probe.py > "$out/tail_probe.txt" 2> /dev/null
if [ -f _out/exp/lib_exp39.so ]; then
  for w in final1080 glass1080 dense1080; do
    RT_AMD_DEVICE_LIB=$PWD/_out/exp/lib_exp39.so timeout -k 10 120 python3 benchmarks/fill_profile.py --workload $w > "$out/thread_efficiency_by_part_$w.json" 2> /dev/null
  done
fi
echo collected
