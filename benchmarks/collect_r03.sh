#!/bin/bash
# Everything round 3 measures, into gpurun_out/<tag>/ (needs benchmarks/build_exp.sh 31..40 51..58 and _out/mfma_filter):
#   the evidence of benchmarks/collect_r02.sh (bench lines, kernel trace and counters per workload, instruction mix,
#   phase / stage / tail / fill profiles), the stage profile of the dense field, the LDS bank-conflict attribution,
#   the wavefront variant on the final and the glass scene, the tile balance of the 8-rank split, the MFMA form of
#   the filter.     usage: benchmarks/collect_r03.sh <tag>
set -u
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 1100 bash benchmarks/collect_r02.sh $tag || exit 1
RT_STAGE_WORKLOAD=dense1080 timeout -k 10 300 python3 benchmarks/stage_profile.py > "$out/stage_profile_dense1080.jsonl" 2> /dev/null
timeout -k 10 600 bash benchmarks/lds_attribution.sh $tag/lds final1080 dense1080 glass1080 > "$out/lds_log.txt" 2>&1 || exit 1
python3 benchmarks/lds_attribution.py "$out/lds" > "$out/lds_attribution.json" || exit 1
timeout -k 10 300 python3 benchmarks/tile_balance.py --spp 64 > "$out/tile_balance.json" 2> "$out/tile_balance.err" || exit 1
timeout -k 10 600 bash benchmarks/collect_wavefront.sh $tag/wf final1080 > /dev/null 2>&1 || exit 1
timeout -k 10 900 bash benchmarks/collect_wavefront.sh $tag/wf glass1080 > /dev/null 2>&1 || exit 1
for n in 512 2048; do timeout -k 10 120 _out/mfma_filter $n $((102400 / n)) > "$out/mfma_filter_$n.txt" 2>&1 || exit 1; done
echo "round 3 evidence collected"
