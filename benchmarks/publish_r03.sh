#!/bin/bash
# Copies the summaries of gpurun_out/<tag>/ (written by benchmarks/collect_r03.sh <tag>) into profiles/<prefix>_*.
# usage: benchmarks/publish_r03.sh <tag> [prefix]
set -eu
tag=$1
prefix=${2:-$tag}
src=gpurun_out/$tag
# benchmarks/publish_profiles.sh names its files after the tag: publish under the tag, then rename
bash benchmarks/publish_profiles.sh "$tag" > /dev/null
if [ "$prefix" != "$tag" ]; then
  for f in profiles/${tag}_*; do mv "$f" "profiles/${prefix}_${f#profiles/${tag}_}"; done
fi
cp "$src/stage_profile_dense1080.jsonl" "profiles/${prefix}_stage_profile_dense1080.json"
cp "$src/lds_attribution.json" "profiles/${prefix}_lds_attribution.json"
cp "$src/tile_balance.json" "profiles/${prefix}_tile_balance.json"
for w in final1080 glass1080; do
  cp "$src/wf/bench_wavefront_$w.json" "profiles/${prefix}_wavefront_bench_$w.json"
  cp "$src/wf/bench_mega_serial_$w.json" "profiles/${prefix}_wavefront_megakernel_serial_bench_$w.json"
  cp "$src/wf/trace_$w/trace_kernel_stats.csv" "profiles/${prefix}_wavefront_kernel_stats_$w.csv"
  tmp=$(mktemp -d)
  cp -r "$src/wf/pmc_sq1_$w" "$tmp/pmc_sq1"
  python3 benchmarks/summarize_pmc.py "$tmp" | python3 -c "
import json, sys
d = json.load(sys.stdin)
out = {}
for k, v in d.items():
    if 'wf_' in k or 'resolve' in k:
        v = dict(v)
        if v.get('SQ_ACTIVE_INST_VALU'):
            v['thread_efficiency'] = v['SQ_THREAD_CYCLES_VALU'] / (64.0 * v['SQ_ACTIVE_INST_VALU'])
        out[k] = v
print(json.dumps(out, indent=1, sort_keys=True))" > "profiles/${prefix}_wavefront_pmc_summary_$w.json"
  rm -rf "$tmp"
done
cat "$src/mfma_filter_512.txt" "$src/mfma_filter_2048.txt" > "profiles/${prefix}_mfma_filter.txt"
ls profiles | grep "^${prefix}_"
