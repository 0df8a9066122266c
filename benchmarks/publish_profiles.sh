#!/bin/bash
# Copies the summaries of gpurun_out/<tag>/ (written by benchmarks/collect_r02.sh <tag>) into profiles/<tag>_*.
# usage: benchmarks/publish_profiles.sh <tag>
set -eu
tag=$1
src=gpurun_out/$tag
for w in final1080 glass1080 dense1080 final4k; do cp "$src/bench_$w.json" "profiles/${tag}_bench_$w.json"; done
cp "$src/final1080/trace/trace_kernel_stats.csv" "profiles/${tag}_trace_kernel_stats.csv"
python3 benchmarks/summarize_pmc.py "$src/final1080" > "profiles/${tag}_pmc_summary.json"
python3 benchmarks/summarize_pmc.py "$src/final1080_mix" > "profiles/${tag}_instruction_mix.json"
for w in full_scan glass1080 dense1080; do
  cp "$src/$w/trace/trace_kernel_stats.csv" "profiles/${tag}_${w}_trace_kernel_stats.csv"
  python3 benchmarks/summarize_pmc.py "$src/$w" > "profiles/${tag}_${w}_pmc_summary.json"
done
cp "$src/phase_profile.jsonl" "profiles/${tag}_phase_profile.json"
cp "$src/stage_profile.jsonl" "profiles/${tag}_stage_profile.json"
cp "$src/tail_probe.txt" "profiles/${tag}_tail_probe.txt"
for w in final1080 glass1080 dense1080; do
  [ -s "$src/thread_efficiency_by_part_$w.json" ] && cp "$src/thread_efficiency_by_part_$w.json" "profiles/${tag}_thread_efficiency_by_part_$w.json"
done
ls profiles | grep "^${tag}_"
