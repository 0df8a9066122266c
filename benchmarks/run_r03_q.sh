set -u
tag=${1:-r03_q}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
for p in low high default; do
for q in 2 4; do
  RT_EXP_TRACE_PRIORITY=$p GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --no-linear-scan-leg > gpurun_out/$tag/bench_${p}_q${q}.json 2> gpurun_out/$tag/bench_${p}_q${q}.err || exit 1
done
done
python3 - <<PY
import json
for p in ("low","high","default"):
  for q in (2,4):
    b=json.load(open(f"gpurun_out/${tag}/bench_{p}_q{q}.json")); r=b["roofline"]
    print(p, q, "%.4g paths/s" % b["value"], "%.3f ms/step" % b["ms_per_step"], "kernel %.3f" % r["kernel_ms_avg"], r["pipelined_step_ms_min_median_max"])
PY
