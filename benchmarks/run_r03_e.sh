set -u
tag=${1:-r03_e}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
timeout -k 10 120 _out/mfma_filter 512 200 > gpurun_out/$tag/mfma_filter_512.txt 2>&1; echo "mfma rc=$?"; cat gpurun_out/$tag/mfma_filter_512.txt
timeout -k 10 120 _out/mfma_filter 2048 50 > gpurun_out/$tag/mfma_filter_2048.txt 2>&1; echo "mfma rc=$?"; cat gpurun_out/$tag/mfma_filter_2048.txt
for rep in 1 2 3; do
  timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --no-linear-scan-leg > gpurun_out/$tag/bench_$rep.json 2> gpurun_out/$tag/bench_$rep.err || exit 1
done
python3 - <<PY
import json
for rep in (1,2,3):
    b=json.load(open(f"gpurun_out/${tag}/bench_{rep}.json")); r=b["roofline"]
    print(rep, "%.4g paths/s" % b["value"], "%.3f ms/step" % b["ms_per_step"], "kernel %.3f" % r["kernel_ms_avg"], "spacing", r["pipelined_step_ms_min_median_max"])
PY
