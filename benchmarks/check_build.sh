# GPU suite, bench lines of three workloads and the LDS attribution of one build:  benchmarks/check_build.sh <tag>
set -u
tag=${1:-r03_c}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$tag/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/$tag/pytest_gpu.txt; exit 1; }
tail -3 gpurun_out/$tag/pytest_gpu.txt
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_final1080.json 2> gpurun_out/$tag/bench_final1080.err || exit 1
for w in glass1080 dense1080; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/bench_$w.json 2> gpurun_out/$tag/bench_$w.err || exit 1
done
python3 - <<PY
import json
for w in ("final1080","glass1080","dense1080"):
    b=json.load(open(f"gpurun_out/${tag}/bench_{w}.json"))
    print(w, "%.4g paths/s" % b["value"], "%.3f ms/step" % b["ms_per_step"], "kernel %.3f ms" % b["roofline"]["kernel_ms_avg"], "traffic", b["roofline"].get("traffic"))
PY
timeout -k 10 900 bash benchmarks/lds_attribution.sh $tag/lds final1080 dense1080 || exit 1
python3 benchmarks/lds_attribution.py gpurun_out/$tag/lds > gpurun_out/$tag/lds_attribution.json || exit 1
