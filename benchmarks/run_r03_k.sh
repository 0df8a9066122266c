set -u
tag=${1:-r03_k}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "dense or streamed or too_large or distance_cut or largest or random_scenes or close_calls or list_length or outside or wavefront" > gpurun_out/$tag/pytest_dense.txt 2>&1 || { tail -30 gpurun_out/$tag/pytest_dense.txt; exit 1; }
tail -2 gpurun_out/$tag/pytest_dense.txt
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --workload dense1080 --steps 10 --warmup 2 --no-cpu-baseline --no-live-pmc --no-linear-scan-leg > gpurun_out/$tag/bench_dense_$rep.json 2> gpurun_out/$tag/bench_dense_$rep.err || exit 1
done
python3 - <<PY
import json
for v in (1,2):
    b=json.load(open(f"gpurun_out/${tag}/bench_dense_{v}.json")); r=b["roofline"]
    print(v, "%.4g paths/s" % b["value"], "%.3f ms/step" % b["ms_per_step"], "kernel %.3f" % r["kernel_ms_avg"], "box/path %.1f sphere/path %.1f" % (r["box_tests_per_path"], r["sphere_tests_per_path"]))
PY
