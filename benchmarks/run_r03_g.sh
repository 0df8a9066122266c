set -u
tag=${1:-r03_g}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
for v in base bw5; do
  if [ $v = base ]; then unset RT_AMD_DEVICE_LIB; else export RT_AMD_DEVICE_LIB=$PWD/_out/exp/lib_$v.so; fi
  timeout -k 10 300 python3 bench.py --workload dense1080 --steps 10 --warmup 2 --no-cpu-baseline --no-live-pmc --no-linear-scan-leg > gpurun_out/$tag/bench_dense_$v.json 2> gpurun_out/$tag/bench_dense_$v.err || exit 1
done
unset RT_AMD_DEVICE_LIB
python3 - <<PY
import json
for v in ("base","bw5"):
    b=json.load(open(f"gpurun_out/${tag}/bench_dense_{v}.json")); r=b["roofline"]
    print(v, "%.4g paths/s" % b["value"], "%.3f ms/step" % b["ms_per_step"], "kernel %.3f" % r["kernel_ms_avg"])
PY
