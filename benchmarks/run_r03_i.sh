set -u
tag=${1:-r03_i}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
true
true
for v in m0_0 m32_1 m40_1 m48_1 m64_1 m96_1 m0_0 m40_1; do
  if [ $v = base ]; then unset RT_AMD_DEVICE_LIB; else export RT_AMD_DEVICE_LIB=$PWD/_out/exp/lib_$v.so; fi
  timeout -k 10 300 python3 bench.py --workload dense1080 --steps 10 --warmup 2 --no-cpu-baseline --no-live-pmc --no-linear-scan-leg > gpurun_out/$tag/bench_dense_$v.json 2> gpurun_out/$tag/bench_dense_$v.err || exit 1
done
unset RT_AMD_DEVICE_LIB
python3 - <<PY
import json
for v in ("m0_0","m32_1","m40_1","m48_1","m64_1","m96_1"):
    b=json.load(open(f"gpurun_out/${tag}/bench_dense_{v}.json")); r=b["roofline"]
    print(v, "%.4g paths/s" % b["value"], "%.3f ms/step" % b["ms_per_step"], "kernel %.3f" % r["kernel_ms_avg"], "box/path %.1f sphere/path %.1f" % (r["box_tests_per_path"], r["sphere_tests_per_path"]))
PY
