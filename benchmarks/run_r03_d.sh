set -u
tag=${1:-r03_d}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
for rep in 1 2; do
for words in 3 4; do
  RT_EXP_RAD_WORDS=$words timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --no-linear-scan-leg > gpurun_out/$tag/bench_w${words}_$rep.json 2> gpurun_out/$tag/bench_w${words}_$rep.err || exit 1
done
done
python3 - <<PY
import json
for rep in (1,2):
  for w in (3,4):
    b=json.load(open(f"gpurun_out/${tag}/bench_w{w}_{rep}.json")); r=b["roofline"]
    print(w, rep, "%.4g paths/s" % b["value"], "%.3f ms/step" % b["ms_per_step"], "kernel %.3f" % r["kernel_ms_avg"], "serial step %.3f" % r["serial_leg_ms_per_step"], "bracketed %.2f" % r["kernel_ms_bracketed_in_pipeline"])
PY
