set -u
tag=${1:-r03_b}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$tag/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/$tag/pytest_gpu.txt; exit 1; }
tail -3 gpurun_out/$tag/pytest_gpu.txt
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_final1080.json 2> gpurun_out/$tag/bench_final1080.err || exit 1
for w in glass1080 dense1080; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/bench_$w.json 2> gpurun_out/$tag/bench_$w.err || exit 1
done
for w in final1080 dense1080; do
  out=$PWD/gpurun_out/$tag/lds_$w
  mkdir -p $out
  spp=16; [ "$w" = dense1080 ] && spp=4
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES \
      --output-format csv -d "$out" -o pmc -- python3 $PWD/bench.py --pmc-child --workload "$w" --spp $spp > "$out/log.txt" 2>&1 || exit 1
done
python3 - <<PY
import json
for w in ("final1080","glass1080","dense1080"):
    b=json.load(open(f"gpurun_out/%s/bench_{w}.json" % "${tag}"))
    print(w, "%.4g paths/s" % b["value"], "%.3f ms/step" % b["ms_per_step"], "kernel %.3f ms" % b["roofline"]["kernel_ms_avg"])
PY
