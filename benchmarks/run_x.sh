set -u
tag=${1:-r03_x}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$tag/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/$tag/pytest_gpu.txt; exit 1; }
tail -2 gpurun_out/$tag/pytest_gpu.txt
for w in final1080 glass1080 dense1080; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-linear-scan-leg > gpurun_out/$tag/bench_$w.json 2> gpurun_out/$tag/bench_$w.err || exit 1
  python3 -c "
import json
b=json.load(open('gpurun_out/$tag/bench_$w.json')); r=b['roofline']; i=r.get('issue') or {}
print('$w', '%.4g' % b['value'], '%.3f' % b['ms_per_step'], 'kernel %.3f' % r['kernel_ms_avg'], 'valu/trip', i.get('valu_wave_instructions_per_trip'), 'cand/seg %.2f' % r['exact_evaluations_per_segment'])"
done
