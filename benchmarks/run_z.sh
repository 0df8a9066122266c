set -u
mkdir -p gpurun_out/r03_z
export TMPDIR=/tmp RT_EXP_REPS=6
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_z/parity.txt 2>&1 || { tail -20 gpurun_out/r03_z/parity.txt; exit 1; }
tail -2 gpurun_out/r03_z/parity.txt
out=gpurun_out/r03_z/batch_sweep2.txt
: > $out
for i in 1 2 3; do
  timeout -k 10 120 python3 benchmarks/exp_scene_time.py 100 16 >> $out 2>&1 || exit 1
done
cat $out
timeout -k 10 300 python3 bench.py --workload dense1080 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03_z/bench_dense1080.json 2> gpurun_out/r03_z/bench_dense1080.err || exit 1
python3 -c "
import json; b=json.load(open('gpurun_out/r03_z/bench_dense1080.json')); print('dense %.4g paths/s %.3f ms/step kernel %.3f' % (b['value'], b['ms_per_step'], b['roofline']['kernel_ms_avg']))"
