# What the round end runs, plus the two-rank rehearsal on one GPU:  benchmarks/check_round_end.sh <tag>
set -u
tag=${1:-r03_n}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
RT_BENCH_REHEARSAL=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 2 --warmup 1 --spp 64 > gpurun_out/$tag/bench_rehearsal_2ranks.json 2> gpurun_out/$tag/bench_rehearsal_2ranks.err || { tail -20 gpurun_out/$tag/bench_rehearsal_2ranks.err; exit 1; }
python3 -c "
import json
b=json.load(open('gpurun_out/$tag/bench_rehearsal_2ranks.json'))
print(b['n_gpus'], b['scaling'], b['value'], b['config']['workload'][:80], b['per_rank'], b.get('weak_scaling'))
"
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$tag/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/$tag/pytest_gpu.txt; exit 1; }
tail -2 gpurun_out/$tag/pytest_gpu.txt
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 400 python3 bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err || exit 1
python3 -c "
import json
b=json.load(open('gpurun_out/$tag/bench_default.json'))
print(b['value'], b['ms_per_step'], b['steps'], b['roofline']['frac'], b['cpu_baseline']['value'])
"
