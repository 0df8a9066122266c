set -u
mkdir -p gpurun_out/r03_a
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_a/bench_final1080.json 2> gpurun_out/r03_a/bench_final1080.err || exit 1
echo bench done
timeout -k 10 900 bash benchmarks/lds_attribution.sh r03_a/lds final1080 dense1080 glass1080 || exit 1
python3 benchmarks/lds_attribution.py gpurun_out/r03_a/lds > gpurun_out/r03_a/lds_attribution.json || exit 1
echo lds done
timeout -k 10 300 python3 benchmarks/tile_balance.py --spp 64 > gpurun_out/r03_a/tile_balance.json 2> gpurun_out/r03_a/tile_balance.err || exit 1
echo balance done
timeout -k 10 700 bash benchmarks/collect_wavefront.sh r03_a/wf final1080 || exit 1
timeout -k 10 900 bash benchmarks/collect_wavefront.sh r03_a/wf glass1080 || exit 1
