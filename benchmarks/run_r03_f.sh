set -u
tag=${1:-r03_f}
mkdir -p gpurun_out/$tag
for n in 512 2048; do
  timeout -k 10 120 _out/mfma_filter $n $((102400 / n)) > gpurun_out/$tag/mfma_filter_$n.txt 2>&1; echo "mfma rc=$?"; cat gpurun_out/$tag/mfma_filter_$n.txt
done
