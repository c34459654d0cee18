set -o pipefail
mkdir -p gpurun_out/trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in split bf16; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace/$P -- python3 bench.py --precision $P --steps 8 --warmup 4 --no-cpu-baseline --no-fp32-leg --no-parity-leg --no-inference-leg --no-cnn-leg --profile-steps 0 > gpurun_out/trace/$P.json 2> gpurun_out/trace/$P.err
f=$(find gpurun_out/trace/$P -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $f 3 > gpurun_out/trace/${P}_timeline.txt
rm -rf gpurun_out/trace/$P
done
tail -3 gpurun_out/trace/split_timeline.txt
