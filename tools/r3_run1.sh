set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_split_pair.py -x -q 2>&1 | tail -15 > gpurun_out/r3_pair_tests.log
rc=$?
cat gpurun_out/r3_pair_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --precision split --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-inference-leg --no-cnn-leg > gpurun_out/r3_split_pair.json 2> gpurun_out/r3_split_pair.err && tail -c 1500 gpurun_out/r3_split_pair.json &&
SHM_PAIR_LAUNCH=0 timeout -k 10 300 python bench.py --precision split --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-inference-leg --no-cnn-leg > gpurun_out/r3_split_nopair.json 2> gpurun_out/r3_split_nopair.err && tail -c 1200 gpurun_out/r3_split_nopair.json
