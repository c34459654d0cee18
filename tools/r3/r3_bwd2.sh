set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_vae_bf16.py tests/test_gpu_fullsize_parity.py tests/test_gpu_multirank.py tests/test_gpu_boundary.py -x -q 2>&1 | tail -5 > gpurun_out/r3_bwd2_tests.log; rc=$?; cat gpurun_out/r3_bwd2_tests.log; [ $rc = 0 ] || exit 1
B="python bench.py --precision bf16 --steps 100 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-inference-leg --no-parity-leg"
for i in 1 2; do
for v in 1 0; do
SHM_BWD_PAIR_FORM1=$v timeout -k 10 300 $B > gpurun_out/r3_bwd2_${v}_$i.json 2> gpurun_out/r3_bwd2_${v}_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_bwd2_${v}_$i.json"))
print("form1=$v", j["value"], j["ms_per_step"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "lstm" in k})
PY
done
done
