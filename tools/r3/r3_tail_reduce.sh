set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_vae_bf16.py tests/test_gpu_fullsize_parity.py tests/test_gpu_vae.py tests/test_gpu_multirank.py tests/test_gpu_c_abi.py tests/test_gpu_boundary.py -x -q 2>&1 | tail -3 > gpurun_out/r3_tr_tests.log; rc=$?; cat gpurun_out/r3_tr_tests.log; [ $rc = 0 ] || exit 1
B="python bench.py --precision bf16 --steps 100 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-inference-leg --no-parity-leg"
for i in 1 2 3; do
for v in 0 1; do
SHM_BWD_TAIL_REDUCE=$v timeout -k 10 300 $B > gpurun_out/r3_tr_${v}_$i.json 2> gpurun_out/r3_tr_${v}_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_tr_${v}_$i.json"))
k=j["roofline"]["kernels_ms"]
print("tail=$v", j["value"], j["ms_per_step"], {a:k[a] for a in ("lstm_bwd.dec","lstm_bwd.enc","dw_gemm.dec0","dw_gemm.out")})
PY
done
done
