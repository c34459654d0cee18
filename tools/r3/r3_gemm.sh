set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_split_pair.py tests/test_gpu_fullsize_parity.py tests/test_gpu_vae.py tests/test_gpu_multirank.py tests/test_gpu_vae_bf16.py -x -q 2>&1 | tail -3
B="python bench.py --precision split --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-inference-leg"
for i in 1 2; do
SHM_GEMM_SP_V2=1 timeout -k 10 300 $B > gpurun_out/r3_g_v2_$i.json 2> gpurun_out/r3_g_v2_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_g_v2_$i.json"))
print("gemm v2", j["value"], j["ms_per_step"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "gemm" in k})
PY
timeout -k 10 300 $B > gpurun_out/r3_g_v3_$i.json 2> gpurun_out/r3_g_v3_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_g_v3_$i.json"))
print("gemm v3", j["value"], j["ms_per_step"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "gemm" in k})
PY
done
