set -o pipefail
mkdir -p gpurun_out
SHM_SP_BWD_DX_PAIR=1 timeout -k 10 600 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_vae.py -x -q -k "split or Split" 2>&1 | tail -3
B="python bench.py --precision split --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-inference-leg"
for i in 1 2; do
timeout -k 10 300 $B > gpurun_out/r3_d_off_$i.json 2> gpurun_out/r3_d_off_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_d_off_$i.json"))
print("dx pair off", j["value"], j["ms_per_step"], j["final_stats"]["grad_norm"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "bwd" in k or "dx" in k})
PY
SHM_SP_BWD_DX_PAIR=1 timeout -k 10 300 $B > gpurun_out/r3_d_on_$i.json 2> gpurun_out/r3_d_on_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_d_on_$i.json"))
print("dx pair on ", j["value"], j["ms_per_step"], j["final_stats"]["grad_norm"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "bwd" in k or "dx" in k})
PY
done
