set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split_pair.py tests/test_gpu_fullsize_parity.py -x -q 2>&1 | tail -2
B="python bench.py --precision split --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-inference-leg"
for i in 1 2; do
timeout -k 10 300 $B > gpurun_out/r3_dx2_$i.json 2> gpurun_out/r3_dx2_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_dx2_$i.json"))
print("dx depth 2", j["value"], j["ms_per_step"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "bwd" in k or "fwd" in k})
PY
done
