set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split_pair.py tests/test_gpu_vae.py -x -q 2>&1 | tail -5 > gpurun_out/r3_run3_tests.log
rc=$?
cat gpurun_out/r3_run3_tests.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
timeout -k 10 300 python bench.py --precision split --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-inference-leg --no-cnn-leg > gpurun_out/r3_split_pair_$i.json 2> gpurun_out/r3_split_pair_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_split_pair_$i.json"))
print("pair", j["value"], j["ms_per_step"], j["step_ms"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "fwd" in k or "xproj" in k})
PY
SHM_PAIR_LAUNCH=0 timeout -k 10 300 python bench.py --precision split --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-inference-leg --no-cnn-leg > gpurun_out/r3_split_nopair_$i.json 2> gpurun_out/r3_split_nopair_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_split_nopair_$i.json"))
print("nopair", j["value"], j["ms_per_step"], j["step_ms"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "fwd" in k or "xproj" in k})
PY
done
