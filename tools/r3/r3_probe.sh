set -o pipefail
mkdir -p gpurun_out
B="python bench.py --precision split --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-inference-leg"
for i in 1 2; do
timeout -k 10 300 $B > gpurun_out/r3_p_base_$i.json 2> gpurun_out/r3_p_base_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_p_base_$i.json"))
print("base   ", j["value"], j["ms_per_step"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "lstm" in k})
PY
SHM_HIP_LIB=$PWD/hybrid-vae-cnn-for-shm_amd/csrc/libshm_probe16.so timeout -k 10 300 $B > gpurun_out/r3_p_probe_$i.json 2> gpurun_out/r3_p_probe_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_p_probe_$i.json"))
print("probe16", j["value"], j["ms_per_step"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "lstm" in k})
PY
done
timeout -k 10 300 python bench.py --precision split --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-parity-leg --profile-steps 0 > gpurun_out/r3_p_inf.json 2> gpurun_out/r3_p_inf.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_p_inf.json"))
print("inference", j["inference"])
PY
