set -o pipefail
mkdir -p gpurun_out
B="python bench.py --precision bf16 --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-inference-leg --no-parity-leg"
for v in base b1 b2 b3; do
if [ $v = base ]; then L=""; else L="$PWD/hybrid-vae-cnn-for-shm_amd/csrc/libshm_probe_$v.so"; fi
SHM_HIP_LIB=$L timeout -k 10 300 $B > gpurun_out/r3_bp_${v}.json 2> gpurun_out/r3_bp_${v}.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_bp_${v}.json"))
print("$v", j["value"], j["ms_per_step"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "lstm" in k})
PY
done
