set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_vae_bf16.py tests/test_gpu_split_pair.py tests/test_gpu_fullsize_parity.py tests/test_gpu_vae.py tests/test_gpu_multirank.py tests/test_gpu_c_abi.py -x -q 2>&1 | tail -3 > gpurun_out/r3_poll_tests.log; rc=$?; cat gpurun_out/r3_poll_tests.log; [ $rc = 0 ] || exit 1
for P in bf16; do
B="python bench.py --precision $P --steps 100 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-inference-leg --no-parity-leg"
for i in 1 2; do
for v in prev new; do
if [ $v = new ]; then L=""; else L="$PWD/hybrid-vae-cnn-for-shm_amd/csrc/libshm_prev.so"; fi
SHM_HIP_LIB=$L timeout -k 10 300 $B > gpurun_out/r3_poll_${P}_${v}_$i.json 2> gpurun_out/r3_poll_${P}_${v}_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_poll_${P}_${v}_$i.json"))
print("$P $v", j["value"], j["ms_per_step"], {k:v for k,v in j["roofline"]["kernels_ms"].items() if "lstm" in k})
PY
done
done
done
