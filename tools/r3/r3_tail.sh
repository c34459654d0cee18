set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split_pair.py tests/test_gpu_fullsize_parity.py tests/test_gpu_vae.py -x -q 2>&1 | tail -3
B="python bench.py --precision split --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-inference-leg --profile-steps 0"
for i in 1 2; do
SHM_SP_TAIL_ON_SIDE=1 timeout -k 10 300 $B > gpurun_out/r3_t_side_$i.json 2> gpurun_out/r3_t_side_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_t_side_$i.json"))
print("tail on side", j["value"], j["ms_per_step"], j["final_stats"]["grad_norm"])
PY
timeout -k 10 300 $B > gpurun_out/r3_t_main_$i.json 2> gpurun_out/r3_t_main_$i.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_t_main_$i.json"))
print("tail on main", j["value"], j["ms_per_step"], j["final_stats"]["grad_norm"])
PY
done
