set -o pipefail
mkdir -p gpurun_out
B="python bench.py --precision split --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-cnn-leg --no-inference-leg --profile-steps 0"
for n in 1 2 4 1 2 4; do
SHM_SP_BWD_CHUNKS=$n timeout -k 10 300 $B > gpurun_out/r3_bc_$n.json 2> gpurun_out/r3_bc_$n.err && python - <<PY
import json
j=json.load(open("gpurun_out/r3_bc_$n.json"))
print("bwd chunks $n", j["value"], j["ms_per_step"], j["final_stats"])
PY
done
