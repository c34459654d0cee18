# end-of-round evidence: profiles (r03), the default bench line, the driver's short bench line, step timelines
set -o pipefail
mkdir -p gpurun_out
bash tools/profile_round.sh r03 > gpurun_out/r3_profile.log 2>&1
echo "profiles done: $(ls gpurun_out/prof | grep -c r03_) files"
timeout -k 10 600 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "default bench rc $?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_driver.json 2> gpurun_out/r3_bench_driver.err; echo "driver bench rc $?"
bash tools/r3/r3_trace.sh > gpurun_out/r3_trace.log 2>&1; echo "trace rc $?"
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3_bench_default.json"))
print("bf16", j["value"], j["ms_per_step"], "frac", j["roofline"]["frac"], j["roofline"]["kernel"])
pp=j["parity_path"]; print("split", pp["value"], pp["ms_per_step"], pp["roofline"]["kernel"], pp["roofline"]["frac"], pp["roofline"]["avg_launch_ms"])
print("inference", {k:v["value"] for k,v in j["inference"].items()})
print("cnn", j["cnn_train"]["value"], "fp32", j["fp32_exact_path"]["value"], "cpu", j["cpu_baseline"]["value"])
PY
