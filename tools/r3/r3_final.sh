# end-of-round evidence in ONE gpurun call: the full GPU suite (parity errors collected), smoke(), profiles (r03), the default bench line, the
# driver's short bench line, step timelines
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/parity_errors.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r3_fullsuite.log; cat gpurun_out/r3_fullsuite.log
python tools/collect_parity.py gpurun_out/parity_errors.jsonl gpurun_out/r03_parity_errors.json | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
bash tools/profile_round.sh r03 > gpurun_out/r3_profile.log 2>&1
echo "profiles done: $(ls gpurun_out/prof | grep -c r03_) files"
cp gpurun_out/prof/r03_* profiles/     # on the box: the bench below quotes roofline.traffic from summaries whose source hashes match this tree
timeout -k 10 600 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "default bench rc $?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_driver.json 2> gpurun_out/r3_bench_driver.err; echo "driver bench rc $?"
bash tools/r3/r3_trace.sh > gpurun_out/r3_trace.log 2>&1; echo "trace rc $?"
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3_bench_default.json"))
print("bf16", j["value"], j["ms_per_step"], "frac", j["roofline"]["frac"], j["roofline"]["kernel"], j["roofline"]["traffic"])
pp=j["parity_path"]; print("split", pp["value"], pp["ms_per_step"], pp["roofline"]["kernel"], pp["roofline"]["frac"], pp["roofline"]["avg_launch_ms"], pp["roofline"]["traffic"])
print("inference", {k:v["value"] for k,v in j["inference"].items()})
print("cnn", j["cnn_train"]["value"], "fp32", j["fp32_exact_path"]["value"], "cpu", j["cpu_baseline"]["value"])
PY
