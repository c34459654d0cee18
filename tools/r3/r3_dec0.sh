set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_split_pair.py tests/test_gpu_vae.py tests/test_gpu_fullsize_parity.py tests/test_gpu_boundary.py -x -q 2>&1 | tail -3
timeout -k 10 300 python tools/r3/r3_wscale.py 2>&1 | grep split
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/parity_errors.jsonl")]
for r in rows:
    if r["test"]=="forward_goldens" and "split" in r["case"]: print(r["case"], {k:float(f"{v:.2e}") for k,v in r["errors"].items()})
    if r["test"]=="bench_shape_train_step" and r["case"]=="split": print("B512", {k:float(f"{v:.2e}") for k,v in r["errors"].items()})
PY
