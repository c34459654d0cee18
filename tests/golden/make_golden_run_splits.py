#!/usr/bin/env python3
"""Golden for the run-splits description (reference 4DOF/Scripts/00_make_run_splits.py): the reference's main() run in a scratch tree of
header-only-plus-N-rows CSV files.  Run in the build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_run_splits.py
Writes tests/golden/run_splits_small.json = {"rows": {relative path: data rows}, "splits": <the reference's run_splits.json>}."""
import importlib.util
import json
import os
import sys
import tempfile
from pathlib import Path

OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
ROWS = {"Data/raw/normal/normal_seed2025.csv": 1001, "Data/raw/normal/normal_seed2026.csv": 257, "Data/raw/normal/short.csv": 99,
        "Data/raw/faults/sensor_fault/bias_x3/bias_x3.csv": 100, "Data/raw/faults/sensor_fault/noise_x4/noise_x4.csv": 103,
        "Data/raw/faults/structural_fault/stiff_red_40pct/stiff_red_40pct.csv": 1001}


def main():
    spec = importlib.util.spec_from_file_location("ref_splits", "/root/reference/4DOF/Scripts/00_make_run_splits.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory() as td:
        for rel, n in ROWS.items():
            p = Path(td) / rel
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_text("x1,x2\n" + "".join("0.0,1.0\n" for _ in range(n)))
        cwd = os.getcwd()
        os.chdir(td)
        try:
            mod.main()
            splits = json.loads(Path("Data/processed/run_splits.json").read_text())
        finally:
            os.chdir(cwd)
    (OUT / "run_splits_small.json").write_text(json.dumps({"rows": ROWS, "splits": splits}))
    print({k: v for k, v in splits["totals"].items()})


if __name__ == "__main__":
    main()
