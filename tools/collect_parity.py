#!/usr/bin/env python3
"""Aggregate gpurun_out/parity_errors.jsonl (appended by the -m gpu tests through tests/conftest.py::record_parity) into the tracked
summary profiles/rNN_parity_errors.json: the LAST record of every (test, case) wins, so re-runs refresh instead of piling up.

    python tools/collect_parity.py [gpurun_out/parity_errors.jsonl] [profiles/r02_parity_errors.json]
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "parity_errors.jsonl"
    dst = Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "profiles" / "r02_parity_errors.json"
    out = {}
    for line in src.read_text().splitlines():
        if line.strip():
            d = json.loads(line)
            out.setdefault(d["test"], {})[d["case"]] = d["errors"]
    note = ("Measured errors of the GPU parity tests (pytest -m gpu on 1x MI355X), written by tests/conftest.py::record_parity and collected with "
            "tools/collect_parity.py.  forward_goldens / dropin_train / fused_trainer: vs tests/golden/*.npz generated from the reference's own classes; "
            "bench_shape_train_step / ten_step_drift_h128: vs oracle/torch_cpu.py computed on the host inside the test (B = 512, T = 256 / ten steps at "
            "H = 128); reference_artifacts: stages 00-05 on the reference's own CSV windows vs the numbers the reference holds (held_*).")
    dst.write_text(json.dumps({"_note": note, **out}, indent=1) + "\n")
    print(dst, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
