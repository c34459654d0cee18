"""Counterpart of reference 4DOF/Scripts/01_generate_normal_runs.py (constants :23-26; plotting omitted): the jittered normal runs,
integrated on the MI355X in one launch, written as Data/raw/normal/normal_seed<k>.csv.
    python -m Scripts.01_generate_normal_runs --root /path/to/4DOF [--runs 10]"""
import argparse
import sys
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent.parent
if str(_PKG) not in sys.path:
    sys.path.insert(0, str(_PKG))

from shm_hip import simulate   # noqa: E402

N_NORMAL_RUNS, FORCE_RMS, BASE_SEED = 10, 50.0, 2025


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--runs", type=int, default=N_NORMAL_RUNS)
    ap.add_argument("--jitter-seed", type=int, default=None)
    a = ap.parse_args()
    tables, _ = simulate.normal_runs(a.runs, BASE_SEED, FORCE_RMS, a.jitter_seed, dtype=torch.float64)
    for i in range(a.runs):
        p = simulate.save_csv(tables[i], Path(a.root) / f"Data/raw/normal/normal_seed{BASE_SEED + i}.csv")
        print(f"[OK] normal run saved: {p}")


if __name__ == "__main__":
    main()
