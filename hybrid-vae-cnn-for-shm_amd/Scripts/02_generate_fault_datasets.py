"""Counterpart of reference 4DOF/Scripts/02_generate_fault_datasets.py (constants :27-28, cases :153,193-198; plotting omitted): the
baseline run, four stiffness reductions and four sensor faults, generated on the MI355X.
    python -m Scripts.02_generate_fault_datasets --root /path/to/4DOF"""
import argparse
import sys
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent.parent
if str(_PKG) not in sys.path:
    sys.path.insert(0, str(_PKG))

from shm_hip import simulate   # noqa: E402

FORCE_RMS, FORCE_SEED = 200.0, 42


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--seed", type=int, default=0, help="seed of the sensor-fault draws")
    a = ap.parse_args()
    root = Path(a.root)
    _, structural, sensor = simulate.fault_datasets(FORCE_RMS, FORCE_SEED, a.seed, dtype=torch.float64)
    for label, tab in structural.items():
        simulate.save_csv(tab, root / f"Data/raw/faults/structural_fault/{label}/{label}.csv")
        print(f"[OK] structural fault saved: {label}")
    for name, tab in sensor.items():
        simulate.save_csv(tab, root / f"Data/raw/faults/sensor_fault/{name}/{name}.csv")
        print(f"[OK] sensor fault saved: {name}")
    print("[SUCCESS] Fault datasets generated.")


if __name__ == "__main__":
    main()
