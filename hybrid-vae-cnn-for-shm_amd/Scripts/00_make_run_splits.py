"""Counterpart of reference 4DOF/Scripts/00_make_run_splits.py: scan <root>/Data/raw and write Data/processed/run_splits.json.
    python -m Scripts.00_make_run_splits --root /path/to/4DOF"""
import argparse
import sys
from pathlib import Path

_PKG = Path(__file__).resolve().parent.parent
if str(_PKG) not in sys.path:
    sys.path.insert(0, str(_PKG))

from Scripts.Models.cnn_model import SEQ_LEN   # noqa: E402
from shm_hip import dataprep                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    a = ap.parse_args()
    out = dataprep.make_run_splits(a.root, seq_len=SEQ_LEN, stride=1)
    print(f"[OK] wrote: {Path(a.root) / 'Data/processed/run_splits.json'}  totals={out['totals']}")


if __name__ == "__main__":
    main()
