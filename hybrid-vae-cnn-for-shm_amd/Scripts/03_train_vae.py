"""Stage 1 on MI355X — counterpart of reference 4DOF/Scripts/03_train_vae.py (constants :27-47; plotting omitted).
    python -m Scripts.03_train_vae --root /path/to/4DOF [--precision split|fp32|bf16]
--root is a directory laid out like the reference's 4DOF/ (Data/processed/run_splits.json + the CSV runs)."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent.parent
if str(_PKG) not in sys.path:
    sys.path.insert(0, str(_PKG))

from Scripts.Models.cnn_model import NUM_FEATURES, SEQ_LEN   # noqa: E402
from shm_hip import dataprep, workflows                      # noqa: E402

TRAIN_FRAC, VAL_FRAC = (0.0, 0.4), (0.4, 0.7)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--precision", choices=["split", "fp32", "bf16"], default="split", help="arithmetic of the VAE kernels: split (default) = bf16 hi+lo operand pairs, holds the reference's fp32 results to 1e-4 at bf16 matrix rates; fp32 = exact-f32 MFMA; bf16 = throughput path with its own tolerances")
    a = ap.parse_args()
    root = Path(a.root)
    splits = json.loads((root / "Data/processed/run_splits.json").read_text())
    files = splits["normal"]["files"]
    Wtr = dataprep.fraction_windows(files, TRAIN_FRAC, SEQ_LEN, NUM_FEATURES, 1, root)
    Wva = dataprep.fraction_windows(files, VAL_FRAC, SEQ_LEN, NUM_FEATURES, 1, root)
    if Wtr.shape[0] == 0 or Wva.shape[0] == 0:
        raise RuntimeError("No normal train/val windows. Check the normal files and the fractions.")
    mean, std = dataprep.mean_std_from_windows(Wtr)
    proc = root / "Data/processed"; proc.mkdir(parents=True, exist_ok=True)
    np.save(proc / "vae_mean.npy", mean); np.save(proc / "vae_std.npy", std); np.savez(proc / "normal_stats.npz", mean=mean, std=std)
    print(f"[INFO] normal windows train/val = {Wtr.shape[0]}/{Wva.shape[0]} (T={SEQ_LEN}, D={NUM_FEATURES})")
    workflows.train_vae(dataprep.normalize_windows(Wtr, mean, std), dataprep.normalize_windows(Wva, mean, std), root / "models",
                        epochs=a.epochs, precision=a.precision)


if __name__ == "__main__":
    main()
