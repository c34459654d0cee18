"""Stage 1b on MI355X — counterpart of reference 4DOF/Scripts/04_vae_thresholding.py (P99 of normal-val scores; plots omitted).
    python -m Scripts.04_vae_thresholding --root /path/to/4DOF"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

_PKG = Path(__file__).resolve().parent.parent
if str(_PKG) not in sys.path:
    sys.path.insert(0, str(_PKG))

from Scripts.Models.cnn_model import NUM_FEATURES, SEQ_LEN   # noqa: E402
from Scripts.Models.temporal_vae import TemporalVAE           # noqa: E402
from shm_hip import dataprep, workflows                      # noqa: E402

HEALTHY_FRAC = (0.4, 0.7)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--root", required=True)
    ap.add_argument("--precision", choices=["split", "fp32", "bf16"], default="split", help="arithmetic of the VAE kernels: split (default) = bf16 hi+lo operand pairs, holds the reference's fp32 results to 1e-4 at bf16 matrix rates; fp32 = exact-f32 MFMA; bf16 = throughput path with its own tolerances")
    a = ap.parse_args()
    root = Path(a.root)
    splits = json.loads((root / "Data/processed/run_splits.json").read_text())
    st = np.load(root / "Data/processed/normal_stats.npz")
    vae = TemporalVAE(NUM_FEATURES, 16, 128, 2, 0.3, precision=a.precision).to("cuda")
    vae.load_state_dict(torch.load(root / "models/temporal_vae_state_dict.pt", map_location="cuda"))
    Zn = dataprep.normalize_windows(dataprep.fraction_windows(splits["normal"]["files"], HEALTHY_FRAC, SEQ_LEN, NUM_FEATURES, 1, root), st["mean"], st["std"])
    thr, scores = workflows.vae_threshold(vae, Zn, root / "Data/processed")
    print(f"[OK] threshold (P99 normal-val) = {thr:.6f} over {scores.size} windows")


if __name__ == "__main__":
    main()
