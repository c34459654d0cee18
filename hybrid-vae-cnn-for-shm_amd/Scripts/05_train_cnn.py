"""Stage 2 on MI355X — counterpart of reference 4DOF/Scripts/05_train_cnn.py (constants :28-44; plots omitted).
    python -m Scripts.05_train_cnn --root /path/to/4DOF"""
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


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--root", required=True); ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--precision", choices=["split", "fp32", "bf16"], default="split", help="arithmetic of the VAE kernels: split (default) = bf16 hi+lo operand pairs, holds the reference's fp32 results to 1e-4 at bf16 matrix rates; fp32 = exact-f32 MFMA; bf16 = throughput path with its own tolerances")
    a = ap.parse_args()
    root = Path(a.root)
    splits = json.loads((root / "Data/processed/run_splits.json").read_text())
    if splits.get("mode") != "window_level_per_file":
        raise RuntimeError("run_splits.json must be window_level_per_file")
    st = np.load(root / "Data/processed/normal_stats.npz")
    vae = TemporalVAE(NUM_FEATURES, 16, 128, 2, 0.3, precision=a.precision).to("cuda")
    vae.load_state_dict(torch.load(root / "models/temporal_vae_state_dict.pt", map_location="cuda"))
    data = {}
    for split in ("train", "val"):
        Ws = dataprep.windows_by_index(splits["sensor_fault"]["files"], splits["sensor_fault"]["window_indices"], split, SEQ_LEN, NUM_FEATURES, root)
        Wt = dataprep.windows_by_index(splits["structural_fault"]["files"], splits["structural_fault"]["window_indices"], split, SEQ_LEN, NUM_FEATURES, root)
        if Ws.shape[0] == 0 or Wt.shape[0] == 0:
            raise RuntimeError(f"empty {split} split")
        Z = dataprep.normalize_windows(np.concatenate([Ws, Wt], 0), st["mean"], st["std"])
        y = np.concatenate([np.zeros(Ws.shape[0], np.int64), np.ones(Wt.shape[0], np.int64)])     # sensor = 0, structural = 1
        perm = np.random.default_rng(42).permutation(Z.shape[0])
        data[split] = (workflows.build_cnn_inputs(vae, Z[perm]), y[perm])
    workflows.train_cnn(data["train"][0], data["train"][1], data["val"][0], data["val"][1], root / "models", epochs=a.epochs)


if __name__ == "__main__":
    main()
