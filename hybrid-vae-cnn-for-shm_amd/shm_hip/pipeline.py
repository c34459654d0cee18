"""Hybrid VAE-gate -> CNN inference on the device (reference 4DOF/Scripts/06_test_full_pipeline.py:327-383).

score windows with the VAE -> anomalous = mse > threshold -> for the anomalies: a SECOND VAE pass (a fresh eps draw, exactly
as the reference does at :362-363), residual channel, CNN, class = argmax + 1, p_struct = softmax[:, 1].
torch is used for the boolean gate / gather only (index plumbing); all arithmetic runs in libshm_hip.so.
"""
from __future__ import annotations

import torch

from .cnn import build_cnn_input, predict


@torch.no_grad()
def score_windows(vae, Z: torch.Tensor, batch: int = 512) -> torch.Tensor:
    """Per-window MSE of 04_vae_thresholding.py:113-124 for Z [N,T,D] (device tensor), in batches."""
    vae.eval()
    out = torch.empty(Z.size(0), dtype=torch.float32, device=Z.device)
    for i in range(0, Z.size(0), batch):
        out[i:i + batch] = vae.score(Z[i:i + batch])
    return out


@torch.no_grad()
def hybrid_predict(vae, cnn, Z: torch.Tensor, threshold: float, batch: int = 512):
    """Returns (mse [N], y_pred [N] in {0 normal, 1 sensor, 2 structural}, p_struct [N]) — 06_test_full_pipeline.py:339-372."""
    vae.eval(); cnn.eval()
    N = Z.size(0)
    mse = score_windows(vae, Z, batch)
    y_pred = torch.zeros(N, dtype=torch.int64, device=Z.device)
    p_struct = torch.zeros(N, dtype=torch.float32, device=Z.device)
    idx = torch.nonzero(mse > threshold, as_tuple=False).flatten()
    for j in range(0, idx.numel(), batch):
        sel = idx[j:j + batch]
        zb = Z.index_select(0, sel)
        xhat, _, _ = vae(zb)                                   # second pass, new eps (reference behaviour)
        logits = cnn(build_cnn_input(zb, xhat))
        cls, p1 = predict(logits)
        y_pred.index_copy_(0, sel, cls + 1)
        p_struct.index_copy_(0, sel, p1)
    return mse, y_pred, p_struct
