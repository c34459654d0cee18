"""numpy restatement of the openLAB fault classifier, eval mode — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Follows reference 20250506_openLAB_tests/Codes/Models/cnn_model.py:14-62:
  block(cin, cout, kt, kf=3, pad (kt//2, 1)) = Conv2d -> GroupNorm(8 groups, eps 1e-5, biased variance) -> SiLU        :17-22
  features = block(1,32,7) MaxPool(2,1) block(32,64,5) MaxPool(2,1) block(64,128,5) MaxPool(2,1) block(128,256,3) AdaptiveAvgPool(1,1)   :24-36
  classifier = Flatten, Linear(256,128), SiLU, Dropout (identity in eval), Linear(128, num_classes)                    :38-44
Only tests/ import this.  Pinned by tests/golden/cnn_openlab_eval_t{200,512}.npz (generated from the reference class).
"""
from __future__ import annotations

import numpy as np

BLOCKS = ((0, 1, 32, 7), (2, 32, 64, 5), (4, 64, 128, 5), (6, 128, 256, 3))     # (features index, cin, cout, kt)
PARAMS = [f"features.{i}.{j}.{w}" for i, _, _, _ in BLOCKS for j in (0, 1) for w in ("weight", "bias")] + \
         ["classifier.1.weight", "classifier.1.bias", "classifier.4.weight", "classifier.4.bias"]


def _silu(x):
    return x / (1.0 + np.exp(-x))


def _conv(x, w, b, pt, pf):
    """x [B,C,H,W], w [O,C,kt,kf] -> [B,O,H,W] (stride 1, zero padding (pt, pf))."""
    B, C, H, W = x.shape
    O, _, kt, kf = w.shape
    xp = np.zeros((B, C, H + 2 * pt, W + 2 * pf), dtype=x.dtype)
    xp[:, :, pt:pt + H, pf:pf + W] = x
    y = np.zeros((B, O, H, W), dtype=x.dtype)
    for dt in range(kt):
        for df in range(kf):
            y += np.einsum("bchw,oc->bohw", xp[:, :, dt:dt + H, df:df + W], w[:, :, dt, df], optimize=True)
    return y + b[None, :, None, None]


def _group_norm(y, gamma, beta, groups=8, eps=1e-5):
    B, C, H, W = y.shape
    g = y.reshape(B, groups, C // groups, H, W)
    mean = g.mean(axis=(2, 3, 4), keepdims=True)
    var = g.var(axis=(2, 3, 4), keepdims=True)
    return ((g - mean) / np.sqrt(var + eps)).reshape(B, C, H, W) * gamma[None, :, None, None] + beta[None, :, None, None]


def forward(p: dict, x: np.ndarray) -> np.ndarray:
    """p: state_dict (numpy), x [B,1,T,W] -> logits [B, num_classes]; fp64 throughout."""
    a = x.astype(np.float64)
    for bi, (i, _, _, kt) in enumerate(BLOCKS):
        y = _conv(a, p[f"features.{i}.0.weight"].astype(np.float64), p[f"features.{i}.0.bias"].astype(np.float64), kt // 2, 1)
        y = _silu(_group_norm(y, p[f"features.{i}.1.weight"].astype(np.float64), p[f"features.{i}.1.bias"].astype(np.float64)))
        if bi < 3:
            H2 = y.shape[2] // 2
            a = np.maximum(y[:, :, 0:2 * H2:2, :], y[:, :, 1:2 * H2:2, :])        # MaxPool2d((2,1)), floor
        else:
            a = y.mean(axis=(2, 3))                                               # AdaptiveAvgPool2d((1,1)) + Flatten
    h = _silu(a @ p["classifier.1.weight"].astype(np.float64).T + p["classifier.1.bias"])
    return h @ p["classifier.4.weight"].astype(np.float64).T + p["classifier.4.bias"]
