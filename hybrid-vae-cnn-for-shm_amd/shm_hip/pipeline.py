"""Hybrid VAE-gate -> CNN inference on the device (reference 4DOF/Scripts/06_test_full_pipeline.py:327-383).

score windows with the VAE -> anomalous = mse > threshold -> for the anomalies: a SECOND VAE pass (a fresh eps draw, exactly
as the reference does at :362-363), residual channel, CNN, class = argmax + 1, p_struct = softmax[:, 1].
torch is used for the boolean gate / gather only (index plumbing); all arithmetic runs in libshm_hip.so.
"""
from __future__ import annotations

import torch

from .cnn import build_cnn_input, predict


def _check_status(vae):
    """Raise if a layer-pipelined launch of the module's engine ever timed out on a hand-off (sticky word, shm_vae_plan_status)."""
    eng = getattr(vae, "_engine", None)
    if eng is not None:
        eng.check_status()


@torch.no_grad()
def score_windows(vae, Z: torch.Tensor, batch: int = 512) -> torch.Tensor:
    """Per-window MSE of 04_vae_thresholding.py:113-124 for Z [N,T,D] (device tensor), in batches."""
    vae.eval()
    out = torch.empty(Z.size(0), dtype=torch.float32, device=Z.device)
    for i in range(0, Z.size(0), batch):
        out[i:i + batch] = vae.score(Z[i:i + batch])
    _check_status(vae)              # one synchronisation per call (the scores go to the host next): a hand-off timeout raises here
    return out


@torch.no_grad()
def hybrid_predict(vae, cnn, Z: torch.Tensor, threshold: float, batch: int = 512, reuse_recon: bool = False):
    """Returns (mse [N], y_pred [N] in {0 normal, 1 sensor, 2 structural}, p_struct [N]) — 06_test_full_pipeline.py:339-372.

    reuse_recon=False (default) is the reference's behaviour: the anomalous windows go through the VAE a second time, with a new
    epsilon draw, to form the classifier's residual channel.  reuse_recon=True scores and reconstructs in ONE pass per batch and
    builds the residual channel from that same reconstruction: one VAE forward per window instead of up
    to two; the gate decision is identical in distribution, the classifier sees the draw that tripped the gate."""
    vae.eval(); cnn.eval()
    N = Z.size(0)
    y_pred = torch.zeros(N, dtype=torch.int64, device=Z.device)
    p_struct = torch.zeros(N, dtype=torch.float32, device=Z.device)
    if reuse_recon:
        # scores and reconstructions of every window first (no host synchronisation inside the loop; the reconstructions cost one more
        # copy of the window set in HBM), then the classifier over the anomalous ones in full batches
        mse = torch.empty(N, dtype=torch.float32, device=Z.device)
        recon = torch.empty_like(Z)
        for i in range(0, N, batch):
            mse[i:i + batch], recon[i:i + batch] = vae.score_and_reconstruct(Z[i:i + batch])
        idx = torch.nonzero(mse > threshold, as_tuple=False).flatten()
        for j in range(0, idx.numel(), batch):
            sel = idx[j:j + batch]
            logits = cnn(build_cnn_input(Z.index_select(0, sel), recon.index_select(0, sel)))
            cls, p1 = predict(logits)
            y_pred.index_copy_(0, sel, cls + 1)
            p_struct.index_copy_(0, sel, p1)
        _check_status(vae)
        return mse, y_pred, p_struct
    mse = score_windows(vae, Z, batch)
    idx = torch.nonzero(mse > threshold, as_tuple=False).flatten()
    for j in range(0, idx.numel(), batch):
        sel = idx[j:j + batch]
        zb = Z.index_select(0, sel)
        xhat, _, _ = vae(zb)                                   # second pass, new eps (reference behaviour)
        logits = cnn(build_cnn_input(zb, xhat))
        cls, p1 = predict(logits)
        y_pred.index_copy_(0, sel, cls + 1)
        p_struct.index_copy_(0, sel, p1)
    _check_status(vae)
    return mse, y_pred, p_struct


# ---- batched across the GPUs of a node (SURVEY.md §8e "Inference (config 5)", §8f rank 1): windows are sharded r, r+G, r+2G, ...
# over the ranks, each rank scores / classifies its shard, and the small result vectors (N fp32 scores, N int64 labels, N fp32
# probabilities) are gathered so that EVERY rank holds the full result — the percentile threshold and the metrics need all of them.
# No collective touches window data.  Works on any initialised torch.distributed backend ("nccl" = RCCL on the GPUs, "gloo" in the
# CPU tests); without an initialised process group it degenerates to the single-GPU functions above.
def _world(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(group), dist.get_world_size(group)
    return None, 0, 1


def _gather_sharded(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """local = results of items rank, rank+G, ... ; returns the [n_total] vector in the original item order on every rank."""
    dist, rank, world = _world(group)
    if world == 1:
        return local
    per = (n_total + world - 1) // world
    pad = torch.zeros(per, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.empty(n_total, dtype=local.dtype, device=local.device)
    for r in range(world):
        n_r = len(range(r, n_total, world))
        out[r::world] = parts[r][:n_r]
    return out


def score_windows_sharded(vae, Z: torch.Tensor, batch: int = 512, group=None) -> torch.Tensor:
    """Per-window MSE of ALL N windows, every rank computing N/G of them (04_vae_thresholding.py:113-124 at node scale)."""
    _, rank, world = _world(group)
    local = score_windows(vae, Z[rank::world].contiguous(), batch) if world > 1 else score_windows(vae, Z, batch)
    return _gather_sharded(local, Z.size(0), group)


def hybrid_predict_sharded(vae, cnn, Z: torch.Tensor, threshold: float, batch: int = 512, group=None, reuse_recon: bool = False):
    """hybrid_predict over the ranks of `group`; returns (mse, y_pred, p_struct) of all N windows on every rank."""
    _, rank, world = _world(group)
    if world == 1:
        return hybrid_predict(vae, cnn, Z, threshold, batch, reuse_recon)
    mse, y, p = hybrid_predict(vae, cnn, Z[rank::world].contiguous(), threshold, batch, reuse_recon)
    n = Z.size(0)
    return _gather_sharded(mse, n, group), _gather_sharded(y, n, group), _gather_sharded(p, n, group)


def percentile_threshold(scores: torch.Tensor, percentile: float = 99.0) -> float:
    """np.percentile(scores, p) (linear interpolation), 04_vae_thresholding.py:283 / openLAB P95 — on the gathered scores."""
    import numpy as np
    return float(np.percentile(scores.detach().cpu().numpy(), percentile))


# ---- openLAB hybrid inference (20250506_openLAB_tests/Codes/10_test_hybrid_pipeline.py:241-251, 266-302, 391-409):
# stage 1: VAE reconstruction MSE on the VAE-standardised windows, anomalous = mse > vae_thr; stage 2: the anomalous windows,
# standardised with the CNN's own statistics, go through the openLAB classifier as [n, 1, T, F]; prob_st = softmax[:, 1];
# structural-fault iff prob_st >= cnn_thr.  Labels: 0 normal, 1 sensor fault (SF), 2 structural fault (ST).
@torch.no_grad()
def hybrid_predict_openlab(vae, cnn, Z_vae: torch.Tensor, X_cnn: torch.Tensor, vae_thr: float, cnn_thr: float, batch: int = 512):
    """Z_vae [N,T,Dv], X_cnn [N,T,F]: device tensors already standardised for the VAE / the CNN (dataprep.standardize_openlab or
    dataprep.device_windows(..., clip_z=...)).  Returns (mse [N], y_pred [N] int64, prob_st [N] fp32, 0 where not anomalous)."""
    vae.eval(); cnn.eval()
    N = Z_vae.size(0)
    mse = score_windows(vae, Z_vae, batch)
    y_pred = torch.zeros(N, dtype=torch.int64, device=Z_vae.device)
    prob_st = torch.zeros(N, dtype=torch.float32, device=Z_vae.device)
    idx = torch.nonzero(mse > vae_thr, as_tuple=False).flatten()
    for j in range(0, idx.numel(), batch):
        sel = idx[j:j + batch]
        logits = cnn(X_cnn.index_select(0, sel).unsqueeze(1))
        _, p1 = predict(logits)                                   # softmax[:, 1] in libshm_hip.so
        y_pred.index_copy_(0, sel, 1 + (p1 >= cnn_thr).to(torch.int64))
        prob_st.index_copy_(0, sel, p1)
    return mse, y_pred, prob_st


def hybrid_predict_openlab_sharded(vae, cnn, Z_vae, X_cnn, vae_thr, cnn_thr, batch: int = 512, group=None):
    _, rank, world = _world(group)
    if world == 1:
        return hybrid_predict_openlab(vae, cnn, Z_vae, X_cnn, vae_thr, cnn_thr, batch)
    mse, y, p = hybrid_predict_openlab(vae, cnn, Z_vae[rank::world].contiguous(), X_cnn[rank::world].contiguous(), vae_thr, cnn_thr, batch)
    n = Z_vae.size(0)
    return _gather_sharded(mse, n, group), _gather_sharded(y, n, group), _gather_sharded(p, n, group)
