"""The four stages of the 4DOF hybrid pipeline on the MI355X path (loop semantics of the reference scripts, no plotting).

  train_vae        4DOF/Scripts/03_train_vae.py:177-354   (epoch loop :254-331)
  vae_threshold    4DOF/Scripts/04_vae_thresholding.py:232-311
  train_cnn        4DOF/Scripts/05_train_cnn.py:174-313
  test_pipeline    4DOF/Scripts/06_test_full_pipeline.py:276-383 (metrics reduced to confusion counts; sklearn reporting is out of scope)

Windows are prepared on the host (shm_hip.dataprep), moved to the device ONCE, and every step runs through the fused trainers;
the only host synchronisation is one 4-float read per step for the epoch bookkeeping the reference prints.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

from . import dataprep
from .cnn import build_cnn_input
from .pipeline import hybrid_predict, score_windows
from .schedules import kl_weight_4dof
from .trainer import CNNTrainer, VAETrainer


def _vae_class():
    from Scripts.Models.temporal_vae import TemporalVAE
    return TemporalVAE


def _cnn_class():
    from Scripts.Models.cnn_model import CNN
    return CNN


def train_vae(Ztr: np.ndarray, Zva: np.ndarray, out_dir, *, epochs=50, batch_size=256, lr=1e-3, weight_decay=1e-5,
              latent_dim=16, hidden_dim=128, num_layers=2, dropout=0.3, kl_warmup_ratio=0.30, seed=42, precision="fp32",
              device="cuda", log=print):
    """03_train_vae.py:212-354.  Returns (model, history).  Best-validation state_dict -> out_dir/temporal_vae_state_dict.pt."""
    out_dir = Path(out_dir); out_dir.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(seed); np.random.seed(seed)
    D = Ztr.shape[2]
    vae = _vae_class()(D, latent_dim, hidden_dim, num_layers, dropout, precision=precision).to(device)
    tr = VAETrainer(vae, lr=lr, weight_decay=weight_decay, max_norm=2.0, seed=seed)
    xtr, xva = torch.from_numpy(Ztr).to(device), torch.from_numpy(Zva).to(device)
    gen = torch.Generator(); gen.manual_seed(seed)                 # DataLoader(shuffle=True, generator=g), 03:224-233
    hist = {k: [] for k in ("epoch", "train_total", "train_recon", "train_kl", "val_total", "val_recon", "val_kl", "kl_w")}
    best_val, best_epoch = float("inf"), -1
    for epoch in range(1, epochs + 1):
        kl_w = kl_weight_4dof(epoch, epochs, kl_warmup_ratio)
        vae.train()
        perm = torch.randperm(xtr.size(0), generator=gen).to(device)
        sums, n = np.zeros(3), 0
        for i in range(0, xtr.size(0), batch_size):               # drop_last=False: the partial last batch is kept
            xb = xtr.index_select(0, perm[i:i + batch_size])
            s = tr.step(xb, kl_w).cpu().numpy()                    # sample-weighted epoch means, 03:274-282
            tr.check(s)                                            # sticky hand-off status word rides in s[4]: raises, no extra sync
            sums += s[:3] * xb.size(0); n += xb.size(0)
        tr_avg = sums / max(n, 1)
        vae.eval()
        sums, n = np.zeros(3), 0
        for i in range(0, xva.size(0), batch_size):
            xb = xva[i:i + batch_size]
            s = tr.evaluate(xb, kl_w).cpu().numpy()               # validation uses the same kl_w, 03:293
            tr.check_eval(s)
            sums += s[:3] * xb.size(0); n += xb.size(0)
        va_avg = sums / max(n, 1)
        for k, v in zip(("train_total", "train_recon", "train_kl"), tr_avg): hist[k].append(float(v))
        for k, v in zip(("val_total", "val_recon", "val_kl"), va_avg): hist[k].append(float(v))
        hist["epoch"].append(epoch); hist["kl_w"].append(float(kl_w))
        log(f"[train] epoch {epoch:03d}/{epochs} | kl_w={kl_w:.6f} | total={tr_avg[0]:.6f} | recon={tr_avg[1]:.6f} | kl={tr_avg[2]:.6f}")
        log(f"[val  ] epoch {epoch:03d}/{epochs} | kl_w={kl_w:.6f} | total={va_avg[0]:.6f} | recon={va_avg[1]:.6f} | kl={va_avg[2]:.6f}")
        if kl_w > 0.99 and tr_avg[2] < 1e-4:                        # posterior collapse: the decoder ignores z (KL ~ 0 at full weight)
            log(f"[warn ] epoch {epoch:03d}: KL {tr_avg[2]:.2e} at kl_w {kl_w:.3f} — posterior collapse (reconstruction from the prior only)")
            hist.setdefault("collapsed_epochs", []).append(epoch)
        if va_avg[0] < best_val:
            best_val, best_epoch = float(va_avg[0]), epoch
            torch.save(vae.state_dict(), out_dir / "temporal_vae_state_dict.pt")
    meta = dict(seed=seed, epochs=epochs, batch_size=batch_size, latent_dim=latent_dim, hidden_dim=hidden_dim, num_layers=num_layers,
                dropout=dropout, kl_warmup_ratio=kl_warmup_ratio, best_val_total=best_val, best_epoch=best_epoch, precision=precision)
    (out_dir / "stage1_vae_train_meta.json").write_text(json.dumps(meta, indent=2))
    return vae, hist


def train_vae_1dof(Ztr: np.ndarray, out_dir, *, n_epochs=100, batch_size=64, lr=1e-3, latent_dim=5, hidden_dim=32, num_layers=2, dropout=0.2,
                   anneal_ratio=0.3, seed=0, precision="fp32", device="cuda", log=print):
    """1_DOF/Scripts/03_train_vae.py:262-335: the 1_DOF trainer differs from the 4DOF one in every knob around the same step — model
    without LayerNorm, Adam without weight decay, no gradient clipping, 0-based epoch in the KL schedule, the epoch average taken over
    BATCHES (sum of batch means / number of batches, :316-318), no validation pass, final (not best) weights saved."""
    from .schedules import kl_weight_1dof
    out_dir = Path(out_dir); out_dir.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(seed); np.random.seed(seed)
    vae = _vae_class()(Ztr.shape[2], latent_dim, hidden_dim, num_layers, dropout, layer_norm=False, precision=precision).to(device)
    tr = VAETrainer(vae, lr=lr, weight_decay=0.0, max_norm=None, seed=seed)
    xtr = torch.from_numpy(Ztr).to(device)
    hist = {k: [] for k in ("epoch", "loss_total", "loss_recon", "loss_kl", "kl_weight")}
    for epoch in range(n_epochs):
        w = kl_weight_1dof(epoch, n_epochs, anneal_ratio)
        vae.train()
        perm = torch.randperm(xtr.size(0)).to(device)                 # DataLoader(shuffle=True) on the global torch RNG
        sums, nb = np.zeros(3), 0
        for i in range(0, xtr.size(0), batch_size):
            s = tr.step(xtr.index_select(0, perm[i:i + batch_size]), w).cpu().numpy()
            tr.check(s)
            sums += s[:3]; nb += 1
        avg = sums / max(nb, 1)
        for k, v in zip(("loss_total", "loss_recon", "loss_kl"), avg): hist[k].append(float(v))
        hist["epoch"].append(epoch + 1); hist["kl_weight"].append(float(w))
        if (epoch + 1) % 10 == 0:
            log(f"[epoch {epoch + 1:3d}/{n_epochs}] loss={avg[0]:.6f} recon={avg[1]:.6f} kl={avg[2]:.6f} w={w:.3f}")
    torch.save(vae.state_dict(), out_dir / "temporal_vae_state_dict.pt")
    return vae, hist


def vae_threshold(vae, Z_normal_val: np.ndarray, out_dir, *, percentile=99.0, batch=512, device="cuda", extra=None):
    """04_vae_thresholding.py:113-124,283,293-311: threshold = P99 of the normal-validation per-window MSE."""
    scores = score_windows(vae, torch.from_numpy(Z_normal_val).to(device), batch).cpu().numpy()
    thr = float(np.percentile(scores, percentile))
    out = dict(threshold=thr, score_def="mean over (T, D) of squared reconstruction error per window", percentile=percentile,
               score_summary=dict(normal_val=dict(n=int(scores.size), mean=float(scores.mean()), std=float(scores.std()))))
    if extra:
        out["score_summary"].update(extra)
    Path(out_dir).mkdir(parents=True, exist_ok=True)
    (Path(out_dir) / "vae_threshold.json").write_text(json.dumps(out, indent=2))
    return thr, scores


def build_cnn_inputs(vae, Z: np.ndarray, *, batch=512, device="cuda") -> torch.Tensor:
    """05_train_cnn.py:118-141 — frozen eval VAE, xin = stack([z, (z - zhat)^2]); stays on the device."""
    vae.eval()
    z = torch.from_numpy(Z).to(device)
    out = torch.empty(z.size(0), 2, z.size(1), z.size(2), dtype=torch.float32, device=device)
    with torch.no_grad():
        for i in range(0, z.size(0), batch):
            zb = z[i:i + batch]
            zhat, _, _ = vae(zb)
            out[i:i + batch] = build_cnn_input(zb, zhat)
    return out


def train_cnn(Xtr: torch.Tensor, ytr: np.ndarray, Xva: torch.Tensor, yva: np.ndarray, out_dir, *, epochs=100, batch_size=100,
              lr=1e-4, weight_decay=5e-5, dropout=0.5, patience=15, seed=42, log=print):
    """05_train_cnn.py:255-313: Adam + CrossEntropy, shuffle each epoch, early stopping on the validation loss."""
    out_dir = Path(out_dir); out_dir.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(seed)
    dev = Xtr.device
    model = _cnn_class()(2, 2, dropout, seq_len=Xtr.size(2), num_features=Xtr.size(3)).to(dev)
    tr = CNNTrainer(model, lr=lr, weight_decay=weight_decay, seed=seed)
    yt, yv = torch.from_numpy(ytr.astype(np.int64)).to(dev), torch.from_numpy(yva.astype(np.int64)).to(dev)
    best_val, best_state, no_improve, hist = float("inf"), None, 0, []
    for ep in range(1, epochs + 1):
        model.train()
        perm = torch.randperm(Xtr.size(0)).to(dev)                # DataLoader(shuffle=True) on the global torch RNG, 05:252
        tot, n = 0.0, 0
        for i in range(0, Xtr.size(0), batch_size):
            sel = perm[i:i + batch_size]
            loss = float(tr.step(Xtr.index_select(0, sel), yt.index_select(0, sel))[0])
            tot += loss * sel.numel(); n += sel.numel()
        vt, vn = 0.0, 0
        for i in range(0, Xva.size(0), batch_size):
            nb = min(batch_size, Xva.size(0) - i)
            vt += float(tr.evaluate(Xva[i:i + batch_size], yv[i:i + batch_size])) * nb; vn += nb
        tr_avg, va_avg = tot / max(n, 1), vt / max(vn, 1)
        hist.append((tr_avg, va_avg))
        log(f"[cnn] epoch {ep:03d}/{epochs} train={tr_avg:.6f} val={va_avg:.6f}")
        if va_avg < best_val:
            best_val, no_improve = va_avg, 0
            best_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        else:
            no_improve += 1
            if no_improve >= patience:
                log(f"[cnn] early stopping at epoch {ep}")
                break
    if best_state is not None:
        model.load_state_dict(best_state)
    torch.save(model.state_dict(), out_dir / "cnn_state_dict.pt")
    (out_dir / "stage2_cnn_train_meta.json").write_text(json.dumps(dict(best_val_loss=best_val, epochs_run=len(hist)), indent=2))
    return model, hist


def test_pipeline(vae, cnn, groups: dict, threshold: float, *, batch=512, device="cuda"):
    """06_test_full_pipeline.py:327-383 for groups = {"normal": Z0, "sensor": Z1, "structural": Z2} (ground truth 0/1/2).
    Returns the 3x3 confusion matrix (rows = truth) and the per-group gate statistics."""
    conf = np.zeros((3, 3), dtype=np.int64)
    gate = {}
    for gt, name in enumerate(("normal", "sensor", "structural")):
        Z = groups.get(name)
        if Z is None or Z.shape[0] == 0:
            continue
        mse, y_pred, _ = hybrid_predict(vae, cnn, torch.from_numpy(Z).to(device), threshold, batch)
        yp = y_pred.cpu().numpy()
        for c in range(3):
            conf[gt, c] += int((yp == c).sum())
        anom = int((mse > threshold).sum())
        gate[name] = dict(anom=anom, total=int(Z.shape[0]), anom_rate=anom / max(Z.shape[0], 1))
    return conf, gate
