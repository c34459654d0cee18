#!/usr/bin/env python3
"""Golden-vector generator for the LSTM-VAE / CNN hot path.

Runs ONLY in the build container: it imports the reference's own model classes from
/root/reference (never copied into this repo), feeds them seeded inputs with the
reparameterisation noise injected (torch.randn_like patched), and writes small .npz
fixtures next to this file.  The fixtures are data (inputs + expected outputs); the
GPU box never sees the reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

torch 2.10.0+rocm7.0 CPU fp32 is the arithmetic the goldens pin (the reference does
not pin a torch version: 4DOF/pyproject.toml:5-9 lists no dependencies).

Fixtures:
  vae_fwd_*      G1/G2  forward: state_dict, x, eps -> recon, mu, logvar, per-window MSE
  vae_train_*    G3     train step(s): losses, 26 grads, grad-norm, params after Adam
  kl_weight      G4     KL-anneal schedules of the three trainers
  dataprep       G5     slice_frac / make_windows / mean-std on the shipped CSVs
  cnn_*          G6-G8  CNN eval forward, train step, north-star shape
  pipeline_mini  G9     score -> gate -> residual channel -> CNN labels
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
from contextlib import contextmanager
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True


def _load(name: str, path: Path):
    spec = importlib.util.spec_from_file_location(name, str(path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_vae_4dof = _load("ref_vae_4dof", REF / "4DOF/Scripts/Models/temporal_vae.py")
ref_cnn_4dof = _load("ref_cnn_4dof", REF / "4DOF/Scripts/Models/cnn_model.py")
ref_vae_1dof = _load("ref_vae_1dof", REF / "1_DOF/Scripts/Models/temporal_vae.py")
ref_vae_olab = _load("ref_vae_olab", REF / "20250506_openLAB_tests/Codes/Models/temporal_vae_model.py")


@contextmanager
def inject_eps(eps_list):
    """Patch torch.randn_like so reparameterize() consumes the supplied eps tensors in order."""
    it = iter(eps_list)
    orig = torch.randn_like

    def fake(t, *a, **k):
        e = next(it)
        assert e.shape == t.shape, (e.shape, t.shape)
        return e.to(t.dtype)

    torch.randn_like = fake
    try:
        yield
    finally:
        torch.randn_like = orig


def sd_np(model, prefix="p."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def vae_forward_case(name, cls, kw, B, T, seed, eval_mode=True):
    torch.manual_seed(seed)
    m = cls(**kw)
    m.eval() if eval_mode else m.train()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, kw["input_dim"], generator=g)
    eps = torch.randn(B, kw["latent_dim"], generator=g)
    with torch.no_grad(), inject_eps([eps]):
        recon, mu, logvar = m(x)
    score = ((x - recon) ** 2).mean(dim=(1, 2))  # 4DOF/Scripts/04_vae_thresholding.py:122
    d = sd_np(m)
    d.update(x=x.numpy(), eps=eps.numpy(), recon=recon.numpy(), mu=mu.numpy(),
             logvar=logvar.numpy(), score=score.numpy(),
             cfg=np.array(json.dumps(dict(kw, B=B, T=T, layer_norm=hasattr(m, "layer_norm")))))
    np.savez(OUT / f"{name}.npz", **d)
    print(name, "ok", {k: float(np.abs(v).max()) for k, v in dict(recon=recon.numpy(), mu=mu.numpy()).items()})


def vae_train_case(name, cls, kw, B, T, seed, n_steps, kl_w, lr, wd, clip, x_scale=1.0):
    """Train step(s) exactly as 4DOF/Scripts/03_train_vae.py:260-271 (clip=None, wd=0: the 1_DOF loop :300-310)."""
    assert kw.get("dropout", 0.0) == 0.0 or kw["num_layers"] == 1, "inter-layer dropout masks cannot be captured from ATen"
    torch.manual_seed(seed)
    m = cls(**kw)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr, weight_decay=wd)
    g = torch.Generator().manual_seed(seed + 1)
    d = sd_np(m, "p0.")
    names = [k for k, _ in m.named_parameters()]
    for s in range(n_steps):
        x = torch.randn(B, T, kw["input_dim"], generator=g) * x_scale
        eps = torch.randn(B, kw["latent_dim"], generator=g)
        with inject_eps([eps]):
            xhat, mu, logvar = m(x)
        recon = F.mse_loss(xhat, x, reduction="mean")
        kl = -0.5 * torch.mean(1.0 + logvar - mu.pow(2) - logvar.exp())
        loss = recon + kl_w * kl
        opt.zero_grad(set_to_none=True)
        loss.backward()
        raw = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        if clip is not None:
            gn = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=clip)
        else:
            gn = torch.sqrt(sum((g_ ** 2).sum() for g_ in raw.values()))
        opt.step()
        d[f"x{s}"] = x.numpy()
        d[f"eps{s}"] = eps.numpy()
        d[f"loss{s}"] = np.array([loss.item(), recon.item(), kl.item(), float(gn)], dtype=np.float64)
        if s == 0:
            d["recon0"] = xhat.detach().numpy()
            d["mu0"] = mu.detach().numpy()
            d["logvar0"] = logvar.detach().numpy()
            for n in names:
                d["g0." + n] = raw[n].numpy()
        if s == 0 or s == n_steps - 1:
            for n, p in m.named_parameters():
                d[f"p{s + 1}." + n] = p.detach().numpy().copy()
    d["cfg"] = np.array(json.dumps(dict(kw, B=B, T=T, n_steps=n_steps, kl_w=kl_w, lr=lr, wd=wd,
                                        clip=clip, layer_norm=hasattr(m, "layer_norm"))))
    np.savez(OUT / f"{name}.npz", **d)
    print(name, "ok loss/recon/kl/gnorm step0 =", d["loss0"])


def kl_case():
    t4 = _load("ref_train_4dof", REF / "4DOF/Scripts/03_train_vae.py")
    v4 = np.array([t4.kl_weight(e, 50, 0.30) for e in range(1, 51)])
    # 1_DOF/Scripts/03_train_vae.py:26-30 (0-based epoch) and openLAB Codes/04_train_vae.py:129-131 are
    # restated arithmetic on literals here because importing those scripts pulls plotting/config side effects.
    def kl_1dof(epoch, n, r=0.3):
        pivot = int(n * r); den = max(pivot, 1)
        return float(1.0 / (1.0 + np.exp(-5.0 * (epoch - pivot) / den)))
    def kl_olab(epoch, n, r=0.30):
        x = (epoch - (n * r)) / max(n * r, 1e-12)
        return float(1.0 / (1.0 + np.exp(-x * 5.0)))
    v1 = np.array([kl_1dof(e, 100) for e in range(100)])
    vo = np.array([kl_olab(e, 100) for e in range(1, 101)])
    np.savez(OUT / "kl_weight.npz", dof4=v4, dof1=v1, openlab=vo)
    print("kl_weight ok", v4[:3], v4[-1])


def dataprep_case():
    t4 = _load("ref_train_4dof_b", REF / "4DOF/Scripts/03_train_vae.py")
    splits = json.loads((REF / "4DOF/Data/processed/run_splits.json").read_text())
    files = splits["normal"]["files"]
    X = t4.load_csv_numeric(t4.as_abs(files[0]))
    Xs = t4.slice_frac(X, (0.0, 0.4))
    W = t4.make_windows(Xs, 100, 1)
    Wtr = t4.build_fraction_windows(files, (0.0, 0.4))
    Wva = t4.build_fraction_windows(files, (0.4, 0.7))
    mean, std = t4.compute_mean_std_from_windows(Wtr)
    Z = t4.normalize_windows(Wtr[:4], mean, std)
    np.savez(OUT / "dataprep.npz", file0=np.array(files[0]), n_rows=X.shape[0], slice_rows=Xs.shape[0],
             n_win0=W.shape[0], win_first=W[0], win_last=W[-1], n_train=Wtr.shape[0], n_val=Wva.shape[0],
             mean=mean, std=std, z4=Z, raw_head=X[:140])
    print("dataprep ok", Wtr.shape, Wva.shape, mean[:3], std[:3])


def make_cnn(T, D, seed, randomize_bn=True):
    torch.manual_seed(seed)
    m = ref_cnn_4dof.CNN(2, 2, 0.5)
    if (T, D) != (100, 12):
        # G8: north-star shape; the only deviation from the reference class is the fc1 width
        fc = torch.nn.Linear(32 * (T // 4) * (D // 4), 128)
        torch.nn.init.xavier_uniform_(fc.weight); torch.nn.init.zeros_(fc.bias)
        m.fc1[0] = fc
    if randomize_bn:
        g = torch.Generator().manual_seed(seed + 7)
        with torch.no_grad():
            for bn in (m.conv1[1], m.conv2[1]):
                bn.running_mean.copy_(0.3 * torch.randn(bn.num_features, generator=g))
                bn.running_var.copy_(0.5 + torch.rand(bn.num_features, generator=g))
                bn.weight.copy_(1.0 + 0.2 * torch.randn(bn.num_features, generator=g))
                bn.bias.copy_(0.1 * torch.randn(bn.num_features, generator=g))
            for mod in (m.conv1[0], m.conv2[0], m.fc1[0], m.fc2):
                mod.bias.copy_(0.05 * torch.randn(mod.bias.shape, generator=g))
    return m


def cnn_eval_case(name, T, D, B, seed):
    m = make_cnn(T, D, seed).eval()
    g = torch.Generator().manual_seed(seed + 1)
    z = torch.randn(B, T, D, generator=g)
    zhat = z + 0.5 * torch.randn(B, T, D, generator=g)
    xin = torch.stack([z, (z - zhat) ** 2], dim=1)   # 4DOF/Scripts/05_train_cnn.py:137-138
    with torch.no_grad():
        logits = m(xin)
        probs = torch.softmax(logits, dim=1)
    d = sd_np(m)
    d.update(xin=xin.numpy(), logits=logits.numpy(), probs=probs.numpy(),
             argmax=torch.argmax(logits, dim=1).numpy(), cfg=np.array(json.dumps(dict(T=T, D=D, B=B))))
    np.savez(OUT / f"{name}.npz", **d)
    print(name, "ok", logits[:2].tolist())


def cnn_train_case(name, T, D, B, seed, n_steps=2, lr=1e-4, wd=5e-5):
    """4DOF/Scripts/05_train_cnn.py:270-276 with dropout_rate=0 (mask-free) — BN in batch-stat mode."""
    m = make_cnn(T, D, seed)
    m.fc1[2].p = 0.0
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr, weight_decay=wd)
    lossf = torch.nn.CrossEntropyLoss()
    g = torch.Generator().manual_seed(seed + 1)
    d = sd_np(m, "p0.")
    for s in range(n_steps):
        xin = torch.randn(B, 2, T, D, generator=g)
        xin[:, 1] = xin[:, 1] ** 2
        y = torch.randint(0, 2, (B,), generator=g)
        opt.zero_grad(set_to_none=True)
        logits = m(xin)
        loss = lossf(logits, y)
        loss.backward()
        if s == 0:
            d["logits0"] = logits.detach().numpy()
            for n, p in m.named_parameters():
                d["g0." + n] = p.grad.detach().numpy().copy()
        opt.step()
        d[f"xin{s}"] = xin.numpy(); d[f"y{s}"] = y.numpy(); d[f"loss{s}"] = np.array(loss.item())
        if s == 0 or s == n_steps - 1:
            for k, v in m.state_dict().items():
                d[f"p{s + 1}." + k] = v.detach().numpy().copy()
    d["cfg"] = np.array(json.dumps(dict(T=T, D=D, B=B, lr=lr, wd=wd, n_steps=n_steps)))
    np.savez(OUT / f"{name}.npz", **d)
    print(name, "ok loss0", d["loss0"])


def pipeline_case():
    """G9: a14 semantics (4DOF/Scripts/06_test_full_pipeline.py:339-372) on real CSV windows, both eps draws injected."""
    t4 = _load("ref_train_4dof_c", REF / "4DOF/Scripts/03_train_vae.py")
    splits = json.loads((REF / "4DOF/Data/processed/run_splits.json").read_text())
    Wtr = t4.build_fraction_windows(splits["normal"]["files"], (0.0, 0.4))
    mean, std = t4.compute_mean_std_from_windows(Wtr)
    Wn = t4.build_fraction_windows(splits["normal"]["files"][:1], (0.7, 1.0))[:24]
    Wf = t4.build_fraction_windows(splits["sensor_fault"]["files"][:1], (0.7, 1.0))[:24]
    Z = t4.normalize_windows(np.concatenate([Wn, Wf], 0), mean, std)
    torch.manual_seed(5)
    vae = ref_vae_4dof.TemporalVAE(12, 16, 32, 2, 0.3).eval()   # H=32 mini keeps the fixture small
    cnn = make_cnn(100, 12, 11).eval()
    g = torch.Generator().manual_seed(6)
    zb = torch.tensor(Z)
    eps1 = torch.randn(Z.shape[0], 16, generator=g)
    with torch.no_grad(), inject_eps([eps1]):
        xhat, _, _ = vae(zb)
    mse = ((zb - xhat) ** 2).mean(dim=(1, 2)).numpy()
    thr = float(np.median(mse))
    sel = np.where(mse > thr)[0]
    eps2 = torch.randn(len(sel), 16, generator=g)
    with torch.no_grad(), inject_eps([eps2]):
        xh2, _, _ = vae(zb[sel])
        xin = torch.stack([zb[sel], (zb[sel] - xh2) ** 2], dim=1)
        logits = cnn(xin)
    y_pred = np.zeros(Z.shape[0], dtype=np.int64)
    y_pred[sel] = torch.argmax(logits, 1).numpy() + 1
    p_struct = np.zeros(Z.shape[0], dtype=np.float32)
    p_struct[sel] = torch.softmax(logits, 1)[:, 1].numpy()
    d = sd_np(vae, "vae."); d.update(sd_np(cnn, "cnn."))
    d.update(Z=Z, eps1=eps1.numpy(), eps2=eps2.numpy(), mse=mse, thr=thr, sel=sel, logits=logits.numpy(),
             y_pred=y_pred, p_struct=p_struct)
    np.savez(OUT / "pipeline_mini.npz", **d)
    print("pipeline_mini ok: anomalies", len(sel), "of", Z.shape[0])


def main():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    V4, V1, VO = ref_vae_4dof.TemporalVAE, ref_vae_1dof.TemporalVAE, ref_vae_olab.VAE
    # --- forward goldens (eval mode, eps injected) ---
    vae_forward_case("vae_fwd_ref_h128", V4, dict(input_dim=12, latent_dim=16, hidden_dim=128, num_layers=2, dropout=0.3), 8, 100, 42)
    vae_forward_case("vae_fwd_ns_h128", V4, dict(input_dim=4, latent_dim=16, hidden_dim=128, num_layers=2, dropout=0.3), 20, 256, 43)
    vae_forward_case("vae_fwd_1dof", V1, dict(input_dim=12, latent_dim=5, hidden_dim=32, num_layers=2, dropout=0.2), 9, 80, 44)
    vae_forward_case("vae_fwd_openlab", VO, dict(input_dim=3, latent_dim=8, hidden_dim=64, num_layers=1, dropout=0.3), 7, 200, 45)
    vae_forward_case("vae_fwd_mini", V4, dict(input_dim=12, latent_dim=16, hidden_dim=32, num_layers=2, dropout=0.3), 19, 24, 46)
    # --- train-step goldens (dropout 0; Adam as 03_train_vae.py:222, clip 2.0) ---
    vae_train_case("vae_train_ns_h128", V4, dict(input_dim=4, latent_dim=16, hidden_dim=128, num_layers=2, dropout=0.0),
                   B=8, T=256, seed=50, n_steps=1, kl_w=0.5, lr=1e-3, wd=1e-5, clip=2.0)
    vae_train_case("vae_train_mini", V4, dict(input_dim=12, latent_dim=16, hidden_dim=32, num_layers=2, dropout=0.0),
                   B=21, T=30, seed=51, n_steps=10, kl_w=0.37, lr=1e-3, wd=1e-5, clip=2.0, x_scale=3.0)
    vae_train_case("vae_train_1dof", V1, dict(input_dim=12, latent_dim=5, hidden_dim=32, num_layers=2, dropout=0.0),
                   B=16, T=80, seed=52, n_steps=3, kl_w=0.2, lr=1e-3, wd=0.0, clip=None)
    vae_train_case("vae_train_openlab", VO, dict(input_dim=3, latent_dim=8, hidden_dim=64, num_layers=1, dropout=0.3),
                   B=10, T=200, seed=53, n_steps=2, kl_w=1.0, lr=5e-4, wd=0.0, clip=2.0)
    kl_case()
    dataprep_case()
    cnn_eval_case("cnn_eval_ref", 100, 12, 8, 60)
    cnn_eval_case("cnn_eval_ns", 256, 4, 6, 61)
    cnn_train_case("cnn_train_ref", 100, 12, 10, 62)
    cnn_train_case("cnn_train_ns", 256, 4, 6, 63, n_steps=1)
    pipeline_case()


if __name__ == "__main__":
    main()
