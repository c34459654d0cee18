"""Pins oracle/vae_oracle.py against goldens generated from the reference's own classes (CPU-only tests)."""
import numpy as np
import pytest

from conftest import load_golden, split_prefix
from oracle.vae_oracle import (VAEOracle, kl_weight_1dof, kl_weight_4dof, kl_weight_openlab)

FWD = ["vae_fwd_ref_h128", "vae_fwd_ns_h128", "vae_fwd_1dof", "vae_fwd_openlab", "vae_fwd_mini"]
TRAIN = ["vae_train_ns_h128", "vae_train_mini", "vae_train_1dof", "vae_train_openlab"]


@pytest.mark.parametrize("name", FWD)
def test_forward_matches_reference(name):
    g = load_golden(name)
    cfg = g["cfg"]
    o = VAEOracle(split_prefix(g, "p."), cfg["num_layers"], cfg["layer_norm"])
    recon, mu, lv, score = o.forward(g["x"], g["eps"])
    # fp64 restatement vs torch fp32: agreement is limited by the reference's own fp32 rounding
    assert np.abs(recon - g["recon"]).max() < 5e-6
    assert np.abs(mu - g["mu"]).max() < 5e-6
    assert np.abs(lv - g["logvar"]).max() < 5e-6
    assert np.abs(score - g["score"]).max() < 5e-6


@pytest.mark.parametrize("name", TRAIN)
def test_train_steps_match_reference(name):
    g = load_golden(name)
    cfg = g["cfg"]
    o = VAEOracle(split_prefix(g, "p0."), cfg["num_layers"], cfg["layer_norm"])
    n = cfg["n_steps"]
    for s in range(n):
        losses, grads = o.train_step(g[f"x{s}"], g[f"eps{s}"], cfg["kl_w"], lr=cfg["lr"], wd=cfg["wd"], clip=cfg["clip"])
        ref = g[f"loss{s}"]
        assert abs(losses["loss"] - ref[0]) < 2e-6 * max(1, abs(ref[0]))
        assert abs(losses["recon"] - ref[1]) < 2e-6 * max(1, abs(ref[1]))
        assert abs(losses["kl"] - ref[2]) < 2e-6
        assert abs(losses["grad_norm"] - ref[3]) < 1e-5 * max(1, ref[3])
        if s == 0:
            for k, v in split_prefix(g, "g0.").items():
                scale = max(1e-3, np.abs(v).max())
                assert np.abs(grads[k] - v).max() < 2e-5 * scale, k
        if s == 0 or s == n - 1:
            for k, v in split_prefix(g, f"p{s + 1}.").items():
                # Adam's m/sqrt(v) amplifies fp32 rounding of tiny grads: bound by a fraction of lr
                assert np.abs(o.p[k] - v).max() < 0.05 * cfg["lr"] * (s + 1), (k, s)


def test_kl_schedules():
    g = load_golden("kl_weight")
    assert np.allclose([kl_weight_4dof(e, 50, 0.30) for e in range(1, 51)], g["dof4"], rtol=0, atol=1e-15)
    assert np.allclose([kl_weight_1dof(e, 100) for e in range(100)], g["dof1"], rtol=0, atol=1e-15)
    assert np.allclose([kl_weight_openlab(e, 100) for e in range(1, 101)], g["openlab"], rtol=0, atol=1e-15)


def test_dropout_masks_gradcheck():
    """Hand-written BPTT with inter-layer dropout masks vs central differences (fp64)."""
    rng = np.random.default_rng(0)
    g = load_golden("vae_fwd_mini")
    p = {k: v.astype(np.float64) for k, v in split_prefix(g, "p.").items()}
    B, T = 3, 5
    x = rng.standard_normal((B, T, 12)); eps = rng.standard_normal((B, 16))
    masks = {"enc": (rng.random((B, T, 32)) > 0.3).astype(np.float64), "dec": (rng.random((B, T, 32)) > 0.3).astype(np.float64)}
    o = VAEOracle(p, 2, True)
    o.forward(x, eps, masks, 0.3)
    _, grads = o.loss_and_grads(0.7)
    for name in ["encoder_lstm.weight_hh_l0", "encoder_lstm.weight_ih_l1", "decoder_lstm.weight_ih_l0",
                 "decoder_lstm.bias_hh_l1", "layer_norm.weight", "fc_logvar.weight", "output_layer.weight"]:
        idx = tuple(rng.integers(0, s) for s in p[name].shape)
        h = 1e-6
        vals = []
        for sgn in (+1, -1):
            q = {k: v.copy() for k, v in p.items()}
            q[name][idx] += sgn * h
            oo = VAEOracle(q, 2, True)
            oo.forward(x, eps, masks, 0.3)
            vals.append(oo.loss_and_grads(0.7)[0]["loss"])
        num = (vals[0] - vals[1]) / (2 * h)
        assert abs(num - grads[name][idx]) < 1e-6 * max(1.0, abs(num)), (name, num, grads[name][idx])
