"""Pins oracle/torch_cpu.py (the timed CPU baseline) against the reference goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden, split_prefix
from oracle.torch_cpu import TorchCpuVAE


@pytest.mark.parametrize("name", ["vae_fwd_mini", "vae_fwd_ns_h128", "vae_fwd_1dof", "vae_fwd_openlab"])
def test_forward(name):
    g = load_golden(name)
    m = TorchCpuVAE(split_prefix(g, "p."), g["cfg"]["num_layers"], g["cfg"]["layer_norm"])
    m.training = False
    with torch.no_grad():
        recon, mu, lv = m.forward(torch.tensor(g["x"]), torch.tensor(g["eps"]))
    assert np.abs(recon.numpy() - g["recon"]).max() < 2e-6
    assert np.abs(mu.numpy() - g["mu"]).max() < 2e-6 and np.abs(lv.numpy() - g["logvar"]).max() < 2e-6


@pytest.mark.parametrize("name", ["vae_train_mini", "vae_train_openlab"])
def test_train_steps(name):
    g = load_golden(name)
    cfg = g["cfg"]
    m = TorchCpuVAE(split_prefix(g, "p0."), cfg["num_layers"], cfg["layer_norm"])
    opt = m.make_optimizer(cfg["lr"], cfg["wd"])
    for s in range(cfg["n_steps"]):
        out = m.train_step(opt, torch.tensor(g[f"x{s}"]), torch.tensor(g[f"eps{s}"]), cfg["kl_w"], cfg["clip"])
        assert np.allclose(out, g[f"loss{s}"], rtol=2e-6, atol=2e-7)
    for k, v in split_prefix(g, f"p{cfg['n_steps']}.").items():
        assert np.abs(m.p[k].detach().numpy() - v).max() < 1e-6, k


@pytest.mark.parametrize("name", ["cnn_train_ref", "cnn_train_ns"])
def test_cnn_train_step(name):
    """oracle/torch_cpu_cnn.py (the CPU baseline of bench.py --workload cnn) against the goldens generated from the reference CNN class:
    logits, loss, every gradient, the parameters and BatchNorm running statistics after one Adam step (dropout 0 in the goldens)."""
    from oracle.torch_cpu_cnn import LEARNABLE, TorchCpuCNN
    g = load_golden(name)
    cfg = g["cfg"]
    m = TorchCpuCNN(split_prefix(g, "p0."), dropout=0.0)
    opt = m.make_optimizer(cfg["lr"], cfg["wd"])
    loss, logits = m.train_step(opt, torch.tensor(g["xin0"]), torch.tensor(g["y0"]).long())
    assert abs(loss - float(g["loss0"])) < 2e-6 and np.abs(logits.numpy() - g["logits0"]).max() < 2e-5
    p1 = split_prefix(g, "p1.")
    for k in LEARNABLE:
        assert np.abs(m.p[k].detach().numpy() - p1[k]).max() < 1e-6, k
    for k, v in m.buf.items():
        assert np.abs(v.numpy() - p1[k]).max() < 1e-6, k
