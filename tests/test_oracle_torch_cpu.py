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
