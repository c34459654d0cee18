"""Parity at the BENCH shape, direct: one fused train step at B = 512, T = 256, D = 4, H = 128, L = 2 (BASELINE.json configs[1]) on every
precision against the torch-CPU restatement of the reference step (oracle/torch_cpu.py: ATen's own LSTM / LayerNorm / Linear / mse_loss /
clip_grad_norm_ — the third-party arithmetic 4DOF/Scripts/03_train_vae.py:260-271 runs), computed on the host inside the test.
128 quad tiles, 256-way split-K reductions, the layer-pipelined launches' hand-offs (bf16 path): none of that is reached by the
B <= 21 goldens.  Plus a ten-step parameter-drift check at H = 128.

Bars: fp32 and split paths — loss / recon / kl within 1e-4 relative, gradient norm 1e-3, every gradient tensor within 2e-4 of its
max (north_star's 1e-4 on outputs; gradients as in test_gpu_vae.py); bf16 path — its documented bars (test_gpu_vae_bf16.py).
Dropout is off (ATen does not expose its masks) and eps is injected.
"""
import numpy as np
import pytest
import torch

from conftest import record_parity

pytestmark = pytest.mark.gpu

B, T, D, H, Z, L = 512, 256, 4, 128, 16, 2
KL_W = 0.5


def _model(prec, seed=42, dims=(D, Z, H, L)):
    from Scripts.Models.temporal_vae import TemporalVAE
    torch.manual_seed(seed)
    return TemporalVAE(dims[0], dims[1], dims[2], dims[3], 0.0, precision=prec)


@pytest.fixture(scope="module")
def cpu_reference():
    """The reference step on the host, once per module (a few seconds on the box's 16 cores)."""
    from oracle.torch_cpu import TorchCpuVAE
    m = _model("fp32")
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, T, D, generator=g)
    eps = torch.randn(B, Z, generator=g)
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 0 else 16))
    ref = TorchCpuVAE(sd, L, True, dropout=0.0)
    xhat, mu, lv = ref.forward(x, eps)
    recon = torch.nn.functional.mse_loss(xhat, x)
    kl = -0.5 * torch.mean(1.0 + lv - mu.pow(2) - lv.exp())
    loss = recon + KL_W * kl
    loss.backward()
    grads = {k: p.grad.detach().numpy().astype(np.float64) for k, p in ref.p.items()}
    gn = float(np.sqrt(sum((v ** 2).sum() for v in grads.values())))
    return dict(sd=sd, x=x, eps=eps, loss=float(loss), recon=float(recon), kl=float(kl), gnorm=gn, grads=grads,
                xhat=xhat.detach().numpy(), mu=mu.detach().numpy(), lv=lv.detach().numpy())


@pytest.mark.parametrize("prec", ["fp32", "split", "bf16"])
def test_bench_shape_train_step_matches_cpu_reference(cpu_reference, prec):
    from shm_hip.trainer import VAETrainer
    r = cpu_reference
    m = _model(prec)
    m.load_state_dict(r["sd"], strict=True)
    m = m.cuda().train()
    tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
    stats = tr.step(r["x"].cuda(), KL_W, eps=r["eps"].cuda()).cpu().numpy()
    tr.check(stats)
    tr.engine.check_status()
    got = dict(zip([k for k, _, _ in tr.engine.layout], tr.engine.split(tr.grads)))
    of_max, rel_l2 = {}, {}
    for k, v in r["grads"].items():
        gk = got[k].cpu().numpy().astype(np.float64)
        of_max[k] = float(np.abs(gk - v).max() / max(1e-3, np.abs(v).max()))
        rel_l2[k] = float(np.linalg.norm(gk - v) / max(np.linalg.norm(v), 1e-12))
    err = dict(loss_rel=float(abs(stats[0] - r["loss"]) / max(1.0, abs(r["loss"]))), recon_rel=float(abs(stats[1] - r["recon"]) / max(1.0, abs(r["recon"]))),
               kl_rel=float(abs(stats[2] - r["kl"]) / max(1.0, abs(r["kl"]))), gnorm_rel=float(abs(stats[3] - r["gnorm"]) / max(1.0, r["gnorm"])),
               grad_worst_of_max=max(of_max.values()), grad_worst_rel_l2=max(rel_l2.values()))
    # outputs of the same weights (inference forward, injected eps)
    from test_gpu_vae import inject_eps, maxdiff
    m.load_state_dict(r["sd"], strict=True)
    m.eval()
    with torch.no_grad(), inject_eps([r["eps"]]):
        xhat, mu, lv = m(r["x"].cuda())
    err.update(recon_abs=maxdiff(xhat, r["xhat"]), mu_abs=maxdiff(mu, r["mu"]), logvar_abs=maxdiff(lv, r["lv"]))
    print("bench shape", prec, {k: float(f"{v:.3g}") for k, v in err.items()})
    record_parity("bench_shape_train_step", prec, err)
    if prec != "bf16":
        assert err["loss_rel"] < 1e-4 and err["recon_rel"] < 1e-4 and err["kl_rel"] < 1e-4 and err["gnorm_rel"] < 1e-3, err
        assert err["grad_worst_of_max"] < 2e-4, of_max
        assert max(err["recon_abs"], err["mu_abs"], err["logvar_abs"]) < 1e-4, err
    else:
        assert err["loss_rel"] < 5e-4 and err["gnorm_rel"] < 2e-3 and err["grad_worst_rel_l2"] < 2e-2, (err, rel_l2)
        assert err["recon_abs"] < 2e-3 and err["mu_abs"] < 1e-2 and err["logvar_abs"] < 1e-2, err


@pytest.mark.parametrize("prec", ["fp32", "split", "bf16"])
def test_ten_step_parameter_drift_h128(prec):
    """Ten consecutive optimiser steps at H = 128 (B = 32, T = 64) against ten reference steps on the host: loss per step and the
    parameters at the end.  Adam normalises each element's update to ~lr whatever the gradient's size, so after n steps two correct
    implementations may differ by a small multiple of lr on the few elements whose gradient is ~eps.  Bars: fp32 / split 0.1 * lr * n on
    the worst element and 2e-6 rms (measured 3e-6 / 2.5e-5 worst, 1e-8 / 1e-7 rms); bf16 0.5 * lr * n and 1e-4 rms (measured 2.5e-3, 5e-5)."""
    from oracle.torch_cpu import TorchCpuVAE
    from shm_hip.trainer import VAETrainer
    n, Bs, Ts = 10, 32, 64
    m = _model(prec, seed=11)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda().train()
    tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
    ref = TorchCpuVAE(sd, L, True, dropout=0.0)
    opt = ref.make_optimizer(1e-3, 1e-5)
    g = torch.Generator().manual_seed(12)
    worst_loss = 0.0
    for s in range(n):
        x = torch.randn(Bs, Ts, D, generator=g)
        eps = torch.randn(Bs, Z, generator=g)
        rl = ref.train_step(opt, x, eps, KL_W)
        st = tr.step(x.cuda(), KL_W, eps=eps.cuda()).cpu().numpy()
        worst_loss = max(worst_loss, float(abs(st[0] - rl[0]) / max(1.0, abs(rl[0]))))
    got = m.state_dict()
    drift = max(float((got[k].cpu() - v.detach()).abs().max()) for k, v in ref.p.items())
    rms = float(np.sqrt(np.mean(np.concatenate([(got[k].cpu() - v.detach()).numpy().ravel() ** 2 for k, v in ref.p.items()]))))
    print("ten-step drift", prec, dict(worst_loss_rel=worst_loss, param_max_abs=drift, param_rms=rms))
    record_parity("ten_step_drift_h128", prec, dict(worst_loss_rel=worst_loss, param_max_abs=drift, param_rms=rms))
    assert worst_loss < (1e-4 if prec != "bf16" else 1e-3)
    assert drift < (0.1 if prec != "bf16" else 0.5) * 1e-3 * n and rms < (2e-6 if prec != "bf16" else 1e-4)
