"""Split entry points of the module API (reference 4DOF/Scripts/Models/temporal_vae.py:51-70: encode / reparameterize / decode are
public methods) and the on-device reparameterisation noise of the fused train step."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, split_prefix
from test_gpu_vae import build_module, inject_eps, maxdiff

pytestmark = pytest.mark.gpu


def _kernels_of(fn):
    """Names of the timed launch groups a call issues (shm_profile_begin / _end)."""
    from shm_hip._capi import check, lib
    check(lib.shm_profile_begin(), "shm_profile_begin")
    out = fn()
    buf = C.create_string_buffer(1 << 16)
    check(lib.shm_profile_end(buf, len(buf)), "shm_profile_end")
    return out, set(json.loads(buf.value.decode()))


@pytest.mark.parametrize("prec", ["fp32", "split", "bf16"])
def test_encode_and_decode_are_the_two_halves_of_forward(prec):
    g = load_golden("vae_fwd_ns_h128")
    m = build_module(g["cfg"], split_prefix(g, "p."), precision=prec).eval()
    x, eps = torch.tensor(g["x"]).cuda(), torch.tensor(g["eps"])
    with torch.no_grad(), inject_eps([eps]):
        recon, mu, lv = m(x)
    (mu2, lv2), names = _kernels_of(lambda: m.encode(x))
    assert torch.equal(mu, mu2) and torch.equal(lv, lv2)                       # the same kernels on the same inputs: bitwise
    assert not any(".dec" in n or n == "out_fwd" for n in names), names         # encode() does not pay for the decoder
    assert any(".enc" in n for n in names)
    z = mu + eps.cuda() * torch.exp(0.5 * lv)                                   # temporal_vae.py:60-63 with the injected eps
    recon2, names = _kernels_of(lambda: m.decode(z, x.size(1)))
    assert not any(".enc" in n for n in names) and any(".dec" in n for n in names), names
    # z is formed here by torch's exp instead of the head kernel's expf: equal to rounding, so the reconstructions agree to ~1e-6
    assert recon2.shape == recon.shape and maxdiff(recon2, recon.cpu().numpy()) < (2e-5 if prec != "bf16" else 2e-3)
    if prec != "bf16":
        assert maxdiff(recon2, g["recon"]) < 1e-4 and maxdiff(mu2, g["mu"]) < 1e-4 and maxdiff(lv2, g["logvar"]) < 1e-4
    # another window length through decode(z, seq_len), as the reference signature allows
    r3 = m.decode(z, 40)
    assert r3.shape == (x.size(0), 40, x.size(2)) and maxdiff(r3, recon2[:, :40].cpu().numpy()) < (2e-5 if prec != "bf16" else 2e-3)
    with pytest.raises(RuntimeError):
        m.decode(torch.zeros(2, 3).cuda(), 10)


@pytest.mark.parametrize("prec", ["fp32", "split"])
def test_device_drawn_eps_equals_the_injected_replica(prec):
    """VAETrainer.step without eps draws the noise inside the head kernel (counter RNG keyed by seed / step).  shm_eps_normal replays
    that stream on the host; a second trainer fed the replica as injected eps has to produce the same losses and gradients."""
    from shm_hip._capi import lib
    from shm_hip.trainer import VAETrainer
    g = load_golden("vae_train_mini")
    cfg = g["cfg"]
    x = torch.tensor(g["x0"]).cuda()
    B, Zd = x.size(0), cfg["latent_dim"]
    out = []
    for inject in (False, True):
        m = build_module(cfg, split_prefix(g, "p0."), dropout=0.0, precision=prec).train()
        tr = VAETrainer(m, lr=cfg["lr"], weight_decay=cfg["wd"], max_norm=cfg["clip"], seed=77)
        eps = None
        if inject:
            e = np.zeros(B * Zd, np.float32)
            lib.shm_eps_normal(C.c_uint64(77), C.c_uint64(1), B * Zd, e.ctypes.data_as(C.c_void_p))     # stream (seed, step 1)
            eps = torch.tensor(e.reshape(B, Zd)).cuda()
            assert abs(float(e.mean())) < 0.3 and 0.7 < float(e.std()) < 1.3
        (st, names) = _kernels_of(lambda: tr.step(x, cfg["kl_w"], eps=eps).cpu().numpy())
        out.append((st, tr.grads.cpu().numpy().copy()))
    (s0, g0), (s1, g1) = out
    # libm's logf / cosf on the host vs the device's: the replica matches to a few ulp, the step to ~1e-6
    assert np.allclose(s0[:4], s1[:4], rtol=2e-5, atol=1e-6), (s0, s1)
    assert float(np.abs(g0 - g1).max() / np.abs(g1).max()) < 2e-5
    assert s0[2] > 1e-3                                          # KL > 0: noise was actually applied (z != mu)


def test_no_aten_kernel_on_the_fused_step():
    """The timed step launches only shm:: kernels: the noise comes from the head kernel, not from torch.randn."""
    from torch.profiler import ProfilerActivity, profile
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip.trainer import VAETrainer
    torch.manual_seed(0)
    m = TemporalVAE(4, 16, 128, 2, 0.3, precision="bf16").cuda().train()
    tr = VAETrainer(m)
    x = torch.randn(16, 32, 4).cuda()
    tr.step(x, 0.5)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        tr.step(x, 0.5)
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
    # kernel names arrive demangled ("shm::(anonymous namespace)::...") or mangled ("_ZN3shm12_GLOBAL__N_1..."): both carry the namespace
    foreign = [n for n in names if "shm" not in n and "Memcpy" not in n and "Memset" not in n and "copyBuffer" not in n and "fillBuffer" not in n]
    if not names:
        pytest.skip("torch.profiler returned no device activity on this box")
    assert not foreign, foreign


@pytest.mark.parametrize("prec", ["split", "bf16"])
def test_empty_batch_and_long_sequence(prec):
    """Edge cases of the module API: an empty batch returns empty tensors (as the reference's nn modules do), and a sequence four times
    the north-star length (T = 1024: the persistent loops' counted waits, prefetch rings and 32-bit offsets) still matches the fp64
    oracle — recon / mu / logvar at the precision's forward bars."""
    from oracle.vae_oracle import VAEOracle
    g = load_golden("vae_fwd_ns_h128")
    p = split_prefix(g, "p.")
    m = build_module(g["cfg"], p, precision=prec).eval()
    with torch.no_grad():
        r0, m0, l0 = m(torch.zeros(0, 256, 4).cuda())
        assert r0.shape == (0, 256, 4) and m0.shape == (0, 16) and l0.shape == (0, 16)
        assert m.score(torch.zeros(0, 256, 4).cuda()).shape == (0,) and m.encode(torch.zeros(0, 64, 4).cuda())[0].shape == (0, 16)
        assert m.decode(torch.zeros(0, 16).cuda(), 32).shape == (0, 32, 4)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((6, 1024, 4)).astype(np.float32)
    eps = rng.standard_normal((6, 16)).astype(np.float32)
    recon, mu, lv, score = VAEOracle(p, 2, True).forward(x, eps)
    with torch.no_grad(), inject_eps([torch.tensor(eps)]):
        r2, m2, l2 = m(torch.tensor(x).cuda())
    d = dict(recon=maxdiff(r2, recon), mu=maxdiff(m2, mu), logvar=maxdiff(l2, lv))
    print("T=1024", prec, d)
    if prec == "split":
        assert max(d.values()) < 1e-4, d
    else:
        assert d["recon"] < 2e-3 and d["mu"] < 1e-2 and d["logvar"] < 1e-2, d
