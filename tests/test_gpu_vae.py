"""GPU parity tests of the LSTM-VAE path (run on the MI355X box: pytest -m gpu).

Every test goes through the drop-in module / trainer, i.e. through the C ABI of libshm_hip.so.
Tolerances: north_star's "within 1e-4 fp32" for recon / mu / logvar / score on identical windows with the
reparameterisation noise injected; gradients are compared relative to each tensor's max magnitude.
"""
import json
from contextlib import contextmanager

import numpy as np
import pytest
import torch

from conftest import load_golden, record_parity, split_prefix

pytestmark = pytest.mark.gpu

TOL = 1e-4            # north_star tolerance: held by BOTH parity-grade paths
# "fp32": exact-f32 MFMA (bitwise an fmaf chain); "split": bf16 hi+lo operand pairs, fp32 accumulate (the fast parity path)
PRECS = ["fp32", "split"]


@contextmanager
def inject_eps(eps_list):
    it = iter(eps_list)
    orig = torch.randn_like

    def fake(t, *a, **k):
        e = next(it)
        assert tuple(e.shape) == tuple(t.shape), (e.shape, t.shape)
        return e.to(device=t.device, dtype=t.dtype)

    torch.randn_like = fake
    try:
        yield
    finally:
        torch.randn_like = orig


def build_module(cfg, params, dropout=None, precision="fp32"):
    from Scripts.Models.temporal_vae import TemporalVAE
    m = TemporalVAE(cfg["input_dim"], cfg["latent_dim"], cfg["hidden_dim"], cfg["num_layers"],
                    cfg["dropout"] if dropout is None else dropout, layer_norm=cfg["layer_norm"], precision=precision)
    missing = m.load_state_dict({k: torch.tensor(v) for k, v in params.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.cuda()


def maxdiff(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a.astype(np.float64) - np.asarray(b, np.float64)).max())


FWD = ["vae_fwd_mini", "vae_fwd_1dof", "vae_fwd_openlab", "vae_fwd_ref_h128", "vae_fwd_ns_h128"]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", FWD)
def test_forward_matches_reference_goldens(name, prec):
    g = load_golden(name)
    m = build_module(g["cfg"], split_prefix(g, "p."), precision=prec).eval()
    x = torch.tensor(g["x"]).cuda()
    with torch.no_grad(), inject_eps([torch.tensor(g["eps"])]):
        recon, mu, lv = m(x)
    with inject_eps([torch.tensor(g["eps"])]):
        score = m.score(x)
    d = dict(recon=maxdiff(recon, g["recon"]), mu=maxdiff(mu, g["mu"]), logvar=maxdiff(lv, g["logvar"]),
             score=maxdiff(score, g["score"]))
    print(name, prec, json.dumps(d))
    record_parity("forward_goldens", f"{name}/{prec}", d)
    assert max(d.values()) < TOL, d


def _grad_report(named_grads, ref):
    rep = {}
    for k, v in ref.items():
        scale = max(1e-3, float(np.abs(v).max()))
        rep[k] = maxdiff(named_grads[k], v) / scale
    return rep


TRAIN = ["vae_train_mini", "vae_train_1dof", "vae_train_openlab", "vae_train_ns_h128"]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", TRAIN)
def test_dropin_train_loop_matches_reference(name, prec):
    """The reference's own loop (03_train_vae.py:260-271) with torch's loss ops, clip_grad_norm_ and torch.optim.Adam
    around the drop-in module: autograd reaches the HIP backward through VAEFunction."""
    import torch.nn.functional as F
    g = load_golden(name)
    cfg = g["cfg"]
    m = build_module(cfg, split_prefix(g, "p0."), dropout=0.0, precision=prec).train()
    opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"], weight_decay=cfg["wd"])
    n = cfg["n_steps"]
    for s in range(n):
        x = torch.tensor(g[f"x{s}"]).cuda()
        with inject_eps([torch.tensor(g[f"eps{s}"])]):
            xhat, mu, logvar = m(x)
        recon = F.mse_loss(xhat, x, reduction="mean")
        kl = -0.5 * torch.mean(1.0 + logvar - mu.pow(2) - logvar.exp())
        loss = recon + cfg["kl_w"] * kl
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if s == 0:
            rep = _grad_report({k: p.grad for k, p in m.named_parameters()}, split_prefix(g, "g0."))
            print(name, prec, "grad rel err", json.dumps({k: round(v, 7) for k, v in rep.items()}))
            record_parity("dropin_train", f"{name}/{prec}", {"grad_worst_of_max": max(rep.values()), "recon": maxdiff(xhat, g["recon0"])})
            assert maxdiff(xhat, g["recon0"]) < TOL and maxdiff(mu, g["mu0"]) < TOL and maxdiff(logvar, g["logvar0"]) < TOL
            assert max(rep.values()) < 2e-4, rep
        if cfg["clip"] is not None:
            gn = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=cfg["clip"])
        else:
            gn = torch.sqrt(sum((p.grad ** 2).sum() for p in m.parameters()))
        opt.step()
        ref = g[f"loss{s}"]
        got = [loss.item(), recon.item(), kl.item(), float(gn)]
        print(name, "step", s, "loss/recon/kl/gnorm", got, "ref", ref.tolist())
        assert abs(got[0] - ref[0]) < 1e-4 * max(1, abs(ref[0]))
        assert abs(got[3] - ref[3]) < 1e-3 * max(1, abs(ref[3]))
        if s == 0 or s == n - 1:
            sd = m.state_dict()
            worst = max(maxdiff(sd[k], v) for k, v in split_prefix(g, f"p{s + 1}.").items())
            print(name, "params after step", s + 1, "max abs diff", worst)
            assert worst < 0.2 * cfg["lr"] * (s + 1)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", TRAIN)
def test_fused_trainer_matches_reference(name, prec):
    """Fused path: forward -> shm_vae_loss -> backward -> shm_adam_step (clip + Adam in HIP)."""
    from shm_hip.trainer import VAETrainer
    g = load_golden(name)
    cfg = g["cfg"]
    m = build_module(cfg, split_prefix(g, "p0."), dropout=0.0, precision=prec).train()
    tr = VAETrainer(m, lr=cfg["lr"], weight_decay=cfg["wd"], max_norm=cfg["clip"])
    n = cfg["n_steps"]
    for s in range(n):
        x = torch.tensor(g[f"x{s}"]).cuda()
        stats = tr.step(x, cfg["kl_w"], eps=torch.tensor(g[f"eps{s}"]).cuda()).cpu().numpy()
        ref = g[f"loss{s}"]
        print(name, "step", s, "stats", stats.tolist(), "ref", ref.tolist())
        assert np.all(np.abs(stats[:3] - ref[:3]) < 1e-4 * np.maximum(1, np.abs(ref[:3])))
        assert abs(stats[3] - ref[3]) < 1e-3 * max(1, abs(ref[3]))
        if s == 0:
            rep = _grad_report(dict(zip([k for k, _, _ in tr.engine.layout], tr.engine.split(tr.grads))), split_prefix(g, "g0."))
            assert max(rep.values()) < 2e-4, rep
        if s == 0 or s == n - 1:
            sd = m.state_dict()
            worst = max(maxdiff(sd[k], v) for k, v in split_prefix(g, f"p{s + 1}.").items())
            print(name, prec, "fused params after step", s + 1, "max abs diff", worst)
            record_parity("fused_trainer", f"{name}/{prec}/step{s + 1}", {"param_max_abs": worst})
            assert worst < 0.2 * cfg["lr"] * (s + 1)


@pytest.mark.parametrize("prec", PRECS)
def test_train_mode_dropout_matches_oracle_with_same_masks(prec):
    """Inter-layer dropout (nn.LSTM dropout=0.3, train mode): the kernels' counter RNG is rebuilt on the host
    (shm_dropout_keep_mask) and fed to the oracle, which then has to agree on losses and gradients."""
    import ctypes as C
    from oracle.vae_oracle import VAEOracle
    from shm_hip._capi import lib
    from shm_hip.trainer import VAETrainer
    g = load_golden("vae_train_mini")
    cfg = g["cfg"]
    p0 = split_prefix(g, "p0.")
    m = build_module(cfg, p0, dropout=0.3, precision=prec).train()
    tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
    x, eps = g["x0"], g["eps0"]
    # the split path keeps its sequences in the 4-window tile-blocked row order, and the RNG is indexed by position in that layout
    mask_fn = lib.shm_dropout_keep_mask_tiled if prec == "split" else lib.shm_dropout_keep_mask
    B, T, H = x.shape[0], x.shape[1], cfg["hidden_dim"]
    seed = 0x1234ABCD5678
    stats = tr.step(torch.tensor(x).cuda(), 0.37, eps=torch.tensor(eps).cuda(), drop_seed=seed).cpu().numpy()
    masks = {}
    for stack, key in ((0, "enc"), (1, "dec")):
        buf = np.zeros(B * T * H, dtype=np.uint8)
        mask_fn(C.c_uint64(seed), stack, B, T, H, C.c_float(0.3), buf.ctypes.data_as(C.c_void_p))
        masks[key] = buf.reshape(B, T, H).astype(np.float64)
        assert 0.6 < masks[key].mean() < 0.8
    o = VAEOracle(p0, 2, True)
    o.forward(x, eps, masks, 0.3)
    losses, grads = o.loss_and_grads(0.37)
    print("dropout stats", stats.tolist(), losses)
    assert abs(stats[0] - losses["loss"]) < 1e-4 * max(1, losses["loss"])
    rep = _grad_report(dict(zip([k for k, _, _ in tr.engine.layout], tr.engine.split(tr.grads))), grads)
    assert max(rep.values()) < 2e-4, rep


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,T", [(1, 1), (1, 7), (17, 3), (33, 20)])
def test_ragged_batches_and_tiny_windows_match_oracle(B, T, prec):
    """Batch sizes that do not fill a 16-window tile (the reference keeps the partial last batch, 03_train_vae.py:232)."""
    from oracle.vae_oracle import VAEOracle
    g = load_golden("vae_fwd_mini")
    cfg = g["cfg"]
    p = split_prefix(g, "p.")
    rng = np.random.default_rng(B * 100 + T)
    x = rng.standard_normal((B, T, cfg["input_dim"])).astype(np.float32)
    eps = rng.standard_normal((B, cfg["latent_dim"])).astype(np.float32)
    o = VAEOracle(p, 2, True)
    recon, mu, lv, score = o.forward(x, eps)
    m = build_module(cfg, p, precision=prec).eval()
    with torch.no_grad(), inject_eps([torch.tensor(eps)]):
        r2, m2, l2 = m(torch.tensor(x).cuda())
    assert maxdiff(r2, recon) < TOL and maxdiff(m2, mu) < TOL and maxdiff(l2, lv) < TOL


@pytest.mark.parametrize("precision,layer_norm,H", [("fp32", False, 32), ("fp32", True, 64), ("split", False, 32), ("split", True, 64), ("split", True, 128),
                                                     ("bf16", False, 32), ("bf16", True, 128)])
def test_baseline_config0_shape_train_step(precision, layer_norm, H):
    """BASELINE.json configs[0]: the 1_DOF-style LSTM-VAE on ONE channel, seq_len 128, latent 8 — one fused train step (forward, loss,
    BPTT, clip, Adam) against the fp64 oracle.  fp32 path: loss 1e-4, gradients 2e-4 of each tensor's max; bf16 path: DESIGN.md §2 bars."""
    from oracle.vae_oracle import VAEOracle
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip.trainer import VAETrainer
    torch.manual_seed(128)
    m = TemporalVAE(1, 8, H, 2, 0.0, layer_norm=layer_norm, precision=precision).cuda().train()
    p0 = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, 128, 1, generator=g)
    eps = torch.randn(37, 8, generator=g)
    tr = VAETrainer(m, lr=1e-3, weight_decay=0.0, max_norm=None if not layer_norm else 2.0)
    stats = tr.step(x.cuda(), 0.25, eps=eps.cuda()).cpu().numpy()
    tr.engine.check_status()
    o = VAEOracle(p0, 2, layer_norm)
    o.forward(x.numpy(), eps.numpy())
    losses, grads = o.loss_and_grads(0.25)
    got = dict(zip([k for k, _, _ in tr.engine.layout], tr.engine.split(tr.grads)))
    if precision != "bf16":
        assert abs(stats[0] - losses["loss"]) < 1e-4 * max(1.0, losses["loss"])
        rep = _grad_report(got, grads)
        assert max(rep.values()) < 2e-4, rep
    else:
        assert abs(stats[0] - losses["loss"]) < 5e-4 * max(1.0, losses["loss"])
        rel = {k: float(np.linalg.norm(got[k].cpu().numpy() - v) / max(np.linalg.norm(v), 1e-12)) for k, v in grads.items()}
        assert max(rel.values()) < 2e-2, rel


def test_cpu_tensor_is_refused():
    g = load_golden("vae_fwd_mini")
    m = build_module(g["cfg"], split_prefix(g, "p."))
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 5, 12))


@pytest.mark.parametrize("prec", PRECS)
def test_full_size_properties(prec):
    """North-star size (B=512, T=256, D=4): determinism, batch-split invariance, score definition."""
    from Scripts.Models.temporal_vae import TemporalVAE
    torch.manual_seed(42)
    m = TemporalVAE(4, 16, 128, 2, 0.3, precision=prec).cuda().eval()
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(512, 256, 4, generator=gen).cuda()
    eps = torch.randn(512, 16, generator=gen)
    with torch.no_grad():
        with inject_eps([eps]):
            r1, m1, l1 = m(x)
        with inject_eps([eps]):
            r2, m2, l2 = m(x)
        with inject_eps([eps[:100]]):
            r3, m3, l3 = m(x[:100])
        with inject_eps([eps]):
            sc = m.score(x)
    assert torch.equal(r1, r2) and torch.equal(m1, m2) and torch.equal(l1, l2)          # bitwise reproducible
    assert torch.equal(r1[:100], r3) and torch.equal(m1[:100], m3)                        # windows are independent
    assert torch.isfinite(r1).all()
    ref_score = ((x - r1) ** 2).mean(dim=(1, 2))
    assert maxdiff(sc, ref_score.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("nc", [2, 4])
def test_split_time_chunked_layer_pipeline_equals_one_launch_per_layer(nc, monkeypatch):
    """SHM_SP_CHUNKS (api.hip: sp_chunks): the split path's layers run as a launch-level pipeline over time chunks, (c) / (dc, dh) carried
    through the workspace between chunk launches.  Same arithmetic in the same order per step: the forward is bit-identical; in the
    backward only sum_t dpre is accumulated chunk by chunk (fp32 re-association), so gradients agree to ~1e-6 of their max.  Dropout
    on: the masks are hashed from (seed, stack, row), not from the launch, and must not move."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip.trainer import VAETrainer
    B, T = 37, 64
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, 4, generator=gen).cuda()
    eps = torch.randn(B, 16, generator=gen).cuda()
    out = {}
    for chunks in (1, nc):
        monkeypatch.setenv("SHM_SP_CHUNKS", str(chunks))
        torch.manual_seed(3)
        m = TemporalVAE(4, 16, 128, 2, 0.3, precision="split").cuda().train()
        m.eval()
        with torch.no_grad(), inject_eps([eps.cpu()]):          # inference forward on the initial weights (the step below moves them)
            rec, mu, lv = m(x)
        m.train()
        tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
        stats = tr.step(x, 0.5, eps=eps, drop_seed=77).cpu().numpy()
        tr.check(stats)
        out[chunks] = (stats, tr.grads.clone(), rec, mu, lv)
    a, b = out[1], out[nc]
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert abs(a[0][0] - b[0][0]) <= 1e-6 * max(1.0, abs(a[0][0]))
    gmax = float(a[1].abs().max())
    assert float((a[1] - b[1]).abs().max()) < 2e-6 * gmax
