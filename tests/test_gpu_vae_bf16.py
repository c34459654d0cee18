"""GPU tests of the bf16 throughput path (bf16 MFMA operands, fp32 accumulate / cell state / master weights).

bf16 cannot meet the fp32 path's 1e-4 (rounding the weights alone to bf16 moves recon by 4.5e-4 on the
reference module), so this path carries its own documented tolerances against the SAME reference goldens:
  forward   recon: abs <= 2e-3 (measured <= 5.7e-4) ; mu, logvar: abs <= 1e-2 (measured <= 3.2e-3) ;
            per-window score: rel <= 5e-4 (measured <= 7.7e-5)
  training  loss: rel <= 5e-4 (measured <= 4e-5) ; every gradient tensor: ||g - g_ref|| / ||g_ref|| <= 2e-2
            (measured <= 5e-3) ; grad-norm rel <= 2e-3 (measured <= 1.5e-4)
The numbers printed by these tests are the measured values; DESIGN.md §5 quotes them.
"""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, record_parity, split_prefix
from test_gpu_vae import inject_eps, maxdiff

pytestmark = pytest.mark.gpu


def build_module(cfg, params, dropout=None):
    from Scripts.Models.temporal_vae import TemporalVAE
    m = TemporalVAE(cfg["input_dim"], cfg["latent_dim"], cfg["hidden_dim"], cfg["num_layers"],
                    cfg["dropout"] if dropout is None else dropout, layer_norm=cfg["layer_norm"], precision="bf16")
    m.load_state_dict({k: torch.tensor(v) for k, v in params.items()}, strict=True)
    return m.cuda()


@pytest.mark.parametrize("name", ["vae_fwd_mini", "vae_fwd_1dof", "vae_fwd_openlab", "vae_fwd_ref_h128", "vae_fwd_ns_h128"])
def test_forward_bf16(name):
    g = load_golden(name)
    m = build_module(g["cfg"], split_prefix(g, "p.")).eval()
    x = torch.tensor(g["x"]).cuda()
    with torch.no_grad(), inject_eps([torch.tensor(g["eps"])]):
        recon, mu, lv = m(x)
    with inject_eps([torch.tensor(g["eps"])]):
        score = m.score(x)
    rel_score = float(np.abs(score.cpu().numpy() - g["score"]).max() / np.abs(g["score"]).max())
    d = dict(recon=maxdiff(recon, g["recon"]), mu=maxdiff(mu, g["mu"]), logvar=maxdiff(lv, g["logvar"]), score_rel=rel_score,
             score_abs=maxdiff(score, g["score"]))
    print("bf16", name, json.dumps(d))
    record_parity("forward_bf16", name, d)
    assert d["recon"] < 2e-3 and d["mu"] < 1e-2 and d["logvar"] < 1e-2 and rel_score < 5e-4, d
    # the quantity north_star names — the per-window reconstruction score — is inside 1e-4 ABSOLUTE on this path too (it averages
    # T*D squared errors, so the elementwise bf16 noise mostly cancels); elementwise recon / mu / logvar are not, hence the bars above
    assert d["score_abs"] < 1e-4, d


@pytest.mark.parametrize("name", ["vae_train_mini", "vae_train_1dof", "vae_train_openlab", "vae_train_ns_h128"])
def test_fused_trainer_bf16(name):
    from shm_hip.trainer import VAETrainer
    g = load_golden(name)
    cfg = g["cfg"]
    m = build_module(cfg, split_prefix(g, "p0."), dropout=0.0).train()
    tr = VAETrainer(m, lr=cfg["lr"], weight_decay=cfg["wd"], max_norm=cfg["clip"])
    x = torch.tensor(g["x0"]).cuda()
    stats = tr.step(x, cfg["kl_w"], eps=torch.tensor(g["eps0"]).cuda()).cpu().numpy()
    ref = g["loss0"]
    grads = dict(zip([k for k, _, _ in tr.engine.layout], tr.engine.split(tr.grads)))
    rep = {}
    for k, v in split_prefix(g, "g0.").items():
        gk = grads[k].cpu().numpy().astype(np.float64)
        rep[k] = float(np.linalg.norm(gk - v) / max(np.linalg.norm(v), 1e-12))
    print("bf16", name, "stats", stats.tolist(), "ref", ref.tolist())
    print("bf16", name, "grad rel-L2", json.dumps({k: round(v, 5) for k, v in rep.items()}))
    assert abs(stats[0] - ref[0]) < 5e-4 * max(1, abs(ref[0]))
    assert abs(stats[3] - ref[3]) < 2e-3 * max(1, abs(ref[3]))
    assert max(rep.values()) < 2e-2, rep


def test_bf16_training_reduces_loss_like_fp32():
    """Short optimisation run on structured synthetic windows: both precisions must descend to the same loss band."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip.trainer import VAETrainer
    rng = np.random.default_rng(0)
    t = np.arange(64)[None, :, None]
    ph = rng.uniform(0, 6.28, (256, 1, 4)); fr = rng.uniform(0.05, 0.3, (256, 1, 4))
    x = torch.tensor(np.sin(fr * t + ph).astype(np.float32)).cuda()
    out = {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(3)
        m = TemporalVAE(4, 8, 64, 2, 0.0, precision=prec).cuda().train()
        tr = VAETrainer(m, lr=2e-3, weight_decay=0.0, max_norm=2.0, seed=5)
        hist = []
        for i in range(60):
            hist.append(float(tr.step(x, 0.05)[0]))
        out[prec] = hist
    print("loss fp32 first/last", out["fp32"][0], out["fp32"][-1], "bf16 first/last", out["bf16"][0], out["bf16"][-1])
    assert out["fp32"][-1] < 0.97 * out["fp32"][0] and out["bf16"][-1] < 0.97 * out["bf16"][0]
    assert abs(out["bf16"][-1] - out["fp32"][-1]) < 2e-3 * out["fp32"][0]            # measured 1e-6


def test_bf16_full_size_properties():
    from Scripts.Models.temporal_vae import TemporalVAE
    torch.manual_seed(42)
    m = TemporalVAE(4, 16, 128, 2, 0.3, precision="bf16").cuda().eval()
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
    assert torch.equal(r1, r2) and torch.equal(m1, m2)
    assert torch.equal(r1[:100], r3) and torch.equal(m1[:100], m3)
    assert torch.isfinite(r1).all()


def test_bf16_train_mode_dropout_matches_oracle_with_same_masks():
    """bf16 path, inter-layer dropout 0.3: masks rebuilt on the host (tile-blocked RNG indexing) and fed to the fp64 oracle."""
    import ctypes as C
    from oracle.vae_oracle import VAEOracle
    from shm_hip._capi import lib
    from shm_hip.trainer import VAETrainer
    g = load_golden("vae_train_mini")
    cfg = g["cfg"]
    p0 = split_prefix(g, "p0.")
    m = build_module(cfg, p0, dropout=0.3).train()
    tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
    x, eps = g["x0"], g["eps0"]
    B, T, H = x.shape[0], x.shape[1], cfg["hidden_dim"]
    seed = 0x77AB12345
    stats = tr.step(torch.tensor(x).cuda(), 0.37, eps=torch.tensor(eps).cuda(), drop_seed=seed).cpu().numpy()
    masks = {}
    for stack, key in ((0, "enc"), (1, "dec")):
        buf = np.zeros(B * T * H, dtype=np.uint8)
        lib.shm_dropout_keep_mask_tiled(C.c_uint64(seed), stack, B, T, H, C.c_float(0.3), buf.ctypes.data_as(C.c_void_p))
        masks[key] = buf.reshape(B, T, H).astype(np.float64)
        assert 0.6 < masks[key].mean() < 0.8
    o = VAEOracle(p0, 2, True)
    o.forward(x, eps, masks, 0.3)
    losses, grads = o.loss_and_grads(0.37)
    assert abs(stats[0] - losses["loss"]) < 5e-4 * max(1, losses["loss"])
    got = dict(zip([k for k, _, _ in tr.engine.layout], tr.engine.split(tr.grads)))
    rep = {k: float(np.linalg.norm(got[k].cpu().numpy().astype(np.float64) - v) / max(np.linalg.norm(v), 1e-12)) for k, v in grads.items()}
    print("bf16 dropout grad rel-L2 max", max(rep.values()))
    assert max(rep.values()) < 2e-2, rep


@pytest.mark.parametrize("train_mode", [False, True])
def test_bf16_layer_pipelined_launch_is_bit_identical(train_mode):
    """2-layer stacks run as ONE launch (layer 1 consuming layer 0's h_t through in-launch sc1 hand-offs).  The schedule must
    not change a bit: compare against one launch per layer, repeatedly (a stale or early read would show up as a mismatch),
    at the north-star shape, with a ragged last tile (B = 510) and with T not a multiple of 3."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip._capi import lib
    from shm_hip.trainer import VAETrainer
    # incl. degenerate lengths: T < the hand-off delay (4), T < one 4-step dx chunk, a single window
    # T % 4 == 0 takes the second form of the backward pair launch (dx formed by the top layer, chunk by chunk), anything else the first;
    # the forward groups its input projection four steps per tile at every length.  drop: both dropout template variants.
    for (B, T, H, drop) in ((512, 256, 128, 0.3), (510, 100, 128, 0.3), (37, 50, 64, 0.3), (9, 31, 32, 0.3), (1, 1, 128, 0.3), (3, 2, 128, 0.3),
                            (5, 3, 64, 0.3), (2, 5, 32, 0.3), (4, 7, 128, 0.3), (37, 48, 64, 0.3), (9, 32, 32, 0.3), (6, 4, 128, 0.3), (5, 8, 64, 0.0),
                            (3, 4, 32, 0.3), (13, 12, 128, 0.0), (64, 40, 128, 0.0),
                            (1100, 20, 128, 0.3)):      # 275 tiles: three ticket groups, the last one partial
        torch.manual_seed(11)
        m = TemporalVAE(4, 16, H, 2, drop, precision="bf16").cuda()
        gen = torch.Generator().manual_seed(B)
        x = torch.randn(B, T, 4, generator=gen).cuda()
        eps = torch.randn(B, 16, generator=gen).cuda()
        ref = None
        for on in (0, 1, 1, 1, 1):
            lib.shm_set_pair_launch(on)
            try:
                if train_mode:
                    torch.manual_seed(11)
                    m2 = TemporalVAE(4, 16, H, 2, drop, precision="bf16").cuda().train()
                    tr = VAETrainer(m2, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
                    stats = tr.step(x, 0.3, eps=eps, drop_seed=99)
                    out = (stats.clone(), tr.grads.clone())
                    tr.engine.check_status()
                else:
                    m.eval()
                    with torch.no_grad(), inject_eps([eps.cpu()]):
                        out = m(x)
                    m._engine.check_status()
            finally:
                lib.shm_set_pair_launch(1)
            if ref is None:
                ref = out
            else:
                for a, b in zip(ref, out):
                    assert torch.equal(a, b), (B, T, H, on, float((a - b).abs().max()))


@pytest.mark.parametrize("shape", [(1, 1, 128), (3, 2, 128), (5, 3, 64), (2, 5, 32), (4, 7, 128), (6, 13, 64), (7, 40, 128, 12), (5, 33, 128, 1),
                                   (6, 30, 64, 12), (9, 1001, 128), (3, 1501, 32)])     # long sequences (reference runs have 1001 samples)
def test_bf16_matches_fp32_path_at_degenerate_shapes(shape):
    """The fp32 path is pinned to the reference at the golden shapes; here it is the checker for sequence lengths around the
    kernels' pipeline depths (hand-off delay 4, 4-step dx chunks, prefetch depth 3/4) and for single windows."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip.trainer import VAETrainer
    B, T, H = shape[:3]
    D = shape[3] if len(shape) > 3 else 4              # D = 12 / 1: the fused 16-column dW_ih product of the H = 128 GEMM with Nx != 4
    gen = torch.Generator().manual_seed(1000 + B * 31 + T)
    x = torch.randn(B, T, D, generator=gen).cuda()
    eps = torch.randn(B, 8, generator=gen).cuda()
    out = {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(5)
        m = TemporalVAE(D, 8, H, 2, 0.0, precision=prec).cuda().train()
        tr = VAETrainer(m, lr=1e-3, weight_decay=0.0, max_norm=2.0)
        stats = tr.step(x, 0.25, eps=eps).cpu().numpy()
        tr.engine.check_status()
        with torch.no_grad(), inject_eps([eps.cpu()]):
            recon, mu, lv = m.eval()(x)
        out[prec] = (stats, tr.grads.cpu().numpy().astype(np.float64), recon.cpu().numpy())
    s32, g32, r32 = out["fp32"]
    s16, g16, r16 = out["bf16"]
    assert abs(s16[0] - s32[0]) < 2e-3 * max(1.0, abs(s32[0])), (s16, s32)
    assert np.linalg.norm(g16 - g32) < 3e-2 * max(np.linalg.norm(g32), 1e-9), np.linalg.norm(g16 - g32) / np.linalg.norm(g32)
    assert np.abs(r16 - r32).max() < 5e-3


def test_bf16_pipelined_handoff_under_memory_load():
    """The in-launch hand-offs must not depend on timing: repeat the pipelined train step while a second stream streams 512 MB copies
    through HBM/L2 (uneven load on the CUs' memory queues), and require bit-identity with the one-launch-per-layer result every time."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip._capi import lib
    from shm_hip.trainer import VAETrainer
    B, T, H = 512, 128, 128
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(B, T, 4, generator=gen).cuda()
    eps = torch.randn(B, 16, generator=gen).cuda()

    def one(on):
        lib.shm_set_pair_launch(on)
        try:
            torch.manual_seed(21)
            m = TemporalVAE(4, 16, H, 2, 0.3, precision="bf16").cuda().train()
            tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
            st = tr.step(x, 0.3, eps=eps, drop_seed=5).clone()
            tr.engine.check_status()
            return st, tr.grads.clone()
        finally:
            lib.shm_set_pair_launch(1)

    ref = one(0)
    side = torch.cuda.Stream()
    a = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda")      # 512 MB
    b = torch.empty_like(a)
    for rep in range(12):
        with torch.cuda.stream(side):
            for _ in range(3 + rep % 4):                                        # varying amounts of background traffic
                b.copy_(a, non_blocking=True)
        got = one(1)
        side.synchronize()
        for u, v in zip(ref, got):
            assert torch.equal(u, v), (rep, float((u - v).abs().max()))


def test_handoff_status_word_is_sticky_and_fails_loudly():
    """A consumer workgroup that gives up waiting sets a status word in the plan's workspace and finishes with garbage.  That word
    must (i) survive the weight re-packing of every optimiser step, (ii) arrive in the trainer's statistics buffer without extra
    synchronisation and (iii) make the run fail loudly a few steps later — here the word is forced by writing it into the workspace."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip._capi import lib
    from shm_hip.pipeline import score_windows
    from shm_hip.trainer import VAETrainer
    torch.manual_seed(3)
    m = TemporalVAE(4, 16, 128, 2, 0.0, precision="bf16").cuda().train()
    tr = VAETrainer(m, lr=1e-3, weight_decay=0.0, max_norm=2.0)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(8, 16, 4, generator=g).cuda()
    for _ in range(3):
        s = tr.step(x, 0.5).cpu().numpy()
        tr.check(s)                                              # healthy: no exception, word == 0
    assert int(s.view(np.uint32)[4]) == 0
    pl = tr.engine.last_plan
    off = int(lib.shm_vae_plan_status_offset(pl.handle))
    assert off >= 0
    pl.ws[off:off + 4].view(torch.int32).fill_(0x80000005 - (1 << 32))      # "the consumer of tile 4 (backward) gave up"
    raised_in_step = False
    try:
        for _ in range(4):                                       # several packs later the word must still be there
            s = tr.step(x, 0.5).cpu().numpy()
    except RuntimeError as e:
        raised_in_step = "hand-off timeout" in str(e)
    assert raised_in_step, "VAETrainer.step did not raise on a set status word"
    assert int(tr.stats.cpu().numpy().view(np.uint32)[4]) == 0x80000005
    with pytest.raises(RuntimeError, match="hand-off timeout"):
        tr.check(tr.stats.cpu().numpy())
    with pytest.raises(RuntimeError, match="gave up waiting"):
        tr.engine.check_status()
    tr.engine.reset_status()
    tr.engine.check_status()
    # inference plans: the scoring loop of the stage scripts checks once per call
    m.eval()
    sc = score_windows(m, x, batch=8)
    assert torch.isfinite(sc).all()
    pli = tr.engine.last_plan
    offi = int(lib.shm_vae_plan_status_offset(pli.handle))
    pli.ws[offi:offi + 4].view(torch.int32).fill_(7)
    with pytest.raises(RuntimeError, match="gave up waiting"):
        score_windows(m, x, batch=8)
