"""GPU parity tests of the CNN classifier path and the hybrid gate->CNN inference (pytest -m gpu), through the C ABI.
Tolerance: north_star's "CNN logits within 1e-4 fp32"."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, split_prefix
from test_gpu_vae import inject_eps, maxdiff

pytestmark = pytest.mark.gpu
TOL = 1e-4
CONV_BIAS = ("conv1.0.bias", "conv2.0.bias")


def build_cnn(cfg, state, dropout=0.5):
    from Scripts.Models.cnn_model import CNN
    m = CNN(2, 2, dropout, seq_len=cfg["T"], num_features=cfg["D"])
    m.load_state_dict({k: torch.tensor(v) for k, v in state.items()}, strict=True)
    return m.cuda()


@pytest.mark.parametrize("name", ["cnn_eval_ref", "cnn_eval_ns"])
def test_eval_forward_matches_reference(name):
    from shm_hip.cnn import predict
    g = load_golden(name)
    m = build_cnn(g["cfg"], split_prefix(g, "p.")).eval()
    with torch.no_grad():
        logits = m(torch.tensor(g["xin"]).cuda())
    cls, p1 = predict(logits)
    d = dict(logits=maxdiff(logits, g["logits"]), p1=maxdiff(p1, g["probs"][:, 1]))
    print(name, json.dumps(d))
    assert d["logits"] < TOL and d["p1"] < TOL
    assert np.array_equal(cls.cpu().numpy(), g["argmax"])


@pytest.mark.parametrize("name", ["cnn_train_ref", "cnn_train_ns"])
@pytest.mark.parametrize("fused", [False, True])
def test_train_steps_match_reference(name, fused):
    """fused=False: the reference's own loop (05_train_cnn.py:270-276) with torch CE/Adam around the drop-in module;
    fused=True: CNNTrainer (CE, backward, Adam in HIP)."""
    from shm_hip.trainer import CNNTrainer
    g = load_golden(name)
    cfg = g["cfg"]
    m = build_cnn(cfg, split_prefix(g, "p0."), dropout=0.0).train()
    if fused:
        tr = CNNTrainer(m, lr=cfg["lr"], weight_decay=cfg["wd"])
    else:
        opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"], weight_decay=cfg["wd"])
        lossf = torch.nn.CrossEntropyLoss()
    for s in range(cfg["n_steps"]):
        x, y = torch.tensor(g[f"xin{s}"]).cuda(), torch.tensor(g[f"y{s}"]).cuda()
        if fused:
            loss = float(tr.step(x, y)[0])
            grads = dict(zip([k for k, _, _ in tr.engine.layout], tr.engine.split(tr.grads)))
        else:
            opt.zero_grad(set_to_none=True)
            logits = m(x)
            l = lossf(logits, y)
            l.backward()
            grads = {k: p.grad for k, p in m.named_parameters()}
            if s == 0:
                assert maxdiff(logits, g["logits0"]) < TOL
            opt.step()
            loss = float(l)
        assert abs(loss - float(g[f"loss{s}"])) < 1e-5, (loss, float(g[f"loss{s}"]))
        if s == 0:
            rep = {k: float(maxdiff(grads[k], v) / (np.abs(v).max() + 1e-30)) for k, v in split_prefix(g, "g0.").items() if k not in CONV_BIAS}
            print(name, "fused" if fused else "dropin", "grad rel err", json.dumps({k: round(v, 7) for k, v in rep.items()}))
            assert max(rep.values()) < 2e-4, rep
            for k in CONV_BIAS:                      # true gradient is 0 (BatchNorm cancels the bias): only noise on both sides
                assert float(grads[k].abs().max()) < 1e-5
        if s == 0 or s == cfg["n_steps"] - 1:
            ref = split_prefix(g, f"p{s + 1}.")
            sd = m.state_dict()
            for k, v in ref.items():
                if k.endswith("num_batches_tracked"):
                    assert int(sd[k]) == int(v), k
                elif "running" in k:
                    assert maxdiff(sd[k], v) < 1e-5 + 0.25 * cfg["lr"] * s, k
                else:
                    tol = 2.2 * cfg["lr"] * (s + 1) if k in CONV_BIAS else 0.2 * cfg["lr"] * (s + 1)
                    assert maxdiff(sd[k], v) < tol, (k, s)


def test_train_mode_dropout_matches_oracle_with_same_mask():
    from oracle.cnn_oracle import CNNOracle
    from shm_hip._capi import lib
    from shm_hip.trainer import CNNTrainer
    g = load_golden("cnn_train_ref")
    cfg = g["cfg"]
    st = split_prefix(g, "p0.")
    m = build_cnn(cfg, st, dropout=0.5).train()
    tr = CNNTrainer(m, lr=cfg["lr"], weight_decay=cfg["wd"])
    x, y = g["xin0"], g["y0"]
    seed = 0xABCDEF12345
    loss = float(tr.step(torch.tensor(x).cuda(), torch.tensor(y).cuda(), drop_seed=seed)[0])
    B = x.shape[0]
    keep = np.zeros(B * 128, dtype=np.uint8)
    lib.shm_dropout_keep_mask(C.c_uint64(seed), 2, B, 1, 128, C.c_float(0.5), keep.ctypes.data_as(C.c_void_p))
    o = CNNOracle(st)
    logits = o.forward(x, train=True, drop_keep=keep.reshape(B, 128), drop_p=0.5)
    ref_loss, ref_g = o.loss_and_grads(logits, y)
    assert abs(loss - ref_loss) < 1e-5
    grads = dict(zip([k for k, _, _ in tr.engine.layout], tr.engine.split(tr.grads)))
    for k, v in ref_g.items():
        if k not in CONV_BIAS:
            assert maxdiff(grads[k], v) < 2e-4 * np.abs(v).max() + 1e-7, k


@pytest.mark.parametrize("B,T,D", [(5, 10, 5), (37, 36, 12), (3, 7, 4), (130, 64, 8), (2, 21, 6)])
def test_odd_shapes_train_step_matches_oracle(B, T, D):
    """Window shapes off the two golden ones: odd heights (MaxPool floors), widths that are not powers of two, position counts that
    are not multiples of 4 / 16 (the matrix-pipe kernels pad or hand over to the vector kernels), batches above one workgroup per CU."""
    from oracle.cnn_oracle import CNNOracle
    from Scripts.Models.cnn_model import CNN
    from shm_hip.trainer import CNNTrainer
    torch.manual_seed(B * 1000 + T * 10 + D)
    m = CNN(2, 2, 0.0, seq_len=T, num_features=D).cuda().train()
    with torch.no_grad():                                   # non-trivial BN affine so that every gradient path carries signal
        for bn in (m.conv1[1], m.conv2[1]):
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    p0 = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(B + T + D)
    x = rng.standard_normal((B, 2, T, D)).astype(np.float32)
    y = rng.integers(0, 2, B).astype(np.int64)
    tr = CNNTrainer(m, lr=1e-4, weight_decay=5e-5)
    loss = float(tr.step(torch.tensor(x).cuda(), torch.tensor(y).cuda())[0])
    o = CNNOracle(p0)
    logits = o.forward(x, train=True)
    ref_loss, ref_g = o.loss_and_grads(logits, y)
    assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    grads = dict(zip([k for k, _, _ in tr.engine.layout], tr.engine.split(tr.grads)))
    for k, v in ref_g.items():
        if k not in CONV_BIAS:
            assert maxdiff(grads[k], v) < 2e-4 * np.abs(v).max() + 1e-7, (k, B, T, D)


def test_build_input_and_hybrid_pipeline_match_reference():
    """G9: score -> gate -> second VAE pass -> residual channel -> CNN -> labels, both eps draws injected."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip.pipeline import hybrid_predict
    g = load_golden("pipeline_mini")
    vae = TemporalVAE(12, 16, 32, 2, 0.3)
    vae.load_state_dict({k: torch.tensor(v) for k, v in split_prefix(g, "vae.").items()}, strict=True)
    vae = vae.cuda().eval()
    cnn = build_cnn(dict(T=100, D=12), split_prefix(g, "cnn.")).eval()
    Z = torch.tensor(g["Z"]).cuda()
    with inject_eps([torch.tensor(g["eps1"]), torch.tensor(g["eps2"])]):
        mse, y_pred, p_struct = hybrid_predict(vae, cnn, Z, float(g["thr"]))
    assert maxdiff(mse, g["mse"]) < TOL
    assert np.array_equal(y_pred.cpu().numpy(), g["y_pred"])
    assert maxdiff(p_struct, g["p_struct"]) < TOL


def test_hybrid_single_pass_reuses_the_gate_reconstruction():
    """reuse_recon=True: ONE VAE forward per window.  With the gate's epsilon draw injected, the scores equal the
    golden's first pass, and every anomalous window is classified on the reconstruction of that same draw — checked against the
    module-level calls (vae forward with the same draw, residual channel, classifier)."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip.cnn import build_cnn_input
    from shm_hip.pipeline import hybrid_predict
    g = load_golden("pipeline_mini")
    vae = TemporalVAE(12, 16, 32, 2, 0.3)
    vae.load_state_dict({k: torch.tensor(v) for k, v in split_prefix(g, "vae.").items()}, strict=True)
    vae = vae.cuda().eval()
    cnn = build_cnn(dict(T=100, D=12), split_prefix(g, "cnn.")).eval()
    Z = torch.tensor(g["Z"]).cuda()
    thr = float(g["thr"])
    with inject_eps([torch.tensor(g["eps1"])]):
        mse, y_pred, p_struct = hybrid_predict(vae, cnn, Z, thr, batch=Z.size(0), reuse_recon=True)
    assert maxdiff(mse, g["mse"]) < TOL                       # same first draw as the reference's scoring pass
    with torch.no_grad(), inject_eps([torch.tensor(g["eps1"])]):
        xhat, _, _ = vae(Z)
    sel = torch.nonzero(mse > thr).flatten()
    assert 0 < sel.numel() < Z.size(0)
    with torch.no_grad():
        logits = cnn(build_cnn_input(Z[sel], xhat[sel]))
    want = torch.zeros_like(y_pred); want[sel] = logits.argmax(1) + 1
    assert torch.equal(y_pred, want)
    assert maxdiff(p_struct[sel], torch.softmax(logits, 1)[:, 1].cpu().numpy()) < 1e-5
    rest = torch.ones_like(p_struct, dtype=torch.bool); rest[sel] = False
    assert bool((p_struct[rest] == 0).all())                    # windows under the gate never reach the classifier
    # several batches: same labels for the same draws is not expected (one draw per batch), but shapes / gating must hold
    mse2, y2, p2 = hybrid_predict(vae, cnn, Z, thr, batch=7, reuse_recon=True)
    assert mse2.shape == mse.shape and bool(((y2 > 0) == (mse2 > thr)).all())


def test_cnn_refuses_cpu_and_bad_shapes():
    g = load_golden("cnn_eval_ref")
    m = build_cnn(g["cfg"], split_prefix(g, "p."))
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 2, 100, 12))
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 2, 64, 12).cuda())


def test_large_batch_eval_is_deterministic_and_batch_invariant():
    """512-window inference batch (06_test_full_pipeline.py BATCH): windows are independent in eval mode."""
    g = load_golden("cnn_eval_ns")
    m = build_cnn(g["cfg"], split_prefix(g, "p.")).eval()
    x = torch.randn(512, 2, 256, 4, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        a, b, c = m(x), m(x), m(x[:77])
    assert torch.equal(a, b) and torch.equal(a[:77], c) and torch.isfinite(a).all()
