"""openLAB classifier on the GPU against the goldens generated from the reference class (and the oracle at other shapes)."""
import numpy as np
import pytest
import torch

from conftest import PKG, load_golden, split_prefix

pytestmark = pytest.mark.gpu


def _module(p, in_ch=1, nc=2):
    import sys
    if str(PKG / "Codes") not in sys.path:
        sys.path.insert(0, str(PKG / "Codes"))            # the reference scripts put Codes/ on sys.path
    from Models.cnn_model import CNN
    m = CNN(in_ch, nc, 0.4)
    m.load_state_dict({k: torch.tensor(v) for k, v in p.items()}, strict=True)       # reference checkpoints load strictly
    return m.cuda().eval()


@pytest.mark.parametrize("name", ["cnn_openlab_eval_t200", "cnn_openlab_eval_t512"])
def test_logits_match_reference(name):
    g = load_golden(name)
    m = _module(split_prefix(g, "p."))
    with torch.no_grad():
        logits = m(torch.tensor(g["x"]).cuda())
    d = float(np.abs(logits.cpu().numpy() - g["logits"]).max())
    print(name, "max |logit diff|", d)
    assert d < 1e-4                                        # north-star bar; measured ~1e-6
    probs = torch.softmax(logits, dim=1).cpu().numpy()
    assert np.abs(probs - g["probs"]).max() < 1e-5


def test_other_shapes_against_the_oracle_and_error_paths():
    from oracle import cnn_openlab_oracle as O
    g = load_golden("cnn_openlab_eval_t200")
    p = split_prefix(g, "p.")
    m = _module(p)
    rng = np.random.default_rng(4)
    for (B, T, W) in ((1, 8, 1), (5, 37, 3), (130, 64, 4), (2, 203, 6)):        # odd lengths: MaxPool floors; a batch > one wave of CUs
        x = rng.standard_normal((B, 1, T, W)).astype(np.float32)
        with torch.no_grad():
            got = m(torch.tensor(x).cuda()).cpu().numpy()
        assert np.abs(got - O.forward(p, x)).max() < 1e-4, (B, T, W)
    with pytest.raises(RuntimeError):
        m.train()(torch.zeros(1, 1, 16, 4).cuda())
    with pytest.raises(RuntimeError):
        m.eval()(torch.zeros(1, 1, 16, 4))
    # a parameter update must be picked up (version-keyed flat copy)
    x = torch.tensor(rng.standard_normal((3, 1, 40, 4)).astype(np.float32)).cuda()
    with torch.no_grad():
        a = m.eval()(x)
        m.classifier[4].bias.add_(1.0)
        b = m(x)
    assert torch.allclose(b, a + 1.0, atol=1e-6)


def test_openlab_hybrid_inference_semantics():
    """10_test_hybrid_pipeline.py:266-302, 391-409 on the device: gate by the VAE score, classify the anomalous windows with the
    openLAB CNN on their own standardisation, threshold softmax[:, 1]."""
    import sys
    sys.path.insert(0, str(PKG))
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip import dataprep
    from shm_hip.pipeline import hybrid_predict_openlab, score_windows
    g = load_golden("cnn_openlab_eval_t200")
    cnn = _module(split_prefix(g, "p."))
    torch.manual_seed(3)
    vae = TemporalVAE(3, 8, 64, 1, 0.3, precision="bf16").cuda().eval()       # the openLAB VAE configuration (L = 1)
    rng = np.random.default_rng(9)
    N, T = 300, 200
    raw = rng.standard_normal((N, T, 4)).astype(np.float32) * np.linspace(0.5, 3.0, N, dtype=np.float32)[:, None, None]
    mu, sd = dataprep.fit_mean_std_openlab(raw)
    Xc = torch.tensor(dataprep.standardize_openlab(raw, mu, sd, 10.0)).cuda()
    Zv = Xc[:, :, :3].contiguous()                                            # VAE on a channel subset (select_channels)
    sc = score_windows(vae, Zv, 128)
    vthr = float(sc.median())
    mse, y, p = hybrid_predict_openlab(vae, cnn, Zv, Xc, vthr, 0.55, batch=128)
    assert mse.shape == sc.shape and torch.isfinite(mse).all()          # (a second pass draws fresh eps: scores differ slightly)
    anom = mse > vthr
    assert 0 < int(anom.sum()) < N and not y[~anom].any() and not p[~anom].any()
    with torch.no_grad():
        pr = torch.softmax(cnn(Xc[anom].unsqueeze(1)), dim=1)[:, 1]
    assert torch.allclose(p[anom], pr, atol=1e-6)
    assert torch.equal(y[anom], 1 + (p[anom] >= 0.55).long())
    assert set(y.unique().tolist()) <= {0, 1, 2}
