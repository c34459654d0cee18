"""Pins oracle/cnn_oracle.py against goldens generated from the reference CNN class (CPU-only)."""
import numpy as np
import pytest

from conftest import load_golden, split_prefix
from oracle.cnn_oracle import BUFFERS, PARAMS, CNNOracle, build_input


@pytest.mark.parametrize("name", ["cnn_eval_ref", "cnn_eval_ns"])
def test_eval_forward(name):
    g = load_golden(name)
    o = CNNOracle(split_prefix(g, "p."))
    logits = o.forward(g["xin"], train=False)
    assert np.abs(logits - g["logits"]).max() < 5e-6
    assert np.abs(o.softmax(logits) - g["probs"]).max() < 5e-6
    assert np.array_equal(logits.argmax(1), g["argmax"])


@pytest.mark.parametrize("name", ["cnn_train_ref", "cnn_train_ns"])
def test_train_steps(name):
    g = load_golden(name)
    cfg = g["cfg"]
    o = CNNOracle(split_prefix(g, "p0."))
    for s in range(cfg["n_steps"]):
        logits = o.forward(g[f"xin{s}"], train=True)
        loss, grads = o.loss_and_grads(logits, g[f"y{s}"])
        assert abs(loss - float(g[f"loss{s}"])) < 2e-6
        if s == 0:
            assert np.abs(logits - g["logits0"]).max() < 1e-5
            for k, v in split_prefix(g, "g0.").items():
                # conv biases sit in front of BatchNorm: their true gradient is 0 and the reference holds fp32 noise (~5e-7)
                assert np.abs(grads[k] - v).max() < 2e-5 * np.abs(v).max() + 2e-6, k
        o.adam_step(grads, lr=cfg["lr"], wd=cfg["wd"])
        if s == 0 or s == cfg["n_steps"] - 1:
            ref = split_prefix(g, f"p{s + 1}.")
            st = o.state()
            for k in PARAMS:
                # conv biases: true gradient 0, so Adam's m/sqrt(v) turns the reference's fp32 rounding noise into +-lr
                # steps whose sign nobody can reproduce; they are only bounded (2*lr per step), everything else is pinned
                tol = 2.2 * cfg["lr"] * (s + 1) if k in ("conv1.0.bias", "conv2.0.bias") else 0.2 * cfg["lr"] * (s + 1)
                assert np.abs(st[k] - ref[k]).max() < tol, (k, s)
            for k in BUFFERS:
                if k.endswith("num_batches_tracked"):
                    assert int(st[k]) == int(ref[k])
                else:
                    # running_mean inherits the +-lr conv-bias ambiguity (x momentum 0.1 per step)
                    assert np.abs(st[k] - ref[k]).max() < 1e-5 + 0.25 * cfg["lr"] * s, k


def test_build_input_matches_pipeline_golden():
    g = load_golden("pipeline_mini")
    z = g["Z"][g["sel"]]
    # the golden stores logits of CNN(stack([z, (z - zhat)^2])); here only the stacking convention is checked
    xin = build_input(z, z * 0.5)
    assert xin.shape == (z.shape[0], 2, z.shape[1], z.shape[2])
    assert np.array_equal(xin[:, 0], z) and np.allclose(xin[:, 1], (0.5 * z) ** 2)
