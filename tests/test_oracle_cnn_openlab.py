"""The openLAB CNN oracle against goldens generated from the reference class (tests/golden/make_golden_openlab_cnn.py)."""
import numpy as np
import pytest

from conftest import load_golden, split_prefix
from oracle import cnn_openlab_oracle as O


@pytest.mark.parametrize("name", ["cnn_openlab_eval_t200", "cnn_openlab_eval_t512"])
def test_oracle_matches_reference_logits(name):
    g = load_golden(name)
    p = split_prefix(g, "p.")
    assert sorted(p) == sorted(O.PARAMS)
    logits = O.forward(p, g["x"])
    assert np.abs(logits - g["logits"]).max() < 2e-5, np.abs(logits - g["logits"]).max()     # fp64 restatement vs torch fp32
