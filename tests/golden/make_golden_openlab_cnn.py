#!/usr/bin/env python3
"""Golden vectors for the openLAB fault classifier (reference 20250506_openLAB_tests/Codes/Models/cnn_model.py:8-62), eval mode.

Run in the build container (imports the reference's own class from /root/reference; never on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_openlab_cnn.py
Writes tests/golden/cnn_openlab_eval_t200.npz / _t512.npz: the state_dict (kaiming init as the reference + randomised GroupNorm
affine and biases so that every parameter matters), inputs [B,1,T,4] and logits / softmax.  T = 512 exercises the size-agnostic
conv stack (AdaptiveAvgPool) at BASELINE config 5's sequence length.
"""
import importlib.util
import json
import sys
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, str(path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def case(name, T, B, seed):
    ref = _load("ref_cnn_olab", REF / "20250506_openLAB_tests/Codes/Models/cnn_model.py")
    torch.manual_seed(seed)
    m = ref.CNN(input_channels=1, num_classes=2, dropout_rate=0.4).eval()
    g = torch.Generator().manual_seed(seed + 7)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.GroupNorm):
                mod.weight.copy_(1.0 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)):
                mod.bias.copy_(0.05 * torch.randn(mod.bias.shape, generator=g))
    x = torch.randn(B, 1, T, 4, generator=g)
    with torch.no_grad():
        logits = m(x)
        probs = torch.softmax(logits, dim=1)
    d = {"p." + k: v.detach().numpy() for k, v in m.state_dict().items()}
    d.update(x=x.numpy(), logits=logits.numpy(), probs=probs.numpy(), cfg=np.array(json.dumps(dict(T=T, B=B, W=4))))
    np.savez(OUT / f"{name}.npz", **d)
    print(name, "ok", [tuple(v.shape) for v in m.state_dict().values()][:4], logits[:2].tolist())


if __name__ == "__main__":
    torch.set_num_threads(4)
    case("cnn_openlab_eval_t200", 200, 6, 61)
    case("cnn_openlab_eval_t512", 512, 3, 62)
