"""Parity of the split path's forward against the fp64 oracle when the LSTM weights are larger than their initialisation (the fp16 + fp8
split of W_ih clamps its residual for |W| >= 16): max |mu|, |logvar|, |recon| error per weight scale."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
for p in (str(ROOT), str(ROOT / "hybrid-vae-cnn-for-shm_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from oracle.vae_oracle import VAEOracle
from Scripts.Models.temporal_vae import TemporalVAE
sys.path.insert(0, str(ROOT / "tests"))
from test_gpu_vae import inject_eps

B, T, H, D = 8, 64, 128, 4
g = torch.Generator().manual_seed(3)
x = torch.randn(B, T, D, generator=g)
eps = torch.randn(B, 16, generator=g)
import itertools
for scale, prec in itertools.product((1.0, 2.0, 3.0, 4.0, 6.0), ("fp32", "split")):
    torch.manual_seed(7)
    m = TemporalVAE(D, 16, H, 2, 0.0, precision=prec)
    with torch.no_grad():
        for k, v in m.named_parameters():
            if "lstm.weight" in k:
                v.mul_(scale)
    m = m.cuda().eval()
    p0 = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    wmax = max(float(np.abs(v).max()) for k, v in p0.items() if "lstm.weight" in k)
    o = VAEOracle(p0, 2, True)
    recon, mu, lv, score = o.forward(x.numpy(), eps.numpy())
    with torch.no_grad(), inject_eps([eps]):
        r2, m2, l2 = m(x.cuda())
    e = lambda a, b: float(np.abs(a.cpu().numpy().astype(np.float64) - b).max())
    print(f"{prec:5s} scale {scale:6.1f}  max|W| {wmax:8.3f}   recon {e(r2, recon):.2e}  mu {e(m2, mu):.2e}  logvar {e(l2, lv):.2e}")
