"""torch-CPU restatement of the LSTM-VAE train step — TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT CODE.

Only tests/ and bench.py's cpu_baseline leg import this.  It drives the same third-party arithmetic the reference
uses (ATen's LSTM, LayerNorm, Linear, mse_loss, clip_grad_norm_, Adam — torch is unpinned by the reference,
4DOF/pyproject.toml:5-9) from a plain state_dict, so it is the fair "reference CPU path" to time beside the GPU:
  forward   4DOF/Scripts/Models/temporal_vae.py:51-77
  step      4DOF/Scripts/03_train_vae.py:262-271
Pinned by tests/test_oracle_torch_cpu.py against the goldens generated from the reference classes.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import _VF


class TorchCpuVAE:
    def __init__(self, params: dict, num_layers: int = 2, layer_norm: bool = True, dropout: float = 0.0):
        self.p = {k: torch.as_tensor(v, dtype=torch.float32).clone().requires_grad_(True) for k, v in params.items()}
        self.L, self.ln, self.dropout = num_layers, layer_norm, dropout
        self.training = True

    def parameters(self):
        return list(self.p.values())

    def _lstm(self, x, prefix):
        flat = []
        for l in range(self.L):
            flat += [self.p[f"{prefix}.{k}_l{l}"] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        B, H = x.size(0), self.p[f"{prefix}.weight_hh_l0"].size(1)
        h0 = x.new_zeros(self.L, B, H)
        out, hn, _ = _VF.lstm(x, (h0, h0.clone()), flat, True, self.L, self.dropout if self.L > 1 else 0.0,
                              self.training, False, True)
        return out, hn

    def forward(self, x, eps):
        p = self.p
        _, hn = self._lstm(x, "encoder_lstm")
        h = hn[-1]
        if self.ln:
            h = F.layer_norm(h, (h.size(1),), p["layer_norm.weight"], p["layer_norm.bias"], 1e-5)
        mu = F.linear(h, p["fc_mu.weight"], p["fc_mu.bias"])
        lv = F.linear(h, p["fc_logvar.weight"], p["fc_logvar.bias"])
        z = mu + eps * torch.exp(0.5 * lv)
        h0 = torch.tanh(F.linear(z, p["fc_latent_to_hidden.weight"], p["fc_latent_to_hidden.bias"]))
        dec, _ = self._lstm(h0.unsqueeze(1).repeat(1, x.size(1), 1), "decoder_lstm")
        return F.linear(dec, p["output_layer.weight"], p["output_layer.bias"]), mu, lv

    def make_optimizer(self, lr=1e-3, wd=1e-5):
        return torch.optim.Adam(self.parameters(), lr=lr, weight_decay=wd)

    def train_step(self, opt, x, eps, kl_w, clip=2.0):
        xhat, mu, lv = self.forward(x, eps)
        recon = F.mse_loss(xhat, x, reduction="mean")
        kl = -0.5 * torch.mean(1.0 + lv - mu.pow(2) - lv.exp())
        loss = recon + kl_w * kl
        opt.zero_grad(set_to_none=True)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm=clip) if clip is not None else None
        opt.step()
        return float(loss), float(recon), float(kl), (float(gn) if gn is not None else float("nan"))
