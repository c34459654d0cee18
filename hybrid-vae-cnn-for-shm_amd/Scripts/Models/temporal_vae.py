# Scripts/Models/temporal_vae.py — MI355X drop-in for the reference module of the same path
# (reference 4DOF/Scripts/Models/temporal_vae.py:8-82; 1_DOF variant: layer_norm=False).
from __future__ import annotations

import torch
import torch.nn as nn

from shm_hip._capi import PRECISIONS
from shm_hip.vae import VAEEngine, VAEFunction


class TemporalVAE(nn.Module):
    """
    LSTM-based Temporal Variational Autoencoder, computed by hand-written gfx950 kernels.
    Input shape:  [B, T, D]      Output: (recon [B, T, D], mu [B, Z], logvar [B, Z])

    Same constructor signature, defaults, parameter names/shapes and state_dict layout as the reference
    (temporal_vae.py:14-49): nn.LSTM / nn.LayerNorm / nn.Linear are kept as parameter containers (so the
    default initialisation consumes the torch RNG exactly like the reference and reference checkpoints load
    with strict=True) but are never called — forward() hands the flat parameter buffer to libshm_hip.so.
    Extra keyword-only arguments: layer_norm (False gives the 1_DOF model, 1_DOF/Scripts/Models/temporal_vae.py:7-64);
    precision: "fp32" (exact-f32 MFMA: bit-level fp32 arithmetic, slow), "split" (every MFMA operand a bf16 pair
    hi + lo, fp32 accumulate / state: holds the reference's fp32 results to ~1e-6 — the fast 1e-4 parity path) or
    "bf16" (single bf16 operands, fp32 accumulate / cell state / master weights — the throughput path, tolerances in DESIGN.md §2).
    """

    def __init__(
        self,
        input_dim: int = 12,
        latent_dim: int = 16,
        hidden_dim: int = 128,
        num_layers: int = 2,
        dropout: float = 0.3,
        *,
        layer_norm: bool = True,
        precision: str = "fp32",
    ) -> None:
        super().__init__()
        self.input_dim = input_dim
        self.latent_dim = latent_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.dropout = float(dropout) if num_layers > 1 else 0.0

        # Encoder (parameter containers; creation order = reference order, temporal_vae.py:28-37)
        self.encoder_lstm = nn.LSTM(input_size=input_dim, hidden_size=hidden_dim, num_layers=num_layers,
                                    batch_first=True, dropout=dropout if num_layers > 1 else 0.0)
        if layer_norm:
            self.layer_norm = nn.LayerNorm(hidden_dim)
        self.fc_mu = nn.Linear(hidden_dim, latent_dim)
        self.fc_logvar = nn.Linear(hidden_dim, latent_dim)

        # Decoder (temporal_vae.py:40-49)
        self.fc_latent_to_hidden = nn.Linear(latent_dim, hidden_dim)
        self.decoder_lstm = nn.LSTM(input_size=hidden_dim, hidden_size=hidden_dim, num_layers=num_layers,
                                    batch_first=True, dropout=dropout if num_layers > 1 else 0.0)
        self.output_layer = nn.Linear(hidden_dim, input_dim)

        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
        self.precision = precision
        self._engine = VAEEngine(input_dim, latent_dim, hidden_dim, num_layers, layer_norm, self.dropout,
                                 precision=PRECISIONS[precision])
        self._drop_gen_calls = 0

    # ------------------------------------------------------------------ reference API
    def reparameterize(self, mu: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
        """temporal_vae.py:60-63, for callers that use it on its own (forward() fuses it into the head kernel).  Three elementwise
        torch ops on [B, Z]: plumbing-sized, and the only arithmetic in this file."""
        std = torch.exp(0.5 * logvar)
        eps = torch.randn_like(std)
        return mu + eps * std

    def forward(self, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """temporal_vae.py:72-77.  eps is drawn with torch.randn_like on the device exactly where the reference draws
        it (reparameterize, :62) — in eval mode too — and the fused kernels consume it."""
        if x.device.type != "cuda":
            raise RuntimeError("TemporalVAE (MI355X build) has no CPU path: call .to('cuda') on the module and the input")
        eng = self._engine
        named = dict(self.named_parameters())
        eng.bind_parameters(named)
        eps = torch.randn_like(torch.empty(x.size(0), self.latent_dim, dtype=torch.float32, device=x.device))
        train_mode = self.training and self.dropout > 0.0
        drop_seed = 0
        if train_mode:
            # one 64-bit seed per call from torch's CPU generator: reproducible under torch.manual_seed
            drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        params = [named[name] for name, _, _ in eng.layout]
        if x.size(0) == 0:
            recon, mu, logvar, _, _, _ = eng.forward(x, eps, training_plan=False, train_mode=False)
            return recon, mu, logvar
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return VAEFunction.apply(eng, x, eps, train_mode, drop_seed, *params)
        recon, mu, logvar, _, _, _ = eng.forward(x, eps, training_plan=False, train_mode=train_mode, drop_seed=drop_seed)
        return recon, mu, logvar

    # encode / decode of the reference as separate entry points (inference; gradients flow through forward() only)
    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """temporal_vae.py:51-58: (mu, logvar); the decoder does not run (shm_vae_encode)."""
        eng = self._engine
        eng.bind_parameters(dict(self.named_parameters()))
        train_mode = self.training and self.dropout > 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if train_mode else 0
        return eng.encode(x, train_mode=train_mode, drop_seed=seed)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, seq_len: int) -> torch.Tensor:
        """temporal_vae.py:65-70: z [B, Z] -> x_hat [B, seq_len, D]; the encoder does not run (shm_vae_decode)."""
        eng = self._engine
        eng.bind_parameters(dict(self.named_parameters()))
        train_mode = self.training and self.dropout > 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if train_mode else 0
        return eng.decode(z, seq_len, train_mode=train_mode, drop_seed=seed)

    @torch.no_grad()
    def score(self, x: torch.Tensor) -> torch.Tensor:
        """Per-window MSE ((x - recon)**2).mean(dim=(1, 2)) of 04_vae_thresholding.py:119-123, fused on device."""
        eng = self._engine
        eng.bind_parameters(dict(self.named_parameters()))
        eps = torch.randn_like(torch.empty(x.size(0), self.latent_dim, dtype=torch.float32, device=x.device))
        _, _, _, score, _, _ = eng.forward(x, eps, training_plan=False, train_mode=False, want_score=True, want_recon=False)
        return score

    def score_and_reconstruct(self, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """(per-window MSE, reconstruction) of ONE eval forward (one epsilon draw): lets the hybrid gate hand the anomalous windows'
        own reconstruction to the classifier instead of running the VAE a second time (the reference
        draws twice, 06_test_full_pipeline.py:342,363)."""
        eng = self._engine
        eng.bind_parameters(dict(self.named_parameters()))
        eps = torch.randn_like(torch.empty(x.size(0), self.latent_dim, dtype=torch.float32, device=x.device))
        recon, _, _, score, _, _ = eng.forward(x, eps, training_plan=False, train_mode=False, want_score=True, want_recon=True)
        return score, recon

# Backward compatibility (if any old scripts still import VAE)
VAE = TemporalVAE
__all__ = ["TemporalVAE", "VAE"]
