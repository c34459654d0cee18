"""CPU oracle for the LSTM-VAE hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (hybrid-vae-cnn-for-shm_amd/) never does and fails loudly when the
HIP library is missing.

A plain-numpy restatement (explicit time loop, hand-written BPTT; float64 by default) of
what the reference computes through third-party PyTorch ops (torch is unpinned by the
reference: 4DOF/pyproject.toml:5-9).  Pinned by goldens generated from the reference's own
classes in the build container (tests/golden/make_golden.py -> tests/golden/*.npz,
torch 2.10.0 CPU fp32); tests/test_oracle_vae.py holds the comparison.

Reference lines restated (relative to /root/reference):
  encode        4DOF/Scripts/Models/temporal_vae.py:51-58   (1_DOF variant without LayerNorm:
                1_DOF/Scripts/Models/temporal_vae.py:41-46)
  reparameterize                                   :60-63
  decode                                           :65-70
  forward                                          :72-77
  loss / step   4DOF/Scripts/03_train_vae.py:264-271
  score         4DOF/Scripts/04_vae_thresholding.py:122
nn.LSTM cell (gate order i,f,g,o; two bias vectors), nn.LayerNorm (biased variance, eps 1e-5),
torch.nn.utils.clip_grad_norm_ and torch.optim.Adam (L2-coupled weight decay) follow the
published PyTorch definitions.
"""
from __future__ import annotations

import numpy as np

PARAM_ORDER_NOTE = "parameter names are the reference state_dict keys (SURVEY.md §8 a1)"


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def vae_param_names(num_layers: int, layer_norm: bool):
    """state_dict key order of TemporalVAE (4DOF/Scripts/Models/temporal_vae.py:28-49)."""
    names = []
    for l in range(num_layers):
        names += [f"encoder_lstm.{k}_l{l}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    if layer_norm:
        names += ["layer_norm.weight", "layer_norm.bias"]
    names += ["fc_mu.weight", "fc_mu.bias", "fc_logvar.weight", "fc_logvar.bias",
              "fc_latent_to_hidden.weight", "fc_latent_to_hidden.bias"]
    for l in range(num_layers):
        names += [f"decoder_lstm.{k}_l{l}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    names += ["output_layer.weight", "output_layer.bias"]
    return names


def _lstm_layer_fwd(xs, const_in, W_ih, W_hh, b, T):
    """One LSTM layer. xs: [B,T,In] or, when const_in, [B,In] fed at every step."""
    B = xs.shape[0]
    H = W_hh.shape[1]
    h = np.zeros((B, H), xs.dtype)
    c = np.zeros((B, H), xs.dtype)
    hs = np.zeros((B, T, H), xs.dtype)
    cache = dict(i=np.zeros((B, T, H), xs.dtype), f=np.zeros((B, T, H), xs.dtype),
                 g=np.zeros((B, T, H), xs.dtype), o=np.zeros((B, T, H), xs.dtype),
                 c=np.zeros((B, T, H), xs.dtype))
    pre_const = xs @ W_ih.T + b if const_in else None
    for t in range(T):
        pre = (pre_const if const_in else xs[:, t] @ W_ih.T + b) + h @ W_hh.T
        i = sigmoid(pre[:, 0:H]); f = sigmoid(pre[:, H:2 * H])
        g = np.tanh(pre[:, 2 * H:3 * H]); o = sigmoid(pre[:, 3 * H:4 * H])
        c = f * c + i * g
        h = o * np.tanh(c)
        hs[:, t] = h
        cache["i"][:, t] = i; cache["f"][:, t] = f; cache["g"][:, t] = g; cache["o"][:, t] = o
        cache["c"][:, t] = c
    cache["hs"] = hs
    return hs, cache


def _lstm_layer_bwd(dhs, xs, const_in, W_ih, W_hh, cache, T, need_dx=True):
    """BPTT through one layer. dhs: [B,T,H] gradient arriving on every h_t. Returns dx, dW_ih, dW_hh, db."""
    B, _, H = dhs.shape
    dt = dhs.dtype
    dW_ih = np.zeros_like(W_ih); dW_hh = np.zeros_like(W_hh); db = np.zeros(4 * H, dt)
    dx = np.zeros_like(xs) if need_dx else None
    dh_rec = np.zeros((B, H), dt); dc = np.zeros((B, H), dt)
    for t in range(T - 1, -1, -1):
        i, f, g, o, c = (cache[k][:, t] for k in "ifgoc")
        c_prev = cache["c"][:, t - 1] if t > 0 else np.zeros((B, H), dt)
        h_prev = cache["hs"][:, t - 1] if t > 0 else np.zeros((B, H), dt)
        dh = dhs[:, t] + dh_rec
        tc = np.tanh(c)
        do = dh * tc
        dc = dc + dh * o * (1.0 - tc * tc)
        di = dc * g; dg = dc * i; df = dc * c_prev
        dc = dc * f
        dpre = np.concatenate([di * i * (1 - i), df * f * (1 - f), dg * (1 - g * g), do * o * (1 - o)], axis=1)
        x_t = xs if const_in else xs[:, t]
        dW_ih += dpre.T @ x_t
        dW_hh += dpre.T @ h_prev
        db += dpre.sum(0)
        dh_rec = dpre @ W_hh
        if need_dx:
            if const_in:
                dx += dpre @ W_ih
            else:
                dx[:, t] = dpre @ W_ih
    return dx, dW_ih, dW_hh, db


class VAEOracle:
    """forward / train-step restatement; params: dict name -> ndarray (reference state_dict keys)."""

    def __init__(self, params, num_layers=2, layer_norm=True, dtype=np.float64):
        self.L = int(num_layers)
        self.ln = bool(layer_norm)
        self.dtype = dtype
        self.p = {k: np.asarray(v, dtype=dtype).copy() for k, v in params.items()}
        self.names = vae_param_names(self.L, self.ln)
        assert set(self.names) == set(self.p), (sorted(set(self.names) ^ set(self.p)))
        self.adam_m = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.adam_v = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.adam_t = 0

    # ---------------------------------------------------------------- forward
    def forward(self, x, eps, drop_masks=None, drop_p=0.0):
        """Returns recon, mu, logvar, per-window score; keeps a cache for backward().

        drop_masks: None (eval / dropout 0) or dict {"enc": [B,T,H] 0/1, "dec": [B,T,H] 0/1} keep-masks applied,
        scaled by 1/(1-p), to the layer-0 outputs feeding layer 1 (nn.LSTM inter-layer dropout, train mode)."""
        p, L = self.p, self.L
        x = np.asarray(x, self.dtype); eps = np.asarray(eps, self.dtype)
        B, T, D = x.shape
        cache = {"x": x, "eps": eps, "T": T}
        scale = 1.0 / (1.0 - drop_p) if drop_masks is not None else 1.0
        # encoder
        inp = x
        cache["enc"] = []
        for l in range(L):
            b = p[f"encoder_lstm.bias_ih_l{l}"] + p[f"encoder_lstm.bias_hh_l{l}"]
            hs, c = _lstm_layer_fwd(inp, False, p[f"encoder_lstm.weight_ih_l{l}"], p[f"encoder_lstm.weight_hh_l{l}"], b, T)
            c["in"] = inp
            cache["enc"].append(c)
            inp = hs
            if l + 1 < L and drop_masks is not None:
                inp = hs * np.asarray(drop_masks["enc"], self.dtype) * scale
        h_last = hs[:, -1]
        if self.ln:
            mean = h_last.mean(1, keepdims=True)
            var = ((h_last - mean) ** 2).mean(1, keepdims=True)
            rstd = 1.0 / np.sqrt(var + 1e-5)
            xhat = (h_last - mean) * rstd
            y = xhat * p["layer_norm.weight"] + p["layer_norm.bias"]
            cache.update(ln_xhat=xhat, ln_rstd=rstd)
        else:
            y = h_last
        mu = y @ p["fc_mu.weight"].T + p["fc_mu.bias"]
        lv = y @ p["fc_logvar.weight"].T + p["fc_logvar.bias"]
        std = np.exp(0.5 * lv)
        z = mu + eps * std
        h0 = np.tanh(z @ p["fc_latent_to_hidden.weight"].T + p["fc_latent_to_hidden.bias"])
        cache.update(y=y, mu=mu, lv=lv, std=std, z=z, h0=h0)
        # decoder: layer 0 sees the same h0 at every step (temporal_vae.py:68)
        cache["dec"] = []
        inp = h0
        for l in range(L):
            b = p[f"decoder_lstm.bias_ih_l{l}"] + p[f"decoder_lstm.bias_hh_l{l}"]
            hs, c = _lstm_layer_fwd(inp, l == 0, p[f"decoder_lstm.weight_ih_l{l}"], p[f"decoder_lstm.weight_hh_l{l}"], b, T)
            c["in"] = inp
            cache["dec"].append(c)
            inp = hs
            if l + 1 < L and drop_masks is not None:
                inp = hs * np.asarray(drop_masks["dec"], self.dtype) * scale
        recon = hs @ p["output_layer.weight"].T + p["output_layer.bias"]
        score = ((x - recon) ** 2).mean(axis=(1, 2))
        cache.update(recon=recon, dec_top=hs, drop_masks=drop_masks, scale=scale)
        self.cache = cache
        return recon, mu, lv, score

    # ---------------------------------------------------------------- loss + backward
    def loss_and_grads(self, kl_w):
        c, p, L = self.cache, self.p, self.L
        x, recon, mu, lv = c["x"], c["recon"], c["mu"], c["lv"]
        B, T, D = x.shape
        Z = mu.shape[1]
        rec = ((recon - x) ** 2).mean()
        kl = -0.5 * (1.0 + lv - mu ** 2 - np.exp(lv)).mean()
        loss = rec + kl_w * kl
        g = {}
        dout = 2.0 * (recon - x) / (B * T * D)
        top = c["dec_top"]
        g["output_layer.weight"] = np.einsum("btd,bth->dh", dout, top)
        g["output_layer.bias"] = dout.sum((0, 1))
        dhs = dout @ p["output_layer.weight"]
        masks, scale = c["drop_masks"], c["scale"]
        for l in range(L - 1, -1, -1):
            cc = c["dec"][l]
            dx, dWi, dWh, db = _lstm_layer_bwd(dhs, cc["in"], l == 0, p[f"decoder_lstm.weight_ih_l{l}"],
                                               p[f"decoder_lstm.weight_hh_l{l}"], cc, T)
            g[f"decoder_lstm.weight_ih_l{l}"] = dWi; g[f"decoder_lstm.weight_hh_l{l}"] = dWh
            g[f"decoder_lstm.bias_ih_l{l}"] = db; g[f"decoder_lstm.bias_hh_l{l}"] = db.copy()
            dhs = dx
            if l > 0 and masks is not None:
                dhs = dx * np.asarray(masks["dec"], self.dtype) * scale
        dh0 = dhs                                   # [B,H] (summed over t inside the const-input layer)
        dpre = dh0 * (1.0 - c["h0"] ** 2)
        g["fc_latent_to_hidden.weight"] = dpre.T @ c["z"]
        g["fc_latent_to_hidden.bias"] = dpre.sum(0)
        dz = dpre @ p["fc_latent_to_hidden.weight"]
        dmu = dz + kl_w * mu / (B * Z)
        dlv = dz * c["eps"] * 0.5 * c["std"] + kl_w * (-0.5) * (1.0 - np.exp(lv)) / (B * Z)
        g["fc_mu.weight"] = dmu.T @ c["y"]; g["fc_mu.bias"] = dmu.sum(0)
        g["fc_logvar.weight"] = dlv.T @ c["y"]; g["fc_logvar.bias"] = dlv.sum(0)
        dy = dmu @ p["fc_mu.weight"] + dlv @ p["fc_logvar.weight"]
        if self.ln:
            xhat, rstd = c["ln_xhat"], c["ln_rstd"]
            g["layer_norm.weight"] = (dy * xhat).sum(0); g["layer_norm.bias"] = dy.sum(0)
            dxh = dy * p["layer_norm.weight"]
            dh_last = rstd * (dxh - dxh.mean(1, keepdims=True) - xhat * (dxh * xhat).mean(1, keepdims=True))
        else:
            dh_last = dy
        H = dh_last.shape[1]
        dhs = np.zeros((B, T, H), self.dtype)
        dhs[:, -1] = dh_last
        for l in range(L - 1, -1, -1):
            cc = c["enc"][l]
            dx, dWi, dWh, db = _lstm_layer_bwd(dhs, cc["in"], False, p[f"encoder_lstm.weight_ih_l{l}"],
                                               p[f"encoder_lstm.weight_hh_l{l}"], cc, T, need_dx=(l > 0))
            g[f"encoder_lstm.weight_ih_l{l}"] = dWi; g[f"encoder_lstm.weight_hh_l{l}"] = dWh
            g[f"encoder_lstm.bias_ih_l{l}"] = db; g[f"encoder_lstm.bias_hh_l{l}"] = db.copy()
            if l > 0:
                dhs = dx
                if masks is not None:
                    dhs = dx * np.asarray(masks["enc"], self.dtype) * scale
        return dict(loss=loss, recon=rec, kl=kl), g

    # ---------------------------------------------------------------- optimiser
    @staticmethod
    def clip_grads(grads, max_norm):
        """torch.nn.utils.clip_grad_norm_ (03_train_vae.py:270): coef = max_norm/(norm+1e-6) clamped to <= 1."""
        total = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in grads.values()))
        if max_norm is None:
            return total, grads
        coef = min(1.0, max_norm / (total + 1e-6))
        return total, {k: v * coef for k, v in grads.items()}

    def adam_step(self, grads, lr=1e-3, wd=1e-5, b1=0.9, b2=0.999, eps=1e-8):
        """torch.optim.Adam defaults (03_train_vae.py:222): weight decay is added to the gradient (not decoupled)."""
        self.adam_t += 1
        t = self.adam_t
        bc1 = 1.0 - b1 ** t; bc2 = 1.0 - b2 ** t
        for k in self.names:
            gk = grads[k] + wd * self.p[k]
            self.adam_m[k] = b1 * self.adam_m[k] + (1 - b1) * gk
            self.adam_v[k] = b2 * self.adam_v[k] + (1 - b2) * gk * gk
            denom = np.sqrt(self.adam_v[k]) / np.sqrt(bc2) + eps
            self.p[k] = self.p[k] - (lr / bc1) * self.adam_m[k] / denom

    def train_step(self, x, eps, kl_w, lr=1e-3, wd=1e-5, clip=2.0, drop_masks=None, drop_p=0.0):
        self.forward(x, eps, drop_masks, drop_p)
        losses, g = self.loss_and_grads(kl_w)
        gnorm, gc = self.clip_grads(g, clip)
        self.adam_step(gc, lr=lr, wd=wd)
        losses["grad_norm"] = gnorm
        return losses, g


# ---------------------------------------------------------------- schedules (a7)
def kl_weight_4dof(epoch, n_epochs, ratio=0.30):
    """4DOF/Scripts/03_train_vae.py:120-142 (epoch 1-based)."""
    e0 = epoch - 1
    warm = max(1, int(n_epochs * ratio))
    return float(1.0 / (1.0 + np.exp(-((e0 - warm) / float(max(warm, 1))) * 5.0)))


def kl_weight_1dof(epoch, n_epochs, ratio=0.3):
    """1_DOF/Scripts/03_train_vae.py:26-30 (epoch 0-based)."""
    pivot = int(n_epochs * ratio)
    return float(1.0 / (1.0 + np.exp(-5.0 * (epoch - pivot) / max(pivot, 1))))


def kl_weight_openlab(epoch, n_epochs, ratio=0.30):
    """20250506_openLAB_tests/Codes/04_train_vae.py:129-131."""
    x = (epoch - (n_epochs * ratio)) / max(n_epochs * ratio, 1e-12)
    return float(1.0 / (1.0 + np.exp(-x * 5.0)))
