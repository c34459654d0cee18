"""CPU oracle for the CNN fault classifier — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Plain-numpy restatement (fp64 by default) of the reference model and its train step:
  model        4DOF/Scripts/Models/cnn_model.py:14-51   conv3x3(pad 1) -> BatchNorm2d -> ReLU -> MaxPool2d(2), twice;
               flatten (C,H,W order) -> Linear -> ReLU -> Dropout -> Linear
  input build  4DOF/Scripts/05_train_cnn.py:136-138      xin = stack([z, (z - zhat)^2], dim=1)
  train step   4DOF/Scripts/05_train_cnn.py:270-276      CrossEntropyLoss (mean), Adam(lr 1e-4, wd 5e-5), no clipping
  inference    4DOF/Scripts/06_test_full_pipeline.py:366-371   argmax, softmax[:, 1]
BatchNorm2d follows the published PyTorch definition (batch statistics with biased variance in train mode, running
statistics updated with momentum 0.1 and the UNBIASED variance, eps 1e-5).  Pinned by tests/test_oracle_cnn.py to goldens
generated from the reference class (tests/golden/cnn_*.npz).
"""
from __future__ import annotations

import numpy as np

PARAMS = ["conv1.0.weight", "conv1.0.bias", "conv1.1.weight", "conv1.1.bias",
          "conv2.0.weight", "conv2.0.bias", "conv2.1.weight", "conv2.1.bias",
          "fc1.0.weight", "fc1.0.bias", "fc2.weight", "fc2.bias"]
BUFFERS = ["conv1.1.running_mean", "conv1.1.running_var", "conv1.1.num_batches_tracked",
           "conv2.1.running_mean", "conv2.1.running_var", "conv2.1.num_batches_tracked"]


def build_input(z, zhat):
    """05_train_cnn.py:137-138."""
    return np.stack([z, (z - zhat) ** 2], axis=1)


def _im2col(x):
    """x [B,C,H,W] -> [B,H,W,C*9] patches of the zero-padded 3x3 neighbourhood (row order c,kh,kw)."""
    B, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    cols = np.empty((B, H, W, C, 3, 3), x.dtype)
    for kh in range(3):
        for kw in range(3):
            cols[:, :, :, :, kh, kw] = xp[:, :, kh:kh + H, kw:kw + W].transpose(0, 2, 3, 1)
    return cols.reshape(B, H, W, C * 9)


def _conv_fwd(x, w, b):
    cols = _im2col(x)
    y = cols @ w.reshape(w.shape[0], -1).T + b
    return y.transpose(0, 3, 1, 2), cols


def _conv_bwd(dy, cols, w, x_shape, need_dx):
    B, C, H, W = x_shape
    d = dy.transpose(0, 2, 3, 1)                                   # [B,H,W,Co]
    dw = (d.reshape(-1, d.shape[-1]).T @ cols.reshape(-1, cols.shape[-1])).reshape(w.shape)
    db = d.sum((0, 1, 2))
    dx = None
    if need_dx:
        dcols = (d @ w.reshape(w.shape[0], -1)).reshape(B, H, W, C, 3, 3)
        dxp = np.zeros((B, C, H + 2, W + 2), dy.dtype)
        for kh in range(3):
            for kw in range(3):
                dxp[:, :, kh:kh + H, kw:kw + W] += dcols[:, :, :, :, kh, kw].transpose(0, 3, 1, 2)
        dx = dxp[:, :, 1:-1, 1:-1]
    return dx, dw, db


def _pool_fwd(a):
    B, C, H, W = a.shape
    Hp, Wp = H // 2, W // 2
    v = a[:, :, :Hp * 2, :Wp * 2].reshape(B, C, Hp, 2, Wp, 2).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, Hp, Wp, 4)
    idx = v.argmax(-1)                                             # first maximum wins (PyTorch tie rule)
    return np.take_along_axis(v, idx[..., None], -1)[..., 0], idx


def _pool_bwd(dp, idx, shape):
    B, C, H, W = shape
    Hp, Wp = H // 2, W // 2
    dv = np.zeros((B, C, Hp, Wp, 4), dp.dtype)
    np.put_along_axis(dv, idx[..., None], dp[..., None], -1)
    da = np.zeros(shape, dp.dtype)
    da[:, :, :Hp * 2, :Wp * 2] = dv.reshape(B, C, Hp, Wp, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, Hp * 2, Wp * 2)
    return da


class CNNOracle:
    def __init__(self, state, dtype=np.float64):
        self.dtype = dtype
        self.p = {k: np.asarray(state[k], dtype).copy() for k in PARAMS}
        self.buf = {k: np.asarray(state[k]).copy() for k in BUFFERS}
        for k in BUFFERS:
            if not k.endswith("num_batches_tracked"):
                self.buf[k] = self.buf[k].astype(dtype)
        self.m = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.t = 0

    def _block_fwd(self, x, pre, train):
        w, b, g, be = (self.p[f"{pre}.0.weight"], self.p[f"{pre}.0.bias"], self.p[f"{pre}.1.weight"], self.p[f"{pre}.1.bias"])
        y, cols = _conv_fwd(x, w, b)
        if train:
            mean = y.mean((0, 2, 3)); var = y.var((0, 2, 3))
            n = y.shape[0] * y.shape[2] * y.shape[3]
            self.buf[f"{pre}.1.running_mean"] = 0.9 * self.buf[f"{pre}.1.running_mean"] + 0.1 * mean
            self.buf[f"{pre}.1.running_var"] = 0.9 * self.buf[f"{pre}.1.running_var"] + 0.1 * var * n / max(n - 1, 1)
            self.buf[f"{pre}.1.num_batches_tracked"] = self.buf[f"{pre}.1.num_batches_tracked"] + 1
        else:
            mean, var = self.buf[f"{pre}.1.running_mean"], self.buf[f"{pre}.1.running_var"]
        rstd = 1.0 / np.sqrt(var + 1e-5)
        xhat = (y - mean[None, :, None, None]) * rstd[None, :, None, None]
        bn = xhat * g[None, :, None, None] + be[None, :, None, None]
        a = np.maximum(bn, 0.0)
        pooled, idx = _pool_fwd(a)
        return pooled, dict(x=x, cols=cols, xhat=xhat, rstd=rstd, bn=bn, idx=idx, shape=a.shape)

    def forward(self, xin, train=False, drop_keep=None, drop_p=0.0):
        """logits [B,2].  drop_keep: None or [B,128] 0/1 keep mask of fc1's Dropout (train mode)."""
        xin = np.asarray(xin, self.dtype)
        h1, c1 = self._block_fwd(xin, "conv1", train)
        h2, c2 = self._block_fwd(h1, "conv2", train)
        flat = h2.reshape(h2.shape[0], -1)
        z1 = flat @ self.p["fc1.0.weight"].T + self.p["fc1.0.bias"]
        a1 = np.maximum(z1, 0.0)
        if drop_keep is not None:
            a1 = a1 * np.asarray(drop_keep, self.dtype) / (1.0 - drop_p)
        logits = a1 @ self.p["fc2.weight"].T + self.p["fc2.bias"]
        self.cache = dict(c1=c1, c2=c2, flat=flat, z1=z1, a1=a1, h2_shape=h2.shape, drop_keep=drop_keep, drop_p=drop_p, train=train)
        return logits

    @staticmethod
    def softmax(logits):
        e = np.exp(logits - logits.max(1, keepdims=True))
        return e / e.sum(1, keepdims=True)

    def loss_and_grads(self, logits, y):
        """nn.CrossEntropyLoss() (mean) and the gradients of all 12 parameter tensors (train-mode BatchNorm)."""
        c = self.cache
        B = logits.shape[0]
        p = self.softmax(logits)
        loss = float(-np.log(p[np.arange(B), y]).mean())
        dl = p.copy(); dl[np.arange(B), y] -= 1.0; dl /= B
        g = {}
        g["fc2.weight"] = dl.T @ c["a1"]; g["fc2.bias"] = dl.sum(0)
        da1 = dl @ self.p["fc2.weight"]
        if c["drop_keep"] is not None:
            da1 = da1 * np.asarray(c["drop_keep"], self.dtype) / (1.0 - c["drop_p"])
        dz1 = da1 * (c["z1"] > 0)
        g["fc1.0.weight"] = dz1.T @ c["flat"]; g["fc1.0.bias"] = dz1.sum(0)
        dh = (dz1 @ self.p["fc1.0.weight"]).reshape(c["h2_shape"])
        for pre, cc, need_dx in (("conv2", c["c2"], True), ("conv1", c["c1"], False)):
            da = _pool_bwd(dh, cc["idx"], cc["shape"]) * (cc["bn"] > 0)
            gam = self.p[f"{pre}.1.weight"]
            g[f"{pre}.1.weight"] = (da * cc["xhat"]).sum((0, 2, 3)); g[f"{pre}.1.bias"] = da.sum((0, 2, 3))
            assert c["train"], "backward is defined for train-mode BatchNorm (batch statistics)"
            dxh = da * gam[None, :, None, None]
            m1 = dxh.mean((0, 2, 3), keepdims=True); m2 = (dxh * cc["xhat"]).mean((0, 2, 3), keepdims=True)
            dy = cc["rstd"][None, :, None, None] * (dxh - m1 - cc["xhat"] * m2)
            dx, dw, db = _conv_bwd(dy, cc["cols"], self.p[f"{pre}.0.weight"], cc["x"].shape, need_dx)
            g[f"{pre}.0.weight"] = dw; g[f"{pre}.0.bias"] = db
            dh = dx
        return loss, g

    def adam_step(self, grads, lr=1e-4, wd=5e-5, b1=0.9, b2=0.999, eps=1e-8):
        self.t += 1
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k in PARAMS:
            gk = grads[k] + wd * self.p[k]
            self.m[k] = b1 * self.m[k] + (1 - b1) * gk
            self.v[k] = b2 * self.v[k] + (1 - b2) * gk * gk
            self.p[k] = self.p[k] - (lr / bc1) * self.m[k] / (np.sqrt(self.v[k]) / np.sqrt(bc2) + eps)

    def state(self):
        d = dict(self.p); d.update(self.buf)
        return d
