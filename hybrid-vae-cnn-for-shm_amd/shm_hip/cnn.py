"""Host side of the CNN classifier: plans, workspaces and the autograd bridge over the C ABI (include/shm_hip.h).

No CPU implementation: non-CUDA tensors are refused.  torch supplies device memory, streams and autograd edges only.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi
from ._capi import lib, check
from .vae import _ptr, _stream

BUFFER_SLOTS = (("conv1.1.running_mean", 0, 16), ("conv1.1.running_var", 16, 16),
                ("conv2.1.running_mean", 32, 32), ("conv2.1.running_var", 64, 32))


class _CnnPlan:
    def __init__(self, cfg, B, training, device):
        self.handle = C.c_void_p(0)
        check(lib.shm_cnn_plan_create(C.byref(cfg), B, int(training), C.byref(self.handle)), "shm_cnn_plan_create")
        self.ws = torch.empty(int(lib.shm_cnn_plan_workspace_bytes(self.handle)), dtype=torch.uint8, device=device)
        self.B, self.training = B, bool(training)

    def __del__(self):
        if getattr(self, "handle", None) and self.handle.value:
            lib.shm_cnn_plan_destroy(self.handle)
            self.handle = C.c_void_p(0)


class CNNEngine:
    MAX_PLANS = 8

    def __init__(self, in_channels, num_classes, seq_len, num_features, dropout):
        self.cfg = _capi.CnnConfig(int(in_channels), int(num_classes), int(seq_len), int(num_features), float(dropout))
        self.layout, self.n_flat = _capi.cnn_param_layout(self.cfg)
        self.flat = None
        self.running = None
        self._plans = {}
        self.last_plan = None

    def bind(self, named_params, named_buffers):
        """Parameters become views into one flat fp32 buffer; the four BatchNorm running tensors views into `running`."""
        params, bufs = dict(named_params), dict(named_buffers)
        dev = next(iter(params.values())).device
        if dev.type != "cuda":
            raise RuntimeError("CNN (MI355X build) runs on a CUDA (ROCm) device only")
        ok = self.flat is not None and self.flat.device == dev
        if ok:
            base, rbase = self.flat.data_ptr(), self.running.data_ptr()
            ok = all(params[n].data_ptr() == base + 4 * off and params[n].is_contiguous() for n, off, _ in self.layout) and \
                all(bufs[n].data_ptr() == rbase + 4 * off for n, off, _ in BUFFER_SLOTS)
        if not ok:
            flat = torch.zeros(self.n_flat, dtype=torch.float32, device=dev)
            for name, off, shape in self.layout:
                p = params[name]
                if tuple(p.shape) != shape or p.dtype != torch.float32:
                    raise RuntimeError(f"parameter {name}: expected fp32 {shape}, got {p.dtype} {tuple(p.shape)}")
                flat[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + p.numel()].view(shape)
            running = torch.zeros(96, dtype=torch.float32, device=dev)
            for name, off, n in BUFFER_SLOTS:
                running[off:off + n].copy_(bufs[name].detach().float())
                bufs[name].data = running[off:off + n]
            self.flat, self.running = flat, running
            self._plans.clear()
        return self.flat

    def plan(self, B, training):
        # shm_cnn_plan_create COPIES the configuration, so the dropout rate is part of the key: a caller that changes fc1[2].p after
        # the first forward gets a new plan instead of silently keeping the old rate
        key = (int(B), bool(training), round(float(self.cfg.dropout), 9))
        pl = self._plans.get(key)
        if pl is None:
            if len(self._plans) >= self.MAX_PLANS:
                self._plans.pop(next(iter(self._plans)))
            pl = self._plans[key] = _CnnPlan(self.cfg, key[0], key[1], self.flat.device)
        return pl

    def forward(self, x, training_plan, train_mode, drop_seed=0, want_logits=True):
        c = self.cfg
        if x.device.type != "cuda":
            raise RuntimeError("CNN (MI355X build): input must live on a CUDA (ROCm) device; there is no CPU path")
        if x.dim() != 4 or tuple(x.shape[1:]) != (c.in_channels, c.seq_len, c.num_features):
            raise RuntimeError(f"expected x of shape [B, {c.in_channels}, {c.seq_len}, {c.num_features}], got {tuple(x.shape)}")
        x = x.contiguous().float()
        pl = self.plan(x.size(0), training_plan)
        # want_logits=False (fused train step): the logits stay in the workspace for shm_cnn_loss, no copy-out launch
        logits = torch.empty(x.size(0), c.num_classes, dtype=torch.float32, device=x.device) if want_logits else None
        check(lib.shm_cnn_forward(pl.handle, _ptr(self.flat), _ptr(self.running), _ptr(x), _ptr(logits), int(bool(train_mode)),
                                  C.c_uint64(int(drop_seed) & (2 ** 64 - 1)), _ptr(pl.ws), _stream()), "shm_cnn_forward")
        self.last_plan = pl
        return logits, pl, x

    def loss(self, pl, labels, out):
        check(lib.shm_cnn_loss(pl.handle, _ptr(labels), _ptr(out), _ptr(pl.ws), _stream()), "shm_cnn_loss")

    def backward(self, pl, x, grads, d_logits=None):
        check(lib.shm_cnn_backward(pl.handle, _ptr(self.flat), _ptr(x), _ptr(d_logits), _ptr(grads), _ptr(pl.ws), _stream()),
              "shm_cnn_backward")

    def split(self, flat):
        return [flat[off:off + int(torch.Size(shape).numel())].view(shape) for _, off, shape in self.layout]


class CNNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, x, train_mode, drop_seed, *params):
        logits, pl, xc = engine.forward(x, training_plan=True, train_mode=train_mode, drop_seed=drop_seed)
        ctx.engine, ctx.pl = engine, pl
        ctx.save_for_backward(xc)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        eng, pl = ctx.engine, ctx.pl
        if eng.last_plan is not pl:
            raise RuntimeError("CNN backward: another forward ran on this module since the one being differentiated")
        (x,) = ctx.saved_tensors
        grads = torch.zeros(eng.n_flat, dtype=torch.float32, device=x.device)
        eng.backward(pl, x, grads, d_logits.contiguous().float())
        return (None, None, None, None, *eng.split(grads))


def build_cnn_input(z: torch.Tensor, zhat: torch.Tensor) -> torch.Tensor:
    """xin = stack([z, (z - zhat)**2], dim=1) on the device (05_train_cnn.py:137-138)."""
    if z.device.type != "cuda":
        raise RuntimeError("build_cnn_input: CUDA (ROCm) tensors only")
    z, zhat = z.contiguous().float(), zhat.contiguous().float()
    B, T, D = z.shape
    xin = torch.empty(B, 2, T, D, dtype=torch.float32, device=z.device)
    check(lib.shm_cnn_build_input(B, T, D, _ptr(z), _ptr(zhat), _ptr(xin), _stream()), "shm_cnn_build_input")
    return xin


def predict(logits: torch.Tensor):
    """(argmax class [B] int64, softmax[:, 1] [B]) — 06_test_full_pipeline.py:367-371."""
    logits = logits.contiguous().float()
    B, NC = logits.shape
    cls = torch.empty(B, dtype=torch.int64, device=logits.device)
    p1 = torch.empty(B, dtype=torch.float32, device=logits.device)
    check(lib.shm_cnn_predict(B, NC, _ptr(logits), _ptr(cls), _ptr(p1), _stream()), "shm_cnn_predict")
    return cls, p1
