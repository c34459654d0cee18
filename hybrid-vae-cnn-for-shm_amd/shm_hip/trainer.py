"""Fused LSTM-VAE train step (reference 4DOF/Scripts/03_train_vae.py:260-271) and its data-parallel form.

One step = forward -> loss -> backward -> [all-reduce of ONE flat gradient bucket] -> clip_grad_norm_ -> Adam,
all through the C ABI on the current HIP stream; no host synchronisation inside a step (the reference's three
.item() calls per step, 03_train_vae.py:275-277, become one read of a 4-float device buffer when the caller asks).

Data parallelism: one process per GPU, every rank holds the full 1.9 MB model, windows are
sharded by rank, gradients are summed with a single torch.distributed all-reduce (backend "nccl" = RCCL over xGMI)
and scaled by 1/world inside the Adam kernel; every rank then applies the bit-identical update, so parameters
never need re-broadcasting after step 0.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._capi import lib, check
from .vae import _ptr, _stream


def shard_indices(n_items: int, rank: int, world: int):
    """Indices of the items rank `rank` owns: r, r+world, r+2*world, ... (SURVEY.md §8e partitioning)."""
    return range(rank, n_items, world)


class FlatAllReduce:
    """Sum-all-reduce of one flat bucket; group=None and world 1 is a no-op.  Kept separate so the CPU tests can
    drive the same exchange logic over gloo."""

    def __init__(self, group=None):
        self.dist = None
        self.world = 1
        self.group = group
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            self.dist = dist
            self.world = dist.get_world_size(group)

    def __call__(self, bucket: torch.Tensor) -> float:
        """Reduces in place; returns the scale (1/world) the optimiser must apply to turn the sum into a mean."""
        if self.world > 1:
            self.dist.all_reduce(bucket, op=self.dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world

    def broadcast(self, t: torch.Tensor, src: int = 0):
        if self.world > 1:
            self.dist.broadcast(t, src=src, group=self.group)


class VAETrainer:
    """Fused train step for a Scripts.Models.temporal_vae.TemporalVAE living on a CUDA (ROCm) device."""

    def __init__(self, model, lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999), eps=1e-8, max_norm=2.0,
                 seed=42, group=None, sync_params=True):
        self.model = model
        self.engine = model._engine
        self._named = dict(model.named_parameters())
        self.flat = self.engine.bind_parameters(self._named)
        dev = self.flat.device
        n = self.engine.n_flat
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.scratch = torch.zeros(int(lib.shm_adam_scratch_bytes(n)), dtype=torch.uint8, device=dev)
        self.stats = torch.zeros(4, dtype=torch.float32, device=dev)      # loss, recon, kl, grad_norm
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        self.step_count = 0
        self.comm = FlatAllReduce(group)
        self.gen = torch.Generator(device=dev)
        rank = self.comm.dist.get_rank(group) if self.comm.world > 1 else 0
        self.gen.manual_seed(seed + 7919 * rank)        # eps / dropout streams differ per rank (different windows)
        self.seed = seed + 7919 * rank
        if sync_params:
            self.comm.broadcast(self.flat, 0)
            self.engine.mark_dirty()

    def step(self, x: torch.Tensor, kl_w: float, eps: torch.Tensor | None = None, drop_seed: int | None = None):
        """One optimiser step on the local shard x [B,T,D]; returns the device tensor [loss, recon, kl, grad_norm]."""
        eng = self.engine
        if eng.bind_parameters(self._named) is not self.flat:
            raise RuntimeError("model parameters were re-allocated (e.g. .to()) after the trainer was built")
        B = x.size(0)
        if eps is None:
            eps = torch.randn(B, eng.cfg.latent_dim, dtype=torch.float32, device=x.device, generator=self.gen)
        self.step_count += 1
        if drop_seed is None:
            drop_seed = (self.seed << 20) + self.step_count
        train_mode = self.model.training and self.model.dropout > 0.0
        _, _, _, _, pl, xc = eng.forward(x, eps, training_plan=True, train_mode=train_mode, drop_seed=drop_seed,
                                         want_recon=False)
        eng.loss(pl, xc, float(kl_w), self.stats)
        eng.backward(pl, xc, self.grads)
        scale = self.comm(self.grads)
        b1, b2 = self.betas
        check(lib.shm_adam_step(C.c_int64(eng.n_flat), _ptr(self.flat), _ptr(self.grads), _ptr(self.exp_avg),
                                _ptr(self.exp_avg_sq), self.step_count, C.c_float(self.lr), C.c_float(b1), C.c_float(b2),
                                C.c_float(self.eps), C.c_float(self.wd),
                                C.c_float(self.max_norm if self.max_norm is not None else 0.0), C.c_float(scale),
                                _ptr(self.scratch), C.c_void_p(self.stats.data_ptr() + 12), _stream()), "shm_adam_step")
        eng.mark_dirty()
        return self.stats

    @torch.no_grad()
    def evaluate(self, x: torch.Tensor, kl_w: float, eps: torch.Tensor | None = None):
        """Validation pass of 03_train_vae.py:284-299: same loss, no state saved, no update. Returns [loss, recon, kl]."""
        eng = self.engine
        B = x.size(0)
        if eps is None:
            eps = torch.randn(B, eng.cfg.latent_dim, dtype=torch.float32, device=x.device, generator=self.gen)
        _, mu, lv, score, _, xc = eng.forward(x, eps, training_plan=False, train_mode=False, want_score=True, want_recon=False)
        out = torch.empty(3, dtype=torch.float32, device=x.device)
        check(lib.shm_vae_eval_loss(_ptr(mu), _ptr(lv), _ptr(score), B, xc.size(1), xc.size(2), eng.cfg.latent_dim, C.c_float(float(kl_w)),
                                    _ptr(out), _stream()), "shm_vae_eval_loss")
        return out


class CNNTrainer:
    """Fused CNN train step (reference 4DOF/Scripts/05_train_cnn.py:270-276): forward (batch-stat BatchNorm, dropout) ->
    CrossEntropy -> backward -> [flat gradient all-reduce] -> Adam(lr 1e-4, wd 5e-5), no clipping.  BatchNorm statistics stay
    per-rank under data parallelism (plain DDP semantics)."""

    def __init__(self, model, lr=1e-4, weight_decay=5e-5, betas=(0.9, 0.999), eps=1e-8, seed=42, group=None, sync_params=True):
        self.model = model
        self.engine = model._engine
        self._named, self._bufs = dict(model.named_parameters()), dict(model.named_buffers())
        self.flat = self.engine.bind(self._named, self._bufs)
        dev, n = self.flat.device, self.engine.n_flat
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.scratch = torch.zeros(int(lib.shm_adam_scratch_bytes(n)), dtype=torch.uint8, device=dev)
        self.stats = torch.zeros(2, dtype=torch.float32, device=dev)      # loss, grad_norm
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.step_count = 0
        self.comm = FlatAllReduce(group)
        rank = self.comm.dist.get_rank(group) if self.comm.world > 1 else 0
        self.seed = seed + 7919 * rank
        if sync_params:
            self.comm.broadcast(self.flat, 0)

    def step(self, x: torch.Tensor, y: torch.Tensor, drop_seed: int | None = None):
        """One optimiser step on the local shard; returns the device tensor [loss, grad_norm]."""
        eng = self.engine
        if eng.bind(self._named, self._bufs) is not self.flat:
            raise RuntimeError("model parameters were re-allocated after the trainer was built")
        eng.cfg.dropout = float(self.model.fc1[2].p)
        self.step_count += 1
        if drop_seed is None:
            drop_seed = (self.seed << 20) + self.step_count
        _, pl, xc = eng.forward(x, training_plan=True, train_mode=True, drop_seed=drop_seed)
        eng.loss(pl, y.contiguous().to(torch.int64), self.stats)
        eng.backward(pl, xc, self.grads)
        scale = self.comm(self.grads)
        b1, b2 = self.betas
        check(lib.shm_adam_step(C.c_int64(eng.n_flat), _ptr(self.flat), _ptr(self.grads), _ptr(self.exp_avg),
                                _ptr(self.exp_avg_sq), self.step_count, C.c_float(self.lr), C.c_float(b1), C.c_float(b2),
                                C.c_float(self.eps), C.c_float(self.wd), C.c_float(0.0), C.c_float(scale),
                                _ptr(self.scratch), C.c_void_p(self.stats.data_ptr() + 4), _stream()), "shm_adam_step")
        self.model.conv1[1].num_batches_tracked += 1
        self.model.conv2[1].num_batches_tracked += 1
        return self.stats

    @torch.no_grad()
    def evaluate(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """Validation CrossEntropy of 05_train_cnn.py:284-292 (eval-mode forward, running statistics); device scalar."""
        self.model.eval()
        logits = self.model(x)
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        check(lib.shm_cnn_eval_loss(_ptr(logits), _ptr(y.contiguous().to(torch.int64)), logits.size(0), logits.size(1), _ptr(out), _stream()),
              "shm_cnn_eval_loss")
        return out
