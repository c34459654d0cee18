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
        # loss, recon, kl, grad_norm, [4] = the plan's sticky hand-off status word (uint32 bits, 0 = healthy): it rides in the buffer
        # the caller reads back anyway, so a timed-out layer hand-off is seen without any extra synchronisation
        self.stats = torch.zeros(5, dtype=torch.float32, device=dev)
        self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._err_event = None
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
        self._raise_if_failed(wait=False)      # the status word of an EARLIER step, if its copy has landed: no synchronisation
        self.step_count += 1
        if drop_seed is None:
            drop_seed = (self.seed << 20) + self.step_count
        train_mode = self.model.training and self.model.dropout > 0.0
        # eps None: the reparameterisation noise is drawn inside the head kernel (counter RNG keyed by this rank's seed and the step) —
        # no ATen RNG launch on the timed path; an injected eps (parity tests) takes precedence
        _, _, _, _, pl, xc = eng.forward(x, eps, training_plan=True, train_mode=train_mode, drop_seed=drop_seed,
                                         want_recon=False, eps_stream=(self.seed, self.step_count))
        eng.loss(pl, xc, float(kl_w), self.stats)
        eng.backward(pl, xc, self.grads)
        scale = self.comm(self.grads)
        b1, b2 = self.betas
        # the plan's sticky status word rides into stats[4] with the optimiser's last launch (no 4-byte copy launch per step)
        off = int(lib.shm_vae_plan_status_offset(pl.handle))
        check(lib.shm_adam_step_status(C.c_int64(eng.n_flat), _ptr(self.flat), _ptr(self.grads), _ptr(self.exp_avg),
                                       _ptr(self.exp_avg_sq), self.step_count, C.c_float(self.lr), C.c_float(b1), C.c_float(b2),
                                       C.c_float(self.eps), C.c_float(self.wd),
                                       C.c_float(self.max_norm if self.max_norm is not None else 0.0), C.c_float(scale),
                                       _ptr(self.scratch), C.c_void_p(self.stats.data_ptr() + 12),
                                       C.c_void_p(pl.ws.data_ptr() + off) if off >= 0 else None, C.c_void_p(self.stats.data_ptr() + 16),
                                       _stream()), "shm_adam_step_status")
        eng.mark_dirty()
        # a host copy of the word for step()'s own check: every 4th step (the word is sticky, so nothing is missed; callers that read
        # the returned statistics see it every step anyway)
        if self.step_count % 4 == 1:
            self._err_host.copy_(self.stats[4:5].view(torch.int32), non_blocking=True)
            self._err_event = torch.cuda.Event()
            self._err_event.record()
        return self.stats

    def _raise_if_failed(self, wait: bool):
        ev = self._err_event
        if ev is None or not (wait or ev.query()):
            return
        if wait:
            ev.synchronize()
        w = int(self._err_host.item()) & 0xFFFFFFFF
        if w:
            raise RuntimeError(f"VAETrainer: a layer-pipelined launch reported a hand-off timeout (status word 0x{w:x}); the parameters, "
                               "optimiser state and statistics of every step since then are invalid")

    def check(self, host_stats=None):
        """Raises if any step so far ran with a failed in-launch hand-off.  host_stats: a host copy of the tensor step() returned
        (then no synchronisation happens here); None: waits for the last step."""
        if host_stats is not None:
            import numpy as np
            w = int(np.asarray(host_stats, dtype=np.float32).view(np.uint32)[4])
            if w:
                raise RuntimeError(f"VAETrainer: hand-off timeout in a layer-pipelined launch (status word 0x{w:x}); results are invalid")
            return
        w = int(self.stats[4:5].view(torch.int32).item()) & 0xFFFFFFFF        # blocking read of the word the last step left
        if w:
            raise RuntimeError(f"VAETrainer: a layer-pipelined launch reported a hand-off timeout (status word 0x{w:x}); the parameters, "
                               "optimiser state and statistics of every step since then are invalid")

    @torch.no_grad()
    def evaluate(self, x: torch.Tensor, kl_w: float, eps: torch.Tensor | None = None):
        """Validation pass of 03_train_vae.py:284-299: same loss, no state saved, no update.  Returns [loss, recon, kl, status] where
        status carries the inference plan's sticky hand-off word (uint32 bits, 0 = healthy; VAETrainer.check_eval raises on it)."""
        eng = self.engine
        B = x.size(0)
        self._eval_count = getattr(self, "_eval_count", 0) + 1
        _, mu, lv, score, pl, xc = eng.forward(x, eps, training_plan=False, train_mode=False, want_score=True, want_recon=False,
                                               eps_stream=(self.seed ^ 0x5EED5EED, self._eval_count))
        out = torch.empty(4, dtype=torch.float32, device=x.device)
        check(lib.shm_vae_eval_loss(_ptr(mu), _ptr(lv), _ptr(score), B, xc.size(1), xc.size(2), eng.cfg.latent_dim, C.c_float(float(kl_w)),
                                    _ptr(out), _stream()), "shm_vae_eval_loss")
        eng.status_word_into(pl, out[3:4].view(torch.int32))
        return out

    @staticmethod
    def check_eval(host_out):
        import numpy as np
        w = int(np.asarray(host_out, dtype=np.float32).view(np.uint32)[3])
        if w:
            raise RuntimeError(f"VAETrainer.evaluate: hand-off timeout in a layer-pipelined launch (status word 0x{w:x}); results are invalid")


class CNNTrainer:
    """Fused CNN train step (reference 4DOF/Scripts/05_train_cnn.py:270-276): forward (batch-stat BatchNorm, dropout) ->
    CrossEntropy -> backward -> [flat gradient all-reduce] -> Adam(lr 1e-4, wd 5e-5), no clipping.  BatchNorm statistics stay
    per-rank under data parallelism (plain DDP semantics)."""

    def __init__(self, model, lr=1e-4, weight_decay=5e-5, betas=(0.9, 0.999), eps=1e-8, seed=42, group=None, sync_params=True,
                 report_grad_norm=False):
        """report_grad_norm: also return the gradient 2-norm in stats[1] (one more launch per step; the reference step neither clips
        nor logs it, so it is off by default and stats[1] stays 0)."""
        self.model = model
        self.engine = model._engine
        self.report_grad_norm = bool(report_grad_norm)
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
        self._nbt_pending = 0
        model.register_state_dict_pre_hook(lambda module, prefix, keep_vars: self.flush_counters())
        self.comm = FlatAllReduce(group)
        rank = self.comm.dist.get_rank(group) if self.comm.world > 1 else 0
        self.seed = seed + 7919 * rank
        if sync_params:
            self.comm.broadcast(self.flat, 0)
            self.comm.broadcast(self.engine.running, 0)

    def sync_running_stats(self):
        """Data parallelism: BatchNorm batch statistics stay local to a rank while training (plain DDP semantics), so the running
        statistics drift apart.  Rank 0's are THE model's (torch DDP broadcasts buffers from rank 0 too): this broadcasts them, and
        evaluate() / a checkpoint writer call it first so that every rank computes the same validation loss and takes the same
        early-stopping / best-checkpoint branch (ranks that disagree there would meet unevenly in the next gradient all-reduce)."""
        self.comm.broadcast(self.engine.running, 0)

    def step(self, x: torch.Tensor, y: torch.Tensor, drop_seed: int | None = None):
        """One optimiser step on the local shard; returns the device tensor [loss, grad_norm]."""
        eng = self.engine
        if eng.bind(self._named, self._bufs) is not self.flat:
            raise RuntimeError("model parameters were re-allocated after the trainer was built")
        eng.cfg.dropout = float(self.model.fc1[2].p)
        self.step_count += 1
        if drop_seed is None:
            drop_seed = (self.seed << 20) + self.step_count
        _, pl, xc = eng.forward(x, training_plan=True, train_mode=True, drop_seed=drop_seed, want_logits=False)
        eng.loss(pl, y.contiguous().to(torch.int64), self.stats)
        eng.backward(pl, xc, self.grads)
        scale = self.comm(self.grads)
        b1, b2 = self.betas
        check(lib.shm_adam_step(C.c_int64(eng.n_flat), _ptr(self.flat), _ptr(self.grads), _ptr(self.exp_avg),
                                _ptr(self.exp_avg_sq), self.step_count, C.c_float(self.lr), C.c_float(b1), C.c_float(b2),
                                C.c_float(self.eps), C.c_float(self.wd), C.c_float(0.0), C.c_float(scale),
                                _ptr(self.scratch), C.c_void_p(self.stats.data_ptr() + 4 if self.report_grad_norm else None), _stream()),
              "shm_adam_step")
        # BatchNorm2d.num_batches_tracked (informational: momentum is the constant 0.1) is counted on the host and written into the two
        # int64 buffers only when somebody looks (state_dict / evaluate): two ATen launches per step otherwise
        self._nbt_pending += 1
        return self.stats

    def flush_counters(self):
        if self._nbt_pending:
            self.model.conv1[1].num_batches_tracked += self._nbt_pending
            self.model.conv2[1].num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0

    @torch.no_grad()
    def evaluate(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """Validation CrossEntropy of 05_train_cnn.py:284-292 (eval-mode forward, running statistics); device scalar."""
        self.sync_running_stats()
        self.flush_counters()
        self.model.eval()
        logits = self.model(x)
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        check(lib.shm_cnn_eval_loss(_ptr(logits), _ptr(y.contiguous().to(torch.int64)), logits.size(0), logits.size(1), _ptr(out), _stream()),
              "shm_cnn_eval_loss")
        return out
