"""Host side of the LSTM-VAE hot path: plans, workspaces and the autograd bridge over the C ABI.

PyTorch is plumbing here (device memory, streams, autograd graph edges); every FLOP of the path runs in
libshm_hip.so.  Non-CUDA tensors are refused — there is no CPU implementation in the product.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi
from ._capi import lib, check


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Plan:
    """One (B, T, training) instance: opaque C plan + torch-owned workspace."""

    def __init__(self, cfg, B, T, training, device):
        self.handle = C.c_void_p(0)
        check(lib.shm_vae_plan_create(C.byref(cfg), B, T, int(training), C.byref(self.handle)), "shm_vae_plan_create")
        self.bytes = int(lib.shm_vae_plan_workspace_bytes(self.handle))
        self.ws = torch.empty(self.bytes, dtype=torch.uint8, device=device)
        assert self.ws.data_ptr() % 256 == 0
        self.packed_version = None
        self.B, self.T, self.training = B, T, bool(training)

    def __del__(self):
        if getattr(self, "handle", None) and self.handle.value:
            lib.shm_vae_plan_destroy(self.handle)
            self.handle = C.c_void_p(0)


class VAEEngine:
    """Owns the flat parameter buffer (reference state_dict order) and dispatches to the HIP library."""

    MAX_PLANS = 8

    def __init__(self, input_dim, latent_dim, hidden_dim, num_layers, layer_norm, dropout, precision=_capi.SHM_PREC_F32):
        self.cfg = _capi.make_config(input_dim, latent_dim, hidden_dim, num_layers, layer_norm, dropout, precision)
        self.layout, self.n_flat = _capi.param_layout(self.cfg)
        self.flat = None            # torch fp32 [n_flat] on the device
        self._manual_version = 0
        self._param_version = 0
        self._plans = {}
        self.last_plan = None

    # -- parameters --------------------------------------------------------------------------------------
    def bind_parameters(self, named_params):
        """Make every nn.Parameter a view into one flat device buffer (no copies per step afterwards)."""
        params = dict(named_params)
        dev = next(iter(params.values())).device
        if dev.type != "cuda":
            raise RuntimeError("hybrid-vae-cnn-for-shm_amd runs on MI355X only: move the module to a CUDA (ROCm) device first")
        ok = self.flat is not None and self.flat.device == dev
        if ok:
            base = self.flat.data_ptr()
            for name, off, shape in self.layout:
                p = params[name]
                if p.data_ptr() != base + 4 * off or tuple(p.shape) != shape or not p.is_contiguous():
                    ok = False
                    break
        if not ok:
            flat = torch.zeros(self.n_flat, dtype=torch.float32, device=dev)
            for name, off, shape in self.layout:
                p = params[name]
                if tuple(p.shape) != shape or p.dtype != torch.float32:
                    raise RuntimeError(f"parameter {name}: expected fp32 {shape}, got {p.dtype} {tuple(p.shape)}")
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(shape)
            self.flat = flat
            self._plans.clear()
        # p.data = view keeps each Parameter's OWN version counter, so in-place writes (optimizer.step,
        # load_state_dict) are seen on the parameters, not on the flat buffer
        self._param_version = sum(params[name]._version for name, _, _ in self.layout)
        return self.flat

    @property
    def version(self):
        """Changes whenever a parameter view is written in place (optimizer step, load_state_dict) or mark_dirty() is called."""
        return (self._manual_version, self._param_version, self.flat._version if self.flat is not None else -1, id(self.flat))

    def mark_dirty(self):
        """For writers that bypass torch's version counter (raw-pointer Adam in the fused trainer, p.data edits)."""
        self._manual_version += 1

    def plan(self, B, T, training):
        key = (int(B), int(T), bool(training))
        pl = self._plans.get(key)
        if pl is None:
            if len(self._plans) >= self.MAX_PLANS:
                self._plans.pop(next(iter(self._plans)))
            pl = _Plan(self.cfg, key[0], key[1], key[2], self.flat.device)
            self._plans[key] = pl
        return pl

    def _pack(self, pl):
        if pl.packed_version != self.version:
            check(lib.shm_vae_pack_weights(pl.handle, _ptr(self.flat), _ptr(pl.ws), _stream()), "shm_vae_pack_weights")
            pl.packed_version = self.version

    # -- compute -----------------------------------------------------------------------------------------
    def _check_x(self, x):
        if x.device.type != "cuda":
            raise RuntimeError("TemporalVAE (MI355X build): input must live on a CUDA (ROCm) device; there is no CPU path")
        if x.dim() != 3 or x.size(2) != self.cfg.input_dim:
            raise RuntimeError(f"expected x of shape [B, T, {self.cfg.input_dim}], got {tuple(x.shape)}")
        return x.contiguous().float()

    def encode(self, x, train_mode=False, drop_seed=0):
        """mu, logvar of temporal_vae.py:51-58; the decoder does not run."""
        x = self._check_x(x)
        B, T, _ = x.shape
        mu = torch.empty(B, self.cfg.latent_dim, dtype=torch.float32, device=x.device)
        lv = torch.empty_like(mu)
        if B == 0:
            return mu, lv
        pl = self.plan(B, T, False)
        self._pack(pl)
        check(lib.shm_vae_encode(pl.handle, _ptr(x), _ptr(mu), _ptr(lv), int(bool(train_mode)), C.c_uint64(int(drop_seed) & (2 ** 64 - 1)),
                                 _ptr(pl.ws), _stream()), "shm_vae_encode")
        self.last_plan = pl
        return mu, lv

    def decode(self, z, seq_len, train_mode=False, drop_seed=0):
        """recon [B, seq_len, D] of temporal_vae.py:65-70; the encoder does not run."""
        if z.device.type != "cuda":
            raise RuntimeError("TemporalVAE (MI355X build): z must live on a CUDA (ROCm) device; there is no CPU path")
        if z.dim() != 2 or z.size(1) != self.cfg.latent_dim:
            raise RuntimeError(f"expected z of shape [B, {self.cfg.latent_dim}], got {tuple(z.shape)}")
        z = z.contiguous().float()
        B = z.size(0)
        recon = torch.empty(B, int(seq_len), self.cfg.input_dim, dtype=torch.float32, device=z.device)
        if B == 0:
            return recon
        pl = self.plan(B, int(seq_len), False)
        self._pack(pl)
        check(lib.shm_vae_decode(pl.handle, _ptr(z), _ptr(recon), int(bool(train_mode)), C.c_uint64(int(drop_seed) & (2 ** 64 - 1)),
                                 _ptr(pl.ws), _stream()), "shm_vae_decode")
        self.last_plan = pl
        return recon

    def forward(self, x, eps, training_plan, train_mode, drop_seed=0, want_score=False, want_recon=True, eps_stream=None):
        """eps: injected noise [B, Z]; or None with eps_stream = (seed, offset): drawn on the device inside the head kernel."""
        x = self._check_x(x)
        B, T, D = x.shape
        if B == 0:      # an empty batch (the reference's modules return empty tensors): nothing to launch
            Z = self.cfg.latent_dim
            e = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=x.device)
            return (e(0, T, D) if want_recon else None), e(0, Z), e(0, Z), (e(0) if want_score else None), None, x
        if eps is not None:
            eps = eps.contiguous().float()
            if tuple(eps.shape) != (B, self.cfg.latent_dim):
                raise RuntimeError(f"eps must be [{B}, {self.cfg.latent_dim}], got {tuple(eps.shape)}")
        elif eps_stream is None:
            raise RuntimeError("forward needs either eps or eps_stream = (seed, offset)")
        pl = self.plan(B, T, training_plan)
        self._pack(pl)
        Z = self.cfg.latent_dim
        recon = torch.empty(B, T, D, dtype=torch.float32, device=x.device) if want_recon else None
        mu = torch.empty(B, Z, dtype=torch.float32, device=x.device)
        lv = torch.empty(B, Z, dtype=torch.float32, device=x.device)
        score = torch.empty(B, dtype=torch.float32, device=x.device) if want_score else None
        if eps is not None:
            check(lib.shm_vae_forward(pl.handle, _ptr(x), _ptr(eps), _ptr(recon), _ptr(mu), _ptr(lv), _ptr(score),
                                      int(bool(train_mode)), C.c_uint64(int(drop_seed) & (2 ** 64 - 1)), _ptr(pl.ws), _stream()),
                  "shm_vae_forward")
        else:
            check(lib.shm_vae_forward_sample(pl.handle, _ptr(x), C.c_uint64(int(eps_stream[0]) & (2 ** 64 - 1)),
                                             C.c_uint64(int(eps_stream[1]) & (2 ** 64 - 1)), _ptr(recon), _ptr(mu), _ptr(lv), _ptr(score),
                                             int(bool(train_mode)), C.c_uint64(int(drop_seed) & (2 ** 64 - 1)), _ptr(pl.ws), _stream()),
                  "shm_vae_forward_sample")
        self.last_plan = pl
        return recon, mu, lv, score, pl, x

    def loss(self, pl, x, kl_w, losses):
        check(lib.shm_vae_loss(pl.handle, _ptr(x), C.c_float(kl_w), _ptr(losses), _ptr(pl.ws), _stream()), "shm_vae_loss")

    def backward(self, pl, x, grads, d_recon=None, d_mu=None, d_lv=None):
        check(lib.shm_vae_backward(pl.handle, _ptr(x), _ptr(d_recon), _ptr(d_mu), _ptr(d_lv), _ptr(grads), _ptr(pl.ws),
                                   _stream()), "shm_vae_backward")

    def check_status(self):
        """Raises if a layer-pipelined launch of any plan EVER reported a hand-off timeout (synchronises the stream).  The word is
        sticky: re-packing the weights after an optimiser step does not clear it, only reset_status() does."""
        for pl in self._plans.values():
            check(lib.shm_vae_plan_status(pl.handle, _ptr(pl.ws), _stream()), "shm_vae_plan_status")

    def status_word_into(self, pl, dst_i32):
        """Copies plan pl's sticky status word (0 = healthy) into the 4-byte device location dst_i32, asynchronously on the stream."""
        check(lib.shm_vae_plan_status_word(pl.handle, _ptr(pl.ws), C.c_void_p(dst_i32.data_ptr()), _stream()), "shm_vae_plan_status_word")

    def reset_status(self):
        for pl in self._plans.values():
            check(lib.shm_vae_plan_reset_status(pl.handle, _ptr(pl.ws), _stream()), "shm_vae_plan_reset_status")

    def split(self, flat):
        return [flat[off:off + int(torch.Size(shape).numel())].view(shape) for _, off, shape in self.layout]


class VAEFunction(torch.autograd.Function):
    """recon, mu, logvar = f(x, eps; params) with a hand-written HIP backward (loss.backward() keeps working)."""

    @staticmethod
    def forward(ctx, engine, x, eps, train_mode, drop_seed, *params):
        recon, mu, lv, _, pl, xc = engine.forward(x, eps, training_plan=True, train_mode=train_mode, drop_seed=drop_seed)
        ctx.engine, ctx.pl, ctx.version = engine, pl, engine.version
        ctx.save_for_backward(xc)
        return recon, mu, lv

    @staticmethod
    def backward(ctx, d_recon, d_mu, d_lv):
        eng, pl = ctx.engine, ctx.pl
        if eng.last_plan is not pl or eng.version != ctx.version:
            raise RuntimeError("TemporalVAE backward: another forward ran on this module since the one being "
                               "differentiated (the HIP workspace holds one forward at a time)")
        (x,) = ctx.saved_tensors
        grads = torch.zeros(eng.n_flat, dtype=torch.float32, device=x.device)
        eng.backward(pl, x, grads, d_recon.contiguous().float(), d_mu.contiguous().float(), d_lv.contiguous().float())
        return (None, None, None, None, None, *eng.split(grads))
