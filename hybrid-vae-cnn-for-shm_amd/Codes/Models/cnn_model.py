"""Drop-in for the openLAB classifier module (reference 20250506_openLAB_tests/Codes/Models/cnn_model.py:1-62), MI355X build.

Same import path (`from Models.cnn_model import CNN, SEQ_LEN, NUM_FEATURES` with `Codes/` on sys.path, as
10_test_hybrid_pipeline.py does), constructor signature, state_dict keys / shapes and initialisation (kaiming-normal weights,
zero biases; GroupNorm 1 / 0).  The forward pass runs in libshm_hip.so (shm_olcnn_forward) and is INFERENCE ONLY: the hybrid pipeline
loads a trained checkpoint and calls the model in eval mode (10_test_hybrid_pipeline.py:289-297).  Training this classifier is not
part of the MI355X path; calling it in train mode raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from shm_hip import _capi
from shm_hip._capi import check, lib

SEQ_LEN = 200
NUM_FEATURES = 4      # raw 4 channels

_BLOCKS = ((32, 7), (64, 5), (128, 5), (256, 3))       # (cout, kt); kf = 3, padding (kt // 2, 1)


class CNN(nn.Module):
    """Input (B, input_channels, T, F); output (B, num_classes) logits.  T and F are free (the reference trains at 200 x 4)."""

    def __init__(self, input_channels: int = 1, num_classes: int = 2, dropout_rate: float = 0.4):
        super().__init__()
        layers, cin = [], input_channels
        for i, (cout, kt) in enumerate(_BLOCKS):
            # parameter containers only (names features.{0,2,4,6}.{0,1}.*); the arithmetic is in libshm_hip.so
            layers.append(nn.Sequential(nn.Conv2d(cin, cout, kernel_size=(kt, 3), padding=(kt // 2, 1)), nn.GroupNorm(8, cout), nn.SiLU()))
            layers.append(nn.MaxPool2d(kernel_size=(2, 1)) if i < 3 else nn.AdaptiveAvgPool2d((1, 1)))
            cin = cout
        self.features = nn.Sequential(*layers)
        self.classifier = nn.Sequential(nn.Flatten(), nn.Linear(256, 128), nn.SiLU(), nn.Dropout(dropout_rate), nn.Linear(128, num_classes))
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
                nn.init.zeros_(m.bias)
        self.input_channels, self.num_classes = int(input_channels), int(num_classes)
        self._flat, self._flat_key, self._ws = None, None, None

    def _params_flat(self, dev, cfg):
        """state_dict order == shm_olcnn_param_layout order; re-packed only when a parameter changed (version counters)."""
        ps = list(self.parameters())
        key = (str(dev),) + tuple((p.data_ptr(), p._version) for p in ps)
        if self._flat is None or key != self._flat_key:
            n = int(lib.shm_olcnn_param_count(C.byref(cfg)))
            descs = (_capi.TensorDesc * 32)()
            k = lib.shm_olcnn_param_layout(C.byref(cfg), descs, 32)
            flat = torch.zeros(n, dtype=torch.float32, device=dev)
            sd = self.state_dict()
            for i in range(k):
                name, off = descs[i].name.decode(), int(descs[i].offset)
                t = sd[name].detach().to(dev, torch.float32).reshape(-1)
                flat[off:off + t.numel()] = t
            self._flat, self._flat_key = flat, key
        return self._flat

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise RuntimeError("openLAB CNN (MI355X build): inference only — call .eval() (the hybrid pipeline loads a trained checkpoint)")
        if x.device.type != "cuda":
            raise RuntimeError("openLAB CNN (MI355X build): input must live on a CUDA (ROCm) device; there is no CPU path")
        if x.dim() != 4 or x.size(1) != self.input_channels:
            raise RuntimeError(f"expected x of shape [B, {self.input_channels}, T, F], got {tuple(x.shape)}")
        x = x.contiguous().float()
        B, _, T, F = x.shape
        cfg = _capi.OlCnnConfig(self.input_channels, self.num_classes, T, F)
        flat = self._params_flat(x.device, cfg)
        nbytes = int(lib.shm_olcnn_workspace_bytes(C.byref(cfg), B))
        if nbytes == 0:
            raise RuntimeError("shm_olcnn_workspace_bytes: " + lib.shm_last_error().decode())
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != x.device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        logits = torch.empty(B, self.num_classes, dtype=torch.float32, device=x.device)
        check(lib.shm_olcnn_forward(C.byref(cfg), C.c_void_p(flat.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(logits.data_ptr()), B,
                                    C.c_void_p(self._ws.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "shm_olcnn_forward")
        return logits
