"""torch-CPU restatement of the CNN train step — TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT CODE.

Only tests/ and bench.py's cpu_baseline leg (--workload cnn) import this.  It drives the same third-party arithmetic the reference
runs (ATen's conv2d, batch_norm, max_pool2d, linear, dropout, cross_entropy, Adam — torch is unpinned by the reference,
4DOF/pyproject.toml:5-9) from a plain state_dict:
  forward   4DOF/Scripts/Models/cnn_model.py:45-51  (conv3x3 pad 1 -> BatchNorm2d -> ReLU -> MaxPool2d(2), twice; flatten; fc1 + ReLU +
            Dropout; fc2)
  step      4DOF/Scripts/05_train_cnn.py:270-276     (CrossEntropyLoss mean, Adam(lr 1e-4, wd 5e-5), no clipping)
Pinned by tests/test_oracle_torch_cpu.py against the goldens generated from the reference class (tests/golden/cnn_train_*.npz).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

LEARNABLE = ["conv1.0.weight", "conv1.0.bias", "conv1.1.weight", "conv1.1.bias", "conv2.0.weight", "conv2.0.bias", "conv2.1.weight",
             "conv2.1.bias", "fc1.0.weight", "fc1.0.bias", "fc2.weight", "fc2.bias"]


class TorchCpuCNN:
    def __init__(self, state: dict, dropout: float = 0.5):
        self.p = {k: torch.as_tensor(state[k], dtype=torch.float32).clone().requires_grad_(True) for k in LEARNABLE}
        self.buf = {k: torch.as_tensor(state[k], dtype=torch.float32).clone() for k in
                    ("conv1.1.running_mean", "conv1.1.running_var", "conv2.1.running_mean", "conv2.1.running_var")}
        self.dropout, self.training = dropout, True

    def parameters(self):
        return list(self.p.values())

    def _stage(self, x, n):
        p, b = self.p, self.buf
        y = F.conv2d(x, p[f"conv{n}.0.weight"], p[f"conv{n}.0.bias"], padding=1)
        y = F.batch_norm(y, b[f"conv{n}.1.running_mean"], b[f"conv{n}.1.running_var"], p[f"conv{n}.1.weight"], p[f"conv{n}.1.bias"],
                         self.training, 0.1, 1e-5)
        return F.max_pool2d(F.relu(y), 2)

    def forward(self, x):
        h = self._stage(self._stage(x, 1), 2).flatten(1)
        h = F.dropout(F.relu(F.linear(h, self.p["fc1.0.weight"], self.p["fc1.0.bias"])), self.dropout, self.training)
        return F.linear(h, self.p["fc2.weight"], self.p["fc2.bias"])

    def make_optimizer(self, lr=1e-4, wd=5e-5):
        return torch.optim.Adam(self.parameters(), lr=lr, weight_decay=wd)

    def train_step(self, opt, x, y):
        logits = self.forward(x)
        loss = F.cross_entropy(logits, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return float(loss), logits.detach()
