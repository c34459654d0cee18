"""MI355X drop-in for the 4DOF fault classifier module (reference 4DOF/Scripts/Models/cnn_model.py).

What is kept from the reference, because callers and checkpoints depend on it: the import path, `SEQ_LEN` / `NUM_FEATURES`, the
class names `CNN` / `CNNClassifier`, the constructor signature and defaults (cnn_model.py:14), the 18 state_dict entries with their
names, shapes and dtypes (conv{1,2}.0 = Conv2d 3x3 pad 1, conv{1,2}.1 = BatchNorm2d incl. running statistics and the int64
num_batches_tracked, fc1.0 = Linear, fc2 = Linear; cnn_model.py:16-34) and the initialisation (xavier-uniform weights, zero biases;
cnn_model.py:39-43).  What is different: nothing here computes — forward() hands the flat parameter buffer to libshm_hip.so
(conv+BN+ReLU+MaxPool x2, dense, dropout, dense; shm_cnn_forward / shm_cnn_backward), and the dense layer's input width follows the
window shape (32 * (T // 4) * (D // 4); the reference hard-codes 32*25*3 for 100 x 12) through the keyword-only seq_len /
num_features so that the north-star 256 x 4 windows can be classified too.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from shm_hip.cnn import CNNEngine, CNNFunction

SEQ_LEN = 100
NUM_FEATURES = 12

_CONV_CHANNELS = (16, 32)      # output channels of the two conv stages
_FC_HIDDEN = 128


def _stage(cin: int, cout: int) -> nn.Sequential:
    """Parameter holders of one conv stage, at the child indices the reference's checkpoints use (0: conv, 1: batch norm)."""
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout))


class CNN(nn.Module):
    """(B, input_channels, T, D) windows -> (B, num_classes) logits; T x D = SEQ_LEN x NUM_FEATURES unless overridden."""

    def __init__(self, input_channels: int = 2, num_classes: int = 2, dropout_rate: float = 0.5, *,
                 seq_len: int | None = None, num_features: int | None = None):
        super().__init__()
        T = int(SEQ_LEN if seq_len is None else seq_len)
        D = int(NUM_FEATURES if num_features is None else num_features)
        c1, c2 = _CONV_CHANNELS
        self.conv1 = _stage(input_channels, c1)
        self.conv2 = _stage(c1, c2)
        # fc1 keeps the reference's three-slot layout (0: Linear, 2: Dropout) because training code reads fc1[2].p
        self.fc1 = nn.Sequential(nn.Linear(c2 * (T // 4) * (D // 4), _FC_HIDDEN), nn.Identity(), nn.Dropout(dropout_rate))
        self.fc2 = nn.Linear(_FC_HIDDEN, num_classes)
        for mod in self.modules():
            if isinstance(mod, (nn.Conv2d, nn.Linear)):
                nn.init.xavier_uniform_(mod.weight)
                nn.init.zeros_(mod.bias)
        self._engine = CNNEngine(input_channels, num_classes, T, D, dropout_rate)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.device.type != "cuda":
            raise RuntimeError("CNN (MI355X build) has no CPU path: call .to('cuda') on the module and the input")
        eng = self._engine
        params = dict(self.named_parameters())
        eng.bind(params, dict(self.named_buffers()))
        eng.cfg.dropout = float(self.fc1[2].p)
        training = self.training
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (training and eng.cfg.dropout > 0) else 0
        ordered = [params[name] for name, _, _ in eng.layout]
        if torch.is_grad_enabled() and any(p.requires_grad for p in ordered):
            logits = CNNFunction.apply(eng, x, training, seed, *ordered)
        else:
            logits, _, _ = eng.forward(x, training_plan=False, train_mode=training, drop_seed=seed)
        if training:        # BatchNorm2d bookkeeping; momentum is the constant 0.1, so the counter is informational
            for stage in (self.conv1, self.conv2):
                stage[1].num_batches_tracked += 1
        return logits


class CNNClassifier(CNN):
    """Alias kept for scripts that import the older name (reference cnn_model.py:54-57)."""

    def __init__(self, dropout_rate: float = 0.5):
        super().__init__(2, 2, dropout_rate)
