# Scripts/Models/cnn_model.py — MI355X drop-in for the reference module of the same path
# (reference 4DOF/Scripts/Models/cnn_model.py:1-57).
import torch
import torch.nn as nn

from shm_hip.cnn import CNNEngine, CNNFunction

SEQ_LEN = 100
NUM_FEATURES = 12


class CNN(nn.Module):
    """
    2D Convolutional Neural Network for sequence-based fault classification, computed by hand-written gfx950 kernels.
    Input: (batch, 2, SEQ_LEN, NUM_FEATURES), e.g., (batch, 2, 100, 12)      Output: (batch, 2) logits.

    Same constructor signature, defaults, state_dict keys/shapes/dtypes (incl. BatchNorm running stats and the int64
    num_batches_tracked) and xavier/zero initialisation as the reference (cnn_model.py:14-43).  The only generalisation:
    fc1's input width is 32*(SEQ_LEN//4)*(NUM_FEATURES//4) from the module constants at construction time (2400 at the
    reference's 100x12; the reference hard-codes 32*25*3, cnn_model.py:30), overridable with the keyword-only
    seq_len / num_features.  nn.Conv2d / nn.BatchNorm2d / nn.Linear are parameter containers only; forward() never calls them.
    """

    def __init__(self, input_channels: int = 2, num_classes: int = 2, dropout_rate: float = 0.5, *,
                 seq_len: int | None = None, num_features: int | None = None):
        super().__init__()
        T = SEQ_LEN if seq_len is None else int(seq_len)
        D = NUM_FEATURES if num_features is None else int(num_features)
        self.conv1 = nn.Sequential(
            nn.Conv2d(input_channels, 16, kernel_size=3, padding=1),
            nn.BatchNorm2d(16),
            nn.ReLU(),
            nn.MaxPool2d(kernel_size=2),
        )
        self.conv2 = nn.Sequential(
            nn.Conv2d(16, 32, kernel_size=3, padding=1),
            nn.BatchNorm2d(32),
            nn.ReLU(),
            nn.MaxPool2d(kernel_size=2),
        )
        self.flatten = nn.Flatten()
        self.fc1 = nn.Sequential(
            nn.Linear(32 * (T // 4) * (D // 4), 128),
            nn.ReLU(),
            nn.Dropout(dropout_rate),
        )
        self.fc2 = nn.Linear(128, num_classes)

        # IMPORTANT: This must match old behavior (cnn_model.py:36-43)
        self.apply(self._init_weights)
        self._engine = CNNEngine(input_channels, num_classes, T, D, dropout_rate)

    def _init_weights(self, m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.device.type != "cuda":
            raise RuntimeError("CNN (MI355X build) has no CPU path: call .to('cuda') on the module and the input")
        eng = self._engine
        named = dict(self.named_parameters())
        eng.bind(named, dict(self.named_buffers()))
        eng.cfg.dropout = float(self.fc1[2].p)
        train_mode = self.training
        drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (train_mode and eng.cfg.dropout > 0) else 0
        params = [named[name] for name, _, _ in eng.layout]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            out = CNNFunction.apply(eng, x, train_mode, drop_seed, *params)
        else:
            out, _, _ = eng.forward(x, training_plan=False, train_mode=train_mode, drop_seed=drop_seed)
        if train_mode:   # nn.BatchNorm2d bookkeeping (momentum is a constant 0.1, so the counter is informational)
            self.conv1[1].num_batches_tracked += 1
            self.conv2[1].num_batches_tracked += 1
        return out


# Compatibility alias so your newer scripts can still import CNNClassifier if needed
class CNNClassifier(CNN):
    def __init__(self, dropout_rate: float = 0.5):
        super().__init__(input_channels=2, num_classes=2, dropout_rate=dropout_rate)
