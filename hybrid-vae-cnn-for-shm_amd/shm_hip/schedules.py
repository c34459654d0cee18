"""KL-weight schedules of the three trainers."""
from __future__ import annotations

import math


def kl_weight_4dof(epoch: int, n_epochs: int, warmup_ratio: float = 0.30) -> float:
    """Sigmoid anneal, epoch 1-based: 4DOF/Scripts/03_train_vae.py:120-142."""
    warm = max(1, int(n_epochs * warmup_ratio))
    x = ((epoch - 1) - warm) / float(max(warm, 1))
    return float(1.0 / (1.0 + math.exp(-x * 5.0)))


def kl_weight_1dof(epoch: int, n_epochs: int, anneal_ratio: float = 0.3) -> float:
    """Epoch 0-based: 1_DOF/Scripts/03_train_vae.py:26-30."""
    pivot = int(n_epochs * anneal_ratio)
    return float(1.0 / (1.0 + math.exp(-5.0 * (epoch - pivot) / max(pivot, 1))))


def kl_weight_openlab(epoch: int, n_epochs: int, anneal_ratio: float = 0.30) -> float:
    """20250506_openLAB_tests/Codes/04_train_vae.py:129-131."""
    x = (epoch - (n_epochs * anneal_ratio)) / max(n_epochs * anneal_ratio, 1e-12)
    return float(1.0 / (1.0 + math.exp(-x * 5.0)))
