#!/usr/bin/env python3
"""Probe: does running the train step as M independent micro-batches on M streams raise throughput?  (No product change: M separate
models / trainers of B/M windows each, one torch stream each; aggregate windows/s against one trainer at the full batch.)"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "hybrid-vae-cnn-for-shm_amd")):
    sys.path.insert(0, p)
from Scripts.Models.temporal_vae import TemporalVAE  # noqa: E402
from shm_hip.trainer import VAETrainer  # noqa: E402

B, T, D = 512, 256, 4


def run(prec, M, steps=60, warm=10):
    streams = [torch.cuda.Stream() for _ in range(M)]
    trs, xs = [], []
    for m in range(M):
        torch.manual_seed(42)
        with torch.cuda.stream(streams[m]):
            model = TemporalVAE(D, 16, 128, 2, 0.3, precision=prec).cuda().train()
            trs.append(VAETrainer(model))
            xs.append(torch.randn(B // M, T, D, device="cuda"))
    torch.cuda.synchronize()

    def go(n):
        for _ in range(n):
            for m in range(M):
                with torch.cuda.stream(streams[m]):
                    trs[m].step(xs[m], 0.5)
    go(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return B * steps / dt, 1e3 * dt / steps


for prec in ("bf16", "split"):
    for M in (1, 2, 4):
        v, ms = run(prec, M)
        print(f"{prec} M={M}: {v:,.0f} windows/s, {ms:.3f} ms per 512-window step", flush=True)
