"""One-off extended stress of the in-launch hand-offs of the split (default) or bf16 path (forward pair, top-layer backward + dx) under uneven background load:
N repetitions of a pipelined train step at three shapes while a second stream streams copies of varying size through HBM / L2, each
compared BIT FOR BIT with the one-launch-per-layer result.  (The GPU suite runs 10 repetitions of one shape; this is the long version.)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
for p in (str(ROOT), str(ROOT / "hybrid-vae-cnn-for-shm_amd")):
    sys.path.insert(0, p)
import torch
from Scripts.Models.temporal_vae import TemporalVAE
from shm_hip._capi import lib
from shm_hip.trainer import VAETrainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
PREC = sys.argv[2] if len(sys.argv) > 2 else "split"          # "bf16": the same stress of the bf16 pair launches
side = torch.cuda.Stream()
a = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
b = torch.empty_like(a)
bad = 0
for (B, T, H) in ((512, 256, 128), (300, 64, 128), (128, 100, 64)):
    gen = torch.Generator().manual_seed(B + T)
    x = torch.randn(B, T, 4, generator=gen).cuda()
    eps = torch.randn(B, 16, generator=gen).cuda()

    def one(on):
        lib.shm_set_pair_launch(on)
        try:
            torch.manual_seed(21)
            m = TemporalVAE(4, 16, H, 2, 0.3, precision=PREC).cuda().train()
            tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
            st = tr.step(x, 0.3, eps=eps, drop_seed=5).clone()
            tr.engine.check_status()
            return st, tr.grads.clone()
        finally:
            lib.shm_set_pair_launch(1)

    ref = one(0)
    for rep in range(N):
        with torch.cuda.stream(side):
            for _ in range(rep % 5):
                n = (1 + rep % 7) * 8 * 1024 * 1024
                b[:n].copy_(a[:n], non_blocking=True)
        got = one(1)
        side.synchronize()
        for u, v in zip(ref, got):
            if not torch.equal(u, v):
                bad += 1
                print("MISMATCH", (B, T, H), rep, float((u - v).abs().max()))
    print((B, T, H), "done", N, "reps")
print("mismatches:", bad)
sys.exit(1 if bad else 0)
