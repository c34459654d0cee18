"""Ad-hoc: train-step throughput vs per-GPU batch, to show how the quad-tile design scales past B = 512.
    python tools/batch_scaling.py [bf16|split]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "hybrid-vae-cnn-for-shm_amd")):
    sys.path.insert(0, p)
import torch
from Scripts.Models.temporal_vae import TemporalVAE
from shm_hip.trainer import VAETrainer
PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for B in (256, 512, 1024, 2048):
    torch.manual_seed(0)
    m = TemporalVAE(4, 16, 128, 2, 0.3, precision=PREC).cuda().train()
    tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
    x = torch.randn(B, 256, 4, device="cuda")
    for _ in range(3): tr.step(x, 0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
    for _ in range(n): tr.step(x, 0.5)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    tr.engine.check_status()
    print(f"{PREC} B={B}: {dt*1e3:.3f} ms/step = {B/dt:,.0f} windows/s", flush=True)
