"""Ad-hoc: host-side enqueue time of one fused train step vs its GPU time (is the step GPU-bound?)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "hybrid-vae-cnn-for-shm_amd")):
    sys.path.insert(0, p)
import torch
from Scripts.Models.temporal_vae import TemporalVAE
from shm_hip.trainer import VAETrainer
torch.manual_seed(0)
m = TemporalVAE(4, 16, 128, 2, 0.3, precision="bf16").cuda().train()
tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
x = torch.randn(512, 256, 4, device="cuda")
for _ in range(5): tr.step(x, 0.5)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n): tr.step(x, 0.5)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"enqueue {t_enq/n*1e3:.3f} ms/step ; complete {t_all/n*1e3:.3f} ms/step (GPU-bound if enqueue < complete)")
