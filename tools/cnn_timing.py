"""Ad-hoc timing of the CNN train step / eval forward (not a bench.py line; numbers quoted in DESIGN.md)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "hybrid-vae-cnn-for-shm_amd")):
    sys.path.insert(0, p)
import torch
from Scripts.Models.cnn_model import CNN
from shm_hip.trainer import CNNTrainer

for (B, T, D) in ((100, 100, 12), (512, 256, 4), (2048, 256, 4)):
    torch.manual_seed(0)
    m = CNN(2, 2, 0.5, seq_len=T, num_features=D).cuda().train()
    tr = CNNTrainer(m, lr=1e-4, weight_decay=5e-5)
    x = torch.randn(B, 2, T, D, device="cuda"); y = torch.randint(0, 2, (B,), device="cuda")
    for _ in range(3): tr.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
    for _ in range(n): tr.step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    m.eval()
    with torch.no_grad():
        for _ in range(3): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): m(x)
        torch.cuda.synchronize(); de = (time.perf_counter() - t0) / n
    print(f"CNN B={B} T={T} D={D}: train {dt*1e3:.3f} ms/step = {B/dt:,.0f} windows/s ; eval fwd {de*1e3:.3f} ms = {B/de:,.0f} windows/s", flush=True)
