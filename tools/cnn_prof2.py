import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "hybrid-vae-cnn-for-shm_amd")):
    sys.path.insert(0, p)
import torch
from Scripts.Models.cnn_model import CNN
from shm_hip.trainer import CNNTrainer
B, T, D = 512, 256, 4
m = CNN(2, 2, 0.5, seq_len=T, num_features=D).cuda().train()
tr = CNNTrainer(m, lr=1e-4, weight_decay=5e-5)
x = torch.randn(B, 2, T, D, device="cuda"); y = torch.randint(0, 2, (B,), device="cuda")
for _ in range(12): tr.step(x, y)
m.eval()
with torch.no_grad():
    for _ in range(12): m(x)
torch.cuda.synchronize()
