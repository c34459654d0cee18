import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/hybrid-vae-cnn-for-shm_amd')
import torch, numpy as np
from Scripts.Models.temporal_vae import TemporalVAE
from shm_hip.trainer import VAETrainer
torch.manual_seed(42)
m = TemporalVAE(4,16,128,2,0.3,precision="bf16").cuda().train()
tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
x = torch.randn(16384,256,4, device="cuda")
evs=[torch.cuda.Event(enable_timing=True) for _ in range(61)]
torch.cuda.synchronize()
evs[0].record()
for i in range(60):
    tr.step(x[(i%32)*512:(i%32+1)*512], 0.3)
    evs[i+1].record()
torch.cuda.synchronize()
ts=[evs[i].elapsed_time(evs[i+1]) for i in range(60)]
print(' '.join('%.3f'%t for t in ts))
