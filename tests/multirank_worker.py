"""Worker of tests/test_gpu_multirank.py: one rank of a 2-rank data-parallel job on ONE GPU.

Started by `python -m torch.distributed.run --nproc-per-node 2 tests/multirank_worker.py ...` as fresh child processes (the
launcher runs before anything touches the GPU).  Both ranks use cuda:0 and the gloo backend — RCCL refuses two ranks on one
device — so what is exercised is the product's own data-parallel code: VAETrainer / CNNTrainer with a process group
(parameter broadcast + mark_dirty, sum-all-reduce of the flat gradient bucket followed by the raw-pointer shm_adam_step with
1/world folded in, per-rank eps / dropout seeds), not the transport.  Each rank saves what it ended up with; the test compares.
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "hybrid-vae-cnn-for-shm_amd"), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["vae", "cnn"], required=True)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from conftest import load_golden, split_prefix
    from shm_hip.trainer import CNNTrainer, VAETrainer, shard_indices

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = Path(a.out)

    if a.mode == "vae":
        from Scripts.Models.temporal_vae import TemporalVAE
        g = load_golden("vae_train_mini")
        cfg = g["cfg"]
        m = TemporalVAE(cfg["input_dim"], cfg["latent_dim"], cfg["hidden_dim"], cfg["num_layers"], 0.0,
                        layer_norm=cfg["layer_norm"], precision=a.precision)
        if rank == 0:       # only rank 0 holds the golden weights: the trainer's start-up broadcast has to deliver them
            m.load_state_dict({k: torch.tensor(v) for k, v in split_prefix(g, "p0.").items()}, strict=True)
        m = m.cuda().train()
        tr = VAETrainer(m, lr=cfg["lr"], weight_decay=cfg["wd"], max_norm=cfg["clip"], seed=42)
        assert tr.comm.world == world
        for s in range(a.steps):
            x, eps = g[f"x{s}"], g[f"eps{s}"]
            n = (x.shape[0] // world) * world                    # equal shards: the mean of the rank means is the global mean
            idx = list(shard_indices(n, rank, world))
            stats = tr.step(torch.tensor(x[idx]).cuda(), cfg["kl_w"], eps=torch.tensor(eps[idx]).cuda()).cpu().numpy()
            tr.check(stats)
            np.save(out / f"vae_stats_r{rank}_s{s}.npy", stats)
            if s == 0:
                np.save(out / f"vae_grads_r{rank}.npy", tr.grads.cpu().numpy())     # the reduced SUM over ranks
        np.save(out / f"vae_params_r{rank}.npy", tr.flat.cpu().numpy())
    else:
        from Scripts.Models.cnn_model import CNN
        torch.manual_seed(1234 + rank)                           # different initialisations per rank: the broadcast has to fix that
        m = CNN(2, 2, 0.0, seq_len=32, num_features=8).cuda().train()
        tr = CNNTrainer(m, lr=1e-3, weight_decay=5e-5, seed=42)
        gen = torch.Generator().manual_seed(99)
        x = torch.randn(a.steps, 24, 2, 32, 8, generator=gen)
        y = torch.randint(0, 2, (a.steps, 24), generator=gen)
        np.save(out / f"cnn_params0_r{rank}.npy", tr.flat.cpu().numpy())
        for s in range(a.steps):
            idx = list(shard_indices(24, rank, world))
            st = tr.step(x[s, idx].cuda(), y[s, idx].cuda()).cpu().numpy()
            np.save(out / f"cnn_stats_r{rank}_s{s}.npy", st)
            if s == 0:
                np.save(out / f"cnn_grads_r{rank}.npy", tr.grads.cpu().numpy())
        np.save(out / f"cnn_params_r{rank}.npy", tr.flat.cpu().numpy())
        np.save(out / f"cnn_running_local_r{rank}.npy", tr.engine.running.cpu().numpy())
        val = tr.evaluate(x[0].cuda(), y[0].cuda()).cpu().numpy()          # syncs the running statistics first
        np.save(out / f"cnn_running_synced_r{rank}.npy", tr.engine.running.cpu().numpy())
        np.save(out / f"cnn_val_r{rank}.npy", val)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
