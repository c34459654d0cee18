"""The data-parallel HIP trainers with world > 1, on the one GPU a test box has (reference step being sharded:
4DOF/Scripts/03_train_vae.py:260-271 and 05_train_cnn.py:270-276).

Two fresh child processes per test (`python -m torch.distributed.run --nproc-per-node 2`, started before they touch the GPU), both
on cuda:0 over gloo: RCCL refuses two ranks on one device, and what needs testing is the product's code around the collective —
start-up broadcast + re-pack, all-reduce of the flat gradient bucket feeding the raw-pointer Adam with 1/world folded in, the
bench's N > 1 control flow — not the transport.  No scaling numbers come out of this.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, split_prefix

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(script_args, env_extra=None, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), *script_args]
    r = subprocess.run(cmd, cwd=str(ROOT), capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r


def _single_process_vae(precision, steps):
    """The same steps on the full (even-sized) batch in THIS process: the reference a 2-rank run has to reproduce."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip.trainer import VAETrainer
    g = load_golden("vae_train_mini")
    cfg = g["cfg"]
    m = TemporalVAE(cfg["input_dim"], cfg["latent_dim"], cfg["hidden_dim"], cfg["num_layers"], 0.0, layer_norm=cfg["layer_norm"], precision=precision)
    m.load_state_dict({k: torch.tensor(v) for k, v in split_prefix(g, "p0.").items()}, strict=True)
    m = m.cuda().train()
    tr = VAETrainer(m, lr=cfg["lr"], weight_decay=cfg["wd"], max_norm=cfg["clip"], seed=42)
    grads0, stats = None, []
    for s in range(steps):
        x, eps = g[f"x{s}"], g[f"eps{s}"]
        n = (x.shape[0] // 2) * 2
        stats.append(tr.step(torch.tensor(x[:n]).cuda(), cfg["kl_w"], eps=torch.tensor(eps[:n]).cuda()).cpu().numpy())
        if s == 0:
            grads0 = tr.grads.cpu().numpy().copy()
    return tr.flat.cpu().numpy(), grads0, stats


@pytest.mark.parametrize("precision", ["fp32", "split", "bf16"])
def test_two_rank_vae_trainer_matches_single_rank_full_batch(tmp_path, precision):
    steps = 2
    _launch([str(ROOT / "tests" / "multirank_worker.py"), "--mode", "vae", "--precision", precision, "--steps", str(steps), "--out", str(tmp_path)])
    p0, p1 = np.load(tmp_path / "vae_params_r0.npy"), np.load(tmp_path / "vae_params_r1.npy")
    g0, g1 = np.load(tmp_path / "vae_grads_r0.npy"), np.load(tmp_path / "vae_grads_r1.npy")
    # (i) every rank applied the bit-identical update: same reduced bucket, same clip coefficient, same Adam
    assert np.array_equal(g0, g1), "reduced gradient buckets differ between ranks"
    assert np.array_equal(p0, p1), "parameters differ between ranks after the data-parallel steps"
    # (ii) and it is the single-process step on the full batch: sum of the rank gradients * 1/world == gradient of the global mean
    pref, gref, sref = _single_process_vae(precision, steps)
    gdp = 0.5 * g0
    # gradients carry the precise check (fp32 / split: summation order only); the parameter bar is the golden tests' (Adam's first steps
    # move an element with |g| ~ eps by up to lr whatever the size of the gradient difference)
    tol_g = 1e-5 if precision != "bf16" else 3e-2
    worst_g = float(np.abs(gdp - gref).max() / np.abs(gref).max())
    worst_p = float(np.abs(p0 - pref).max())
    print(precision, "dp2 vs single: grad rel-of-max", worst_g, "param max abs after", steps, "steps", worst_p)
    assert worst_g < tol_g and worst_p < 0.2 * 1e-3 * steps
    # the loss a rank reports is its LOCAL mean; the mean over ranks is the global loss
    for s in range(steps):
        l0, l1 = np.load(tmp_path / f"vae_stats_r0_s{s}.npy"), np.load(tmp_path / f"vae_stats_r1_s{s}.npy")
        assert abs(0.5 * (l0[0] + l1[0]) - sref[s][0]) < (1e-5 if precision != "bf16" else 1e-3) * max(1.0, abs(sref[s][0]))
        assert l0[3] == l1[3]                                    # gradient norm of the reduced bucket: identical on both ranks
        assert abs(l0[3] - sref[s][3]) < (1e-4 if precision != "bf16" else 1e-2) * max(1.0, sref[s][3])


def test_two_rank_cnn_trainer(tmp_path):
    """CNN under data parallelism: BatchNorm batch statistics local to a rank, gradients reduced, rank 0's running
    statistics are the model's.  Equality target: the average of the two half-batch gradients computed in this process."""
    from Scripts.Models.cnn_model import CNN
    from shm_hip.trainer import CNNTrainer
    steps = 2
    _launch([str(ROOT / "tests" / "multirank_worker.py"), "--mode", "cnn", "--steps", str(steps), "--out", str(tmp_path)])
    L = lambda n: np.load(tmp_path / n)
    # start-up broadcast: both ranks begin from rank 0's initialisation although they were seeded differently
    assert np.array_equal(L("cnn_params0_r0.npy"), L("cnn_params0_r1.npy"))
    assert np.array_equal(L("cnn_grads_r0.npy"), L("cnn_grads_r1.npy"))
    assert np.array_equal(L("cnn_params_r0.npy"), L("cnn_params_r1.npy"))
    # running statistics drift per rank while training and are identical (rank 0's) after the sync in evaluate()
    assert not np.array_equal(L("cnn_running_local_r0.npy"), L("cnn_running_local_r1.npy"))
    assert np.array_equal(L("cnn_running_synced_r0.npy"), L("cnn_running_synced_r1.npy"))
    assert np.array_equal(L("cnn_running_synced_r0.npy"), L("cnn_running_local_r0.npy"))
    assert np.array_equal(L("cnn_val_r0.npy"), L("cnn_val_r1.npy"))
    # reference: each half batch through a single-process trainer that starts from the same parameters, gradients averaged
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(steps, 24, 2, 32, 8, generator=gen)
    y = torch.randint(0, 2, (steps, 24), generator=gen)
    halves = []
    for r in range(2):
        torch.manual_seed(1234)
        m = CNN(2, 2, 0.0, seq_len=32, num_features=8).cuda().train()
        tr = CNNTrainer(m, lr=1e-3, weight_decay=5e-5, seed=42)
        assert np.array_equal(tr.flat.cpu().numpy(), L("cnn_params0_r0.npy"))
        idx = list(range(r, 24, 2))
        tr.step(x[0, idx].cuda(), y[0, idx].cuda())
        halves.append(tr.grads.cpu().numpy().copy())
    gref = 0.5 * (halves[0] + halves[1])
    gdp = 0.5 * L("cnn_grads_r0.npy")
    worst = float(np.abs(gdp - gref).max() / np.abs(gref).max())
    print("cnn dp2 reduced gradient vs mean of the half-batch gradients: rel-of-max", worst)
    assert worst < 1e-5


def test_bench_two_ranks_keeps_the_json_contract(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run), rehearsed on one device: barrier + MAX-reduce of the timed
    region, profile steps on every rank, rank 0 prints ONE JSON line."""
    r = _launch([str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                env_extra={"SHM_BENCH_SINGLE_DEVICE": "1", "SHM_BENCH_BACKEND": "gloo"}, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "bench_2rank_rehearsal.json").write_text(lines[0] + "\n")
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 1024 and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert abs(d["value"] - 1024 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-2 * d["value"]
    assert "cpu_baseline" not in d and d["roofline"] is not None
