"""End-to-end stages 03 -> 04 -> 05 -> 06 on a synthetic 4DOF-like data root (GPU).

The reference's own data (CSV runs) cannot travel; the synthetic root has the same layout (Data/processed/run_splits.json,
mode window_level_per_file) and three signal families: normal (sum of sines), sensor fault (additive bias/spikes on some
channels), structural fault (frequency shift).  Checks are behavioural: loss falls, the P99 gate fires on faults, the CNN
separates the two fault kinds, artifacts exist with the reference's names.
"""
import json
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG

pytestmark = pytest.mark.gpu
T, D = 100, 12


def _signal(n, rng, kind):
    """Slow drift (what a window-level latent can encode — with pure fast sines the reference model itself stays at
    recon = 1.000, checked with oracle/torch_cpu.py) + fast modal content + noise."""
    t = np.arange(n)[:, None] * 0.02
    f = np.linspace(1.0, 4.0, D)[None, :] * (1.35 if kind == "structural" else 1.0)
    amp = 1.5 if kind == "structural" else 0.3
    x = amp * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28, (1, D))) + 0.05 * rng.standard_normal((n, D))
    x += 1.5 * np.sin(2 * np.pi * t / 7.0 + rng.uniform(0, 6.28, (1, D)))
    if kind == "sensor":
        x[:, :4] += 3.0 + 0.5 * rng.standard_normal((n, 4))
    return x.astype(np.float32)


@pytest.fixture(scope="module")
def root(tmp_path_factory):
    root = tmp_path_factory.mktemp("dof4")
    rng = np.random.default_rng(7)
    (root / "Data/raw").mkdir(parents=True); (root / "Data/processed").mkdir(parents=True)
    splits = {"mode": "window_level_per_file"}
    for key, kind in (("normal", "normal"), ("sensor_fault", "sensor"), ("structural_fault", "structural")):
        files, widx = [], {}
        for k in range(2):
            n = 900
            rel = f"Data/raw/{kind}_{k}.csv"
            np.savetxt(root / rel, _signal(n, rng, kind), delimiter=",", header=",".join(f"c{i}" for i in range(D)), comments="")
            files.append(rel)
            starts = rng.permutation(n - T + 1)
            widx[rel] = {"train": sorted(starts[:200].tolist()), "val": sorted(starts[200:280].tolist()), "test": sorted(starts[280:360].tolist())}
        splits[key] = {"files": files, "window_indices": widx}
    (root / "Data/processed/run_splits.json").write_text(json.dumps(splits))
    return root


def test_stages_end_to_end(root):
    from shm_hip import dataprep, workflows
    splits = json.loads((root / "Data/processed/run_splits.json").read_text())
    nf = splits["normal"]["files"]
    Wtr = dataprep.fraction_windows(nf, (0.0, 0.4), T, D, 1, root)
    Wva = dataprep.fraction_windows(nf, (0.4, 0.7), T, D, 1, root)
    assert Wtr.shape == (2 * (360 - T + 1), T, D)
    mean, std = dataprep.mean_std_from_windows(Wtr)
    np.savez(root / "Data/processed/normal_stats.npz", mean=mean, std=std)
    Ztr, Zva = dataprep.normalize_windows(Wtr, mean, std), dataprep.normalize_windows(Wva, mean, std)

    logs = []
    vae, hist = workflows.train_vae(Ztr, Zva, root / "models", epochs=6, batch_size=64, log=logs.append)
    assert len(hist["epoch"]) == 6 and len(logs) == 12
    from shm_hip.schedules import kl_weight_4dof
    assert hist["kl_w"] == [pytest.approx(kl_weight_4dof(e, 6, 0.30)) for e in range(1, 7)]     # schedule pinned in test_host_logic_cpu
    assert hist["train_recon"][-1] < 0.75 * hist["train_recon"][0]
    assert np.isfinite(hist["val_total"]).all()
    meta = json.loads((root / "models/stage1_vae_train_meta.json").read_text())
    assert meta["best_epoch"] >= 1 and (root / "models/temporal_vae_state_dict.pt").exists()
    # the saved state dict has the reference's key names and reloads into a fresh module
    sd = torch.load(root / "models/temporal_vae_state_dict.pt", map_location="cuda", weights_only=True)
    assert "encoder_lstm.weight_ih_l0" in sd and "output_layer.bias" in sd
    from Scripts.Models.temporal_vae import TemporalVAE
    vae2 = TemporalVAE(D, 16, 128, 2, 0.3).to("cuda"); vae2.load_state_dict(sd)

    thr, scores = workflows.vae_threshold(vae2, Zva, root / "Data/processed")
    tj = json.loads((root / "Data/processed/vae_threshold.json").read_text())
    assert tj["threshold"] == pytest.approx(thr) and thr == pytest.approx(float(np.percentile(scores, 99.0)))

    data = {}
    for split in ("train", "val"):
        Ws = dataprep.windows_by_index(splits["sensor_fault"]["files"], splits["sensor_fault"]["window_indices"], split, T, D, root)
        Wt = dataprep.windows_by_index(splits["structural_fault"]["files"], splits["structural_fault"]["window_indices"], split, T, D, root)
        Z = dataprep.normalize_windows(np.concatenate([Ws, Wt]), mean, std)
        y = np.concatenate([np.zeros(len(Ws), np.int64), np.ones(len(Wt), np.int64)])
        data[split] = (workflows.build_cnn_inputs(vae2, Z), y)
    assert data["train"][0].shape == (800, 2, T, D)
    cnn, chist = workflows.train_cnn(*data["train"], *data["val"], root / "models", epochs=8, batch_size=100, lr=1e-3, log=logs.append)
    assert chist[-1][1] < chist[0][1] and (root / "models/cnn_state_dict.pt").exists()

    groups = {name: dataprep.normalize_windows(dataprep.fraction_windows(splits[key]["files"], (0.7, 1.0), T, D, 1, root), mean, std)
              for name, key in (("normal", "normal"), ("sensor", "sensor_fault"), ("structural", "structural_fault"))}
    conf, gate = workflows.test_pipeline(vae2, cnn, groups, thr)
    assert conf.sum(1).tolist() == [g.shape[0] for g in groups.values()]
    assert gate["sensor"]["anom_rate"] > 0.9 and gate["structural"]["anom_rate"] > 0.9 and gate["normal"]["anom_rate"] < 0.6   # 6-epoch VAE, the test fraction is an unseen phase of the drift
    assert conf[1, 1] > 0.8 * conf[1].sum() and conf[2, 2] > 0.8 * conf[2].sum()


def test_numbered_scripts_run(root):
    """The CLI counterparts (03 -> 04 -> 06 needs 05's model: run all four, tiny epoch counts) exit 0 and leave the artifacts."""
    for mod, extra in (("Scripts.03_train_vae", ["--epochs", "2"]), ("Scripts.04_vae_thresholding", []),
                       ("Scripts.05_train_cnn", ["--epochs", "2"]), ("Scripts.06_test_full_pipeline", [])):
        r = subprocess.run([sys.executable, "-m", mod, "--root", str(root), *extra], cwd=str(PKG), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
    assert (root / "Data/processed/vae_mean.npy").exists() and (root / "Data/processed/vae_threshold.json").exists()
    assert '"confusion"' in r.stdout


def test_device_windowing_is_bit_identical_to_host_prep():
    """shm_window_gather vs dataprep.make_windows + normalize_windows (numpy): same fp32 arithmetic, NaN / +-inf -> 0, arbitrary starts."""
    from shm_hip import dataprep
    rng = np.random.default_rng(11)
    X = rng.standard_normal((700, D)).astype(np.float32) * 3.0 + 1.0
    X[5, 2] = np.nan; X[77, 0] = np.inf; X[300, 11] = -np.inf
    W = dataprep.make_windows(X, T, 1)
    mean, std = dataprep.mean_std_from_windows(np.nan_to_num(W))
    std[3] = 1e-6                                                  # the std == 0 replacement value of the reference
    ref = dataprep.normalize_windows(W, mean, std)
    dev = torch.from_numpy(X).cuda()
    got = dataprep.device_windows(dev, np.arange(W.shape[0]), T, mean, std)
    assert got.shape == ref.shape and np.array_equal(got.cpu().numpy(), ref)
    sel = rng.permutation(W.shape[0])[:37]
    got = dataprep.device_windows(dev, sel, T, mean, std)
    assert np.array_equal(got.cpu().numpy(), ref[sel])
    raw = dataprep.device_windows(dev, sel[:5], T)                 # no normalisation: NaN/inf still scrubbed
    assert np.array_equal(raw.cpu().numpy(), np.nan_to_num(W[sel[:5]], nan=0.0, posinf=0.0, neginf=0.0))
    # openLAB variant: clip before the NaN replacement (+-inf -> +-clip)
    mu, sd = dataprep.fit_mean_std_openlab(np.where(np.isfinite(W[:50]), W[:50], np.nan))
    refo = dataprep.standardize_openlab(W[sel], mu, sd, 2.5)
    goto = dataprep.device_windows(dev, sel, T, mu, sd, clip_z=2.5)
    assert np.array_equal(goto.cpu().numpy(), refo) and refo.max() == 2.5
    assert dataprep.device_windows(dev, [], T, mean, std).shape == (0, T, D)
    with pytest.raises(ValueError):
        dataprep.device_windows(dev, [700 - T + 1], T)


def test_train_vae_1dof_loop_semantics(tmp_path):
    """1_DOF trainer (1_DOF/Scripts/03_train_vae.py:262-335): no LayerNorm, no weight decay, no clipping, 0-based KL epoch, batch-mean
    epoch averages, final weights saved; and the 1_DOF evaluation helpers on its reconstruction."""
    from shm_hip import dataprep, workflows
    from shm_hip.schedules import kl_weight_1dof
    rng = np.random.default_rng(2)
    t = np.arange(400)[:, None] * 0.05
    series = (np.sin(t * np.linspace(0.5, 1.5, 12)[None, :]) + 0.05 * rng.standard_normal((400, 12))).astype(np.float32)
    W = dataprep.make_windows(series, 80, 1)
    logs = []
    vae, hist = workflows.train_vae_1dof(W, tmp_path, n_epochs=10, batch_size=64, log=logs.append)
    assert "layer_norm.weight" not in vae.state_dict() and (tmp_path / "temporal_vae_state_dict.pt").exists()
    assert hist["kl_weight"] == [pytest.approx(kl_weight_1dof(e, 10, 0.3)) for e in range(10)] and len(logs) == 1
    assert hist["loss_recon"][-1] < hist["loss_recon"][0]
    # stitched reconstruction of the overlapping windows and its per-segment RMSE (datasets.py:38-71)
    vae.eval()
    with torch.no_grad():
        rec, _, _ = vae(torch.from_numpy(W).cuda())
    stitched = dataprep.stitch_windows(rec.cpu().numpy(), series.shape[0], 1)
    rm = dataprep.segment_rmse(series, stitched, 100)
    assert stitched.shape == series.shape and rm.shape == (4,) and np.isfinite(rm).all()
