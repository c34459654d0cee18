"""The only numbers the reference itself holds for this path, reproduced on its own windows (stages 00 -> 03 -> 04 -> 05).

Inputs: tests/golden/reference_4dof_inputs.npz = the numeric contents (fp32) of the 20 CSV runs the reference ships under
4DOF/Data/raw, written by tests/golden/make_reference_inputs.py (inputs only).  Here the reference's directory layout is rebuilt
from it, our stage code runs on the GPU with the reference's hyper-parameters and seed 42, and the results are compared with
  4DOF/Data/processed/stage1_vae_train_meta.json   best_val_total 0.659690 @ epoch 7      (03_train_vae.py:254-331)
  4DOF/Data/processed/vae_threshold.json           threshold 1.281404, normal-val mean 0.6513, n = 2010 / 804 / 1206 (04:283)
  4DOF/Data/processed/stage2_cnn_train_meta.json   best_val_loss 0.097418                  (05_train_cnn.py:266-313)

What can and cannot be equal.  Window counts and split totals are integer bookkeeping: exact.  The loss values are outcomes of a
stochastic run on the author's machine (unknown torch / GPU): the weight initialisation (torch CPU generator, seed 42, same module
creation order) and the epoch shuffles (DataLoader generator seed 42 -> torch.randperm) are reproduced here, but the
reparameterisation noise and the dropout masks come from device RNG streams that no other implementation can replay
(SURVEY.md §7), and every later step depends on them.  So the floats are checked against BANDS, stated with their reason where
they are asserted, and the measured values go to gpurun_out/parity_errors.jsonl.
"""
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN, record_parity

pytestmark = pytest.mark.gpu
T, D = 100, 12


@pytest.fixture(scope="module")
def ref_root(tmp_path_factory):
    d = np.load(GOLDEN / "reference_4dof_inputs.npz", allow_pickle=False)
    root = tmp_path_factory.mktemp("ref4dof")
    header = str(d["header"])
    for name, tab in zip(d["names"], d["tables"]):
        fp = root / str(name)
        fp.parent.mkdir(parents=True, exist_ok=True)
        np.savetxt(fp, tab, delimiter=",", header=header, comments="", fmt="%.9g")      # 9 significant digits round-trip fp32 exactly
    return root, json.loads(str(d["held"]))


@pytest.mark.parametrize("precision", ["split", "fp32"])
def test_reference_artifacts_on_reference_windows(ref_root, precision):
    from shm_hip import dataprep, workflows
    root, held = ref_root
    # ---- stage 00: run_splits.json totals (00_make_run_splits.py:126-141) — exact
    splits = dataprep.make_run_splits(root, seq_len=T)
    assert splits["totals"] == held["run_splits_totals"]
    # ---- stage 03 (03_train_vae.py:177-354)
    nf = splits["normal"]["files"]
    Wtr = dataprep.fraction_windows(nf, (0.0, 0.4), T, D, 1, root)
    Wva = dataprep.fraction_windows(nf, (0.4, 0.7), T, D, 1, root)
    assert Wtr.shape == (3010, T, D) and Wva.shape == (2010, T, D)                       # 10 files x 301 / 201 windows
    mean, std = dataprep.mean_std_from_windows(Wtr)
    (root / "Data/processed").mkdir(parents=True, exist_ok=True)
    np.savez(root / "Data/processed/normal_stats.npz", mean=mean, std=std)
    Ztr, Zva = dataprep.normalize_windows(Wtr, mean, std), dataprep.normalize_windows(Wva, mean, std)
    hv = held["vae_train"]
    vae, hist = workflows.train_vae(Ztr, Zva, root / f"models_{precision}", epochs=hv["epochs"], batch_size=hv["batch_size"], latent_dim=hv["latent_dim"],
                                    hidden_dim=hv["hidden_dim"], num_layers=hv["num_layers"], dropout=hv["dropout"],
                                    kl_warmup_ratio=hv["kl_warmup_ratio"], seed=hv["seed"], precision=precision, log=lambda s: None)
    meta = json.loads((root / f"models_{precision}/stage1_vae_train_meta.json").read_text())
    # ---- stage 04 (04_vae_thresholding.py:232-311): P99 of the normal-val scores with the best-val weights
    sd = torch.load(root / f"models_{precision}/temporal_vae_state_dict.pt", map_location="cuda", weights_only=True)
    vae.load_state_dict(sd)
    torch.manual_seed(hv["seed"])
    thr, scores = workflows.vae_threshold(vae, Zva, root / "Data/processed")
    counts = {}
    fault_scores = {}
    for key in ("sensor_fault", "structural_fault"):
        W = dataprep.fraction_windows(splits[key]["files"], (0.4, 0.7), T, D, 1, root)
        counts[key] = W.shape[0]
        from shm_hip.pipeline import score_windows
        fault_scores[key] = score_windows(vae, torch.from_numpy(dataprep.normalize_windows(W, mean, std)).cuda()).cpu().numpy()
    ht = held["threshold"]
    assert (scores.size, counts["sensor_fault"], counts["structural_fault"]) == (ht["n_val_windows_normal"], ht["n_val_windows_sensor"], ht["n_val_windows_structural"])
    # ---- stage 05 (05_train_cnn.py:174-313)
    hc = held["cnn_train"]
    data = {}
    for split in ("train", "val"):
        Ws = dataprep.windows_by_index(splits["sensor_fault"]["files"], splits["sensor_fault"]["window_indices"], split, T, D, root)
        Wt = dataprep.windows_by_index(splits["structural_fault"]["files"], splits["structural_fault"]["window_indices"], split, T, D, root)
        assert Ws.shape[0] == held["run_splits_totals"]["sensor_fault"][split] and Wt.shape[0] == held["run_splits_totals"]["structural_fault"][split]
        Z = dataprep.normalize_windows(np.concatenate([Ws, Wt], 0), mean, std)
        y = np.concatenate([np.zeros(Ws.shape[0], np.int64), np.ones(Wt.shape[0], np.int64)])
        perm = np.random.default_rng(42).permutation(Z.shape[0])                           # 05:238-242
        data[split] = (workflows.build_cnn_inputs(vae, Z[perm]), y[perm])
    cnn, chist = workflows.train_cnn(data["train"][0], data["train"][1], data["val"][0], data["val"][1], root / f"models_{precision}", epochs=hc["epochs"],
                                     batch_size=hc["batch_size"], lr=hc["learning_rate"], weight_decay=hc["weight_decay"], dropout=hc["dropout"],
                                     seed=hc["seed"], log=lambda s: None)
    cmeta = json.loads((root / f"models_{precision}/stage2_cnn_train_meta.json").read_text())

    got = dict(best_val_total=meta["best_val_total"], best_epoch=meta["best_epoch"], threshold=thr, normal_val_mean=float(scores.mean()),
               normal_val_std=float(scores.std()), sensor_val_mean=float(fault_scores["sensor_fault"].mean()),
               structural_val_mean=float(fault_scores["structural_fault"].mean()), cnn_best_val_loss=cmeta["best_val_loss"],
               collapsed_epochs=len(hist.get("collapsed_epochs", [])))
    ref = dict(best_val_total=hv["best_val_total"], best_epoch=hv["best_epoch"], threshold=ht["threshold"],
               normal_val_mean=ht["score_summary"]["normal_val"]["mean"], normal_val_std=ht["score_summary"]["normal_val"]["std"],
               sensor_val_mean=ht["score_summary"]["sensor_val"]["mean"], structural_val_mean=ht["score_summary"]["structural_val"]["mean"],
               cnn_best_val_loss=hc["best_val_loss"])
    print("reference artefacts", precision, json.dumps(got), "held:", json.dumps(ref))
    record_parity("reference_artifacts", precision, {**got, **{f"held_{k}": v for k, v in ref.items()}})

    # Bands.  Measured in round 2 (profiles/r02_parity_errors.json; split / fp32): best-val total 0.65846 / 0.65855 at epoch 8 / 7 against the
    # held 0.65969 at epoch 7; normal-val mean 0.65091 / 0.65082 against 0.65131; P99 threshold 1.2908 / 1.2873 against 1.2814; sensor /
    # structural means 33.2 / 38.3 against 33.5 / 38.3; CNN best-val CE 0.1136 / 0.1139 against 0.0974.  The initialisation and the epoch
    # shuffles ARE the reference's (torch CPU generator, seed 42); what differs is the eps / dropout noise, whose effect on these
    # aggregates turns out to be under one percent — the bands leave room for a different noise realisation, not for a wrong model:
    # best-val total (= recon + kl_w * KL at an early epoch, mostly the validation reconstruction error): +-3 %.
    assert abs(got["best_val_total"] - ref["best_val_total"]) < 0.03 * ref["best_val_total"], (got, ref)
    assert 5 <= got["best_epoch"] <= 10, got                                               # held: 7; the minimum sits where the KL ramp starts to bite
    # the normal-val score mean is an average over 2010 windows: +-3 %; the P99 threshold is set by the ~20 largest of them under ONE
    # eps draw (noisier): +-6 %.
    assert abs(got["normal_val_mean"] - ref["normal_val_mean"]) < 0.03 * ref["normal_val_mean"], (got, ref)
    assert abs(got["normal_val_std"] - ref["normal_val_std"]) < 0.06 * ref["normal_val_std"], (got, ref)
    assert abs(got["threshold"] - ref["threshold"]) < 0.06 * ref["threshold"], (got, ref)
    # faults sit one to two orders of magnitude above the gate, as in the held summary (33.5 / 38.3 vs 0.65): +-10 % on their means
    assert abs(got["sensor_val_mean"] - ref["sensor_val_mean"]) < 0.10 * ref["sensor_val_mean"], (got, ref)
    assert abs(got["structural_val_mean"] - ref["structural_val_mean"]) < 0.10 * ref["structural_val_mean"], (got, ref)
    # CNN best validation cross-entropy: its inputs carry the VAE's residual channel (one eps draw per window) and its own dropout /
    # shuffle streams are torch's global RNG after a different call history, so "same order" is the honest claim: within a factor of
    # two of the held 0.0974 (ln 2 = 0.693 is an untrained classifier)
    assert 0.5 * ref["cnn_best_val_loss"] < got["cnn_best_val_loss"] < 2.0 * ref["cnn_best_val_loss"], (got, ref)
    assert got["collapsed_epochs"] == 0
