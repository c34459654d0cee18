"""fp32 parity path vs bf16 throughput path trained on the SAME device-generated S2 data (4-DOF Newmark displacement
windows, T = 256, D = 4), same seeds, same schedule: per-epoch train / validation reconstruction error of both, written to
profiles/r01_s2_convergence_seed*.json (committed copies).    python tools/s2_convergence.py [--runs 160] [--epochs 60] [--seed 42]"""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hybrid-vae-cnn-for-shm_amd"))
from shm_hip import simulate, workflows   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=160)
    ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--stride", type=int, default=4)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out/s2_convergence.json"))
    a = ap.parse_args()
    Ztr, Zva, mean, std = simulate.s2_windows(a.runs, 256, a.stride)
    print(f"S2: {Ztr.shape[0]} train / {Zva.shape[0]} val windows of 256 x 4 from {a.runs} runs; mean {mean.tolist()} std {std.tolist()}")
    ztr, zva = Ztr.cpu().numpy(), Zva.cpu().numpy()
    res = {"runs": a.runs, "epochs": a.epochs, "seed": a.seed, "train_windows": int(Ztr.shape[0]), "val_windows": int(Zva.shape[0])}
    for prec in ("fp32", "bf16"):
        with tempfile.TemporaryDirectory() as td:
            t0 = time.perf_counter()
            _, hist = workflows.train_vae(ztr, zva, td, epochs=a.epochs, batch_size=512, precision=prec, seed=a.seed, log=lambda *_: None)
            dt = time.perf_counter() - t0
        res[prec] = {"train_recon": hist["train_recon"], "val_recon": hist["val_recon"], "val_total": hist["val_total"], "seconds": dt}
        print(f"{prec}: {dt:6.1f} s  train recon {hist['train_recon'][0]:.4f} -> {hist['train_recon'][-1]:.4f}   "
              f"val recon {hist['val_recon'][0]:.4f} -> {hist['val_recon'][-1]:.4f}")
    res["final_val_recon_ratio_bf16_over_fp32"] = res["bf16"]["val_recon"][-1] / res["fp32"]["val_recon"][-1]
    print("final val recon bf16 / fp32 =", res["final_val_recon_ratio_bf16_over_fp32"])
    Path(a.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
