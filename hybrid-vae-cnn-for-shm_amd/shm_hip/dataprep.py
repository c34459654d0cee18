"""Host-side data preparation shared by every stage.

Semantics follow reference 4DOF/Scripts/03_train_vae.py:81-117 (duplicated there in 04:73-110, 05:76-115, 06:91-126):
CSV (one header row) -> fp32 rows; fraction slice with int(n*f); stride-1 windows; per-channel mean/std over all window rows
with std == 0 -> 1e-6; z-score with NaN/inf -> 0.  numpy only; the windows go to the device once, as one tensor.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np


def load_csv_numeric(path, num_features: int) -> np.ndarray:
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"Missing CSV: {path}")
    X = np.loadtxt(str(path), delimiter=",", skiprows=1, dtype=np.float64, ndmin=2).astype(np.float32)
    if X.ndim != 2 or X.shape[1] != num_features:
        raise ValueError(f"Bad CSV shape in {path}: {X.shape}")
    return X


def slice_frac(X: np.ndarray, frac_range) -> np.ndarray:
    n = X.shape[0]
    lo, hi = int(n * float(frac_range[0])), int(n * float(frac_range[1]))
    return X[lo:max(hi, lo)]


def make_windows(X: np.ndarray, win_len: int, stride: int = 1) -> np.ndarray:
    n = X.shape[0] - win_len + 1
    if n <= 0:
        return np.zeros((0, win_len, X.shape[1]), dtype=np.float32)
    starts = np.arange(0, n, stride)
    # one strided view + one copy instead of a Python loop over windows
    view = np.lib.stride_tricks.sliding_window_view(X, win_len, axis=0)      # [n, D, win_len]
    return np.ascontiguousarray(view[starts].transpose(0, 2, 1), dtype=np.float32)


def mean_std_from_windows(W: np.ndarray):
    if W.shape[0] == 0:
        raise RuntimeError("No training windows available to compute mean/std.")
    flat = W.reshape(-1, W.shape[-1])
    mean = flat.mean(axis=0).astype(np.float32)
    std = flat.std(axis=0).astype(np.float32)
    std[std == 0] = 1e-6
    return mean, std


def normalize_windows(W: np.ndarray, mean: np.ndarray, std: np.ndarray) -> np.ndarray:
    Z = (W - mean[None, None, :]) / std[None, None, :]
    return np.nan_to_num(Z, nan=0.0, posinf=0.0, neginf=0.0).astype(np.float32)


def fraction_windows(files, frac, win_len: int, num_features: int, stride: int = 1, root=None) -> np.ndarray:
    """All windows of the time-fraction `frac` of every file (03_train_vae.py:165-175)."""
    out = []
    for fp in files:
        p = Path(fp)
        if not p.is_absolute() and root is not None:
            p = (Path(root) / p).resolve()
        W = make_windows(slice_frac(load_csv_numeric(p, num_features), frac), win_len, stride)
        if W.shape[0]:
            out.append(W)
    if not out:
        return np.zeros((0, win_len, num_features), dtype=np.float32)
    return np.concatenate(out, axis=0).astype(np.float32)


def windows_by_index(files, window_indices, split: str, win_len: int, num_features: int, root=None):
    """Windows selected by run_splits.json's per-file start indices (05_train_cnn.py:209-213)."""
    out = []
    for fp in files:
        p = Path(fp)
        if not p.is_absolute() and root is not None:
            p = (Path(root) / p).resolve()
        X = load_csv_numeric(p, num_features)
        idx = window_indices[fp][split]
        if len(idx):
            out.append(np.stack([X[i:i + win_len] for i in idx], axis=0).astype(np.float32))
    if not out:
        return np.zeros((0, win_len, num_features), dtype=np.float32)
    return np.concatenate(out, axis=0)


# ---- openLAB variant (20250506_openLAB_tests/Codes/04_train_vae.py:106-127): NaN-aware statistics over (N, T), std floor 1e-12 -> 1,
# z-scores clipped to +-clip_z BEFORE the NaN/inf -> 0 replacement, optional channel selection.
def fit_mean_std_openlab(X: np.ndarray):
    mu = np.nanmean(X, axis=(0, 1)).astype(np.float32)
    sd = np.nanstd(X, axis=(0, 1)).astype(np.float32)
    sd = np.where(sd < 1e-12, 1.0, sd).astype(np.float32)
    return mu, sd


def standardize_openlab(X: np.ndarray, mu: np.ndarray, sd: np.ndarray, clip_z: float) -> np.ndarray:
    Xn = (X - mu[None, None, :]) / sd[None, None, :]
    Xn = np.clip(Xn, -float(clip_z), float(clip_z))
    return np.nan_to_num(Xn, nan=0.0, posinf=0.0, neginf=0.0).astype(np.float32)


def select_channels(X: np.ndarray, idx):
    idx = list(idx)
    if X.ndim != 3:
        raise ValueError(f"Expected 3D tensor (N,T,C); got shape {X.shape}.")
    if max(idx) >= X.shape[2]:
        raise ValueError(f"CHANNELS_IDX out of range for C={X.shape[2]}.")
    return X[:, :, idx], idx


# ---- on-device windowing: the raw series stays on the device, windows are gathered by libshm_hip.so
def device_windows(series, starts, win_len: int, mean=None, std=None, clip_z: float = 0.0):
    """Z-scored windows [N, win_len, D] (device fp32) of a device series [n_rows, D]: row block starts[n] .. starts[n]+win_len-1 each,
    (x - mean) / std, NaN/inf -> 0 — make_windows + normalize_windows (03_train_vae.py:98-117) without the win_len-fold host copy.
    `starts` may be a list / numpy array (validated here) or an int32 device tensor (validated with one reduction).
    clip_z > 0: the openLAB variant (z clipped to +-clip_z before the NaN replacement, standardize_openlab above)."""
    import ctypes as C

    import torch

    from ._capi import check, lib
    if series.device.type != "cuda" or series.dtype != torch.float32 or series.dim() != 2:
        raise RuntimeError("device_windows: series must be a 2-D fp32 CUDA tensor (there is no CPU path)")
    series = series.contiguous()
    n_rows, D = series.shape
    if not torch.is_tensor(starts):
        starts = torch.as_tensor(np.asarray(starts, dtype=np.int64))
    if starts.numel() and (int(starts.min()) < 0 or int(starts.max()) > n_rows - win_len):
        raise ValueError(f"device_windows: window starts must lie in [0, {n_rows - win_len}]")
    st = starts.to(device=series.device, dtype=torch.int32).contiguous()
    N = int(st.numel())
    out = torch.empty(N, win_len, D, dtype=torch.float32, device=series.device)
    if N == 0:
        return out
    mp = sp = None
    if mean is not None:
        mt = torch.as_tensor(np.asarray(mean, dtype=np.float32)).to(series.device) if not torch.is_tensor(mean) else mean.to(series.device, torch.float32)
        stt = torch.as_tensor(np.asarray(std, dtype=np.float32)).to(series.device) if not torch.is_tensor(std) else std.to(series.device, torch.float32)
        mp, sp = C.c_void_p(mt.data_ptr()), C.c_void_p(stt.data_ptr())
    check(lib.shm_window_gather(C.c_void_p(series.data_ptr()), C.c_int64(n_rows), D, C.c_void_p(st.data_ptr()) if N else None, N, win_len,
                                mp, sp, C.c_float(float(clip_z)), C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "shm_window_gather")
    return out


# ---- run_splits.json (reference 4DOF/Scripts/00_make_run_splits.py:15-141): per file, contiguous train / val / test blocks of window indices
def contiguous_split(n_windows: int, train_frac: float = 0.40, val_frac: float = 0.30):
    n = max(int(n_windows), 0)
    n_tr, n_va = int(train_frac * n), int(val_frac * n)
    return {"train": list(range(n_tr)), "val": list(range(n_tr, n_tr + n_va)), "test": list(range(n_tr + n_va, n))}


def make_run_splits(root, seq_len: int = 100, stride: int = 1, train_frac: float = 0.40, val_frac: float = 0.30, seed: int = 42, write: bool = True):
    """Scans <root>/Data/raw (normal/*.csv, faults/sensor_fault/**, faults/structural_fault/**) and returns — and by default writes to
    Data/processed/run_splits.json — the split description every stage reads: mode "window_level_per_file", file paths relative to
    root with forward slashes, window indices 0 .. n_windows-1 cut 40 / 30 / rest per file without shuffling."""
    import json
    root = Path(root)
    out = {"mode": "window_level_per_file", "seed": seed,
           "fractions": {"train": train_frac, "val": val_frac, "test": round(1.0 - train_frac - val_frac, 10)},
           "seq_len": seq_len, "stride": stride}
    totals = {}
    for key, pattern in (("normal", "Data/raw/normal/*.csv"), ("sensor_fault", "Data/raw/faults/sensor_fault/**/*.csv"),
                         ("structural_fault", "Data/raw/faults/structural_fault/**/*.csv")):
        files, widx, tot = [], {}, {"train": 0, "val": 0, "test": 0}
        for fp in sorted(root.glob(pattern)):
            with open(fp, "r", encoding="utf-8", errors="ignore") as fh:
                n_rows = max(sum(1 for _ in fh) - 1, 0)
            n_win = 0 if n_rows < seq_len else 1 + (n_rows - seq_len) // stride
            if n_win <= 0:
                continue
            rel = fp.relative_to(root).as_posix()
            files.append(rel)
            widx[rel] = contiguous_split(n_win, train_frac, val_frac)
            for k in tot:
                tot[k] += len(widx[rel][k])
        out[key] = {"files": files, "window_indices": widx}
        totals[key] = tot
    out["totals"] = totals
    out["note"] = "contiguous time-block split per file (no shuffle)"
    if write:
        dst = root / "Data/processed/run_splits.json"
        dst.parent.mkdir(parents=True, exist_ok=True)
        dst.write_text(json.dumps(out, indent=2), encoding="utf-8")
    return out


# ---- 1_DOF evaluation helpers (1_DOF/Scripts/datasets.py:38-71)
def stitch_windows(windows: np.ndarray, full_len: int, stride: int = 1) -> np.ndarray:
    """Average the overlapping sliding-window reconstructions back into one series [full_len, F] (fp64, as the reference)."""
    N, seq_len, F = windows.shape
    out = np.zeros((full_len, F), dtype=float)
    cnt = np.zeros((full_len, 1), dtype=float)
    if stride == 1 and N > 0 and N + seq_len - 1 <= full_len:
        # every window n covers rows n .. n+seq_len-1: accumulate diagonal by diagonal instead of window by window
        for t in range(seq_len):
            out[t:t + N] += windows[:, t, :]
            cnt[t:t + N] += 1.0
    else:
        for n in range(N):
            out[n * stride:n * stride + seq_len] += windows[n]
            cnt[n * stride:n * stride + seq_len] += 1.0
    cnt[cnt == 0.0] = 1.0
    return out / cnt


def segment_rmse(y_true: np.ndarray, y_pred: np.ndarray, segment_len: int) -> np.ndarray:
    T = y_true.shape[0]
    S = int(np.ceil(T / segment_len))
    e2 = (np.asarray(y_pred, dtype=float) - np.asarray(y_true, dtype=float)) ** 2
    return np.array([float(np.sqrt(np.mean(e2[s * segment_len:min((s + 1) * segment_len, T)]))) for s in range(S)], dtype=float)
