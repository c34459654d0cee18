"""Synthetic 4-DOF data on the device: excitation, Newmark-beta responses and the fault datasets.

Host-side mirror of reference 4DOF/Scripts/utils/simulation_4dof.py (SystemConfig :13-22, init_force :25-42, compute_matrices :45-77,
run_simulation :80-137, default_system_config :140-152) and of the generators that call it (01_generate_normal_runs.py:113-131,
02_generate_fault_datasets.py:128-221), batched over runs: the reference integrates one run at a time in a Python loop and writes a
CSV; here R runs are one kernel launch (csrc/simulate.hip: one thread per run, fp64) and the response tables stay on the device,
ready for dataprep.device_windows.  What the host keeps is what is tiny and per run: the nd x nd system matrices (an eigenproblem for
the Rayleigh coefficients) and the inverse of the effective stiffness.

The Gaussian draws come from torch's device generator, not from numpy's MT19937: a seed gives the same distribution as the reference,
not the same numbers.  Pass `base=` (or `z=` / `idx=` for the sensor faults) to inject specific draws — the parity tests do.
There is no CPU path: the tensors must live on the GPU and the HIP library must be built.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Sequence

import numpy as np
import torch

from ._capi import check, lib

NOISE, SPIKES, DRIFT, BIAS = 0, 1, 2, 3
_KINDS = {"noise": NOISE, "spikes": SPIKES, "drift": DRIFT, "bias": BIAS}


@dataclass
class SystemConfig:
    mass: List[float]
    stiffness: List[float]
    damping_ratio: float
    beta: float = 0.25
    gamma: float = 0.5
    num_dofs: int = 4
    dt: float = 0.01
    T_total: float = 10.0

    @property
    def steps(self) -> int:
        return int(self.T_total / self.dt) + 1


def default_system_config() -> SystemConfig:
    m0, k0 = 50.0, 200000.0
    return SystemConfig(mass=[m0 * 1.2, m0, m0, m0 * 0.8], stiffness=[k0 * 1.5, k0 * 1.2, k0, k0 * 0.8], damping_ratio=0.02)


def columns(num_dofs: int = 4) -> List[str]:
    return [f"{q}{j + 1}" for q in "xva" for j in range(num_dofs)]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t: torch.Tensor, what: str) -> None:
    if t.device.type != "cuda":
        raise RuntimeError(f"{what} must be a CUDA tensor (the simulator has no CPU path)")


# ------------------------------------------------------------------------------------------------ system matrices (host, batched)
def chain_matrices(mass, stiffness, zeta):
    """mass, stiffness [R, nd], zeta [R] -> (m [R, nd], C [R, nd, nd], K [R, nd, nd]) in fp64: spring i ties mass i to mass i-1 (the
    first one to the ground); Rayleigh damping alpha M + beta K with the two lowest modes at ratio zeta, alpha >= 0, beta >= 1e-4."""
    m = np.atleast_2d(np.asarray(mass, dtype=np.float64))
    k = np.atleast_2d(np.asarray(stiffness, dtype=np.float64))
    z = np.broadcast_to(np.asarray(zeta, dtype=np.float64).reshape(-1), (m.shape[0],))
    R, nd = m.shape
    K = np.zeros((R, nd, nd))
    i = np.arange(nd)
    K[:, i, i] = k
    K[:, i[:-1], i[:-1]] += k[:, 1:]
    K[:, i[:-1], i[1:]] = -k[:, 1:]
    K[:, i[1:], i[:-1]] = -k[:, 1:]
    lam = np.linalg.eigvals(K / m[:, :, None]).real
    alpha = np.full(R, 0.1)
    beta = np.full(R, 0.001)
    for r in range(R):
        w = np.sqrt(np.sort(lam[r][lam[r] > 0]))
        if w.size >= 2:
            A = np.array([[0.5 / w[0], 0.5 * w[0]], [0.5 / w[1], 0.5 * w[1]]])
            a_, b_ = np.linalg.solve(A, np.array([z[r], z[r]]))
            alpha[r], beta[r] = max(a_, 0.0), max(b_, 1e-4)
    Cm = beta[:, None, None] * K
    Cm[:, i, i] += alpha[:, None] * m
    return m, Cm, K


def pack_systems(configs: Sequence[SystemConfig]) -> np.ndarray:
    """[R, nd + 2 nd^2] fp64 rows {mass | C | (a0 M + a1 C + K)^-1} for shm_sim_newmark; all configs must share nd, dt, beta, gamma."""
    c0 = configs[0]
    for c in configs:
        if (c.num_dofs, c.dt, c.beta, c.gamma, c.T_total) != (c0.num_dofs, c0.dt, c0.beta, c0.gamma, c0.T_total):
            raise ValueError("pack_systems: every run of a batch must share num_dofs, dt, T_total, beta and gamma")
        if len(c.mass) != c.num_dofs or len(c.stiffness) != c.num_dofs:
            raise ValueError("pack_systems: mass / stiffness length must equal num_dofs")
    nd = c0.num_dofs
    m, Cm, K = chain_matrices([c.mass for c in configs], [c.stiffness for c in configs], [c.damping_ratio for c in configs])
    a0, a1 = 1.0 / (c0.beta * c0.dt ** 2), c0.gamma / (c0.beta * c0.dt)
    Keff = a1 * Cm + K
    i = np.arange(nd)
    Keff[:, i, i] += a0 * m
    Kinv = np.linalg.inv(Keff)
    return np.concatenate([m, Cm.reshape(len(configs), -1), Kinv.reshape(len(configs), -1)], axis=1)


# ------------------------------------------------------------------------------------------------ excitation
def smooth_force(base: torch.Tensor, window: int) -> torch.Tensor:
    """base [R, steps, nd] (or [steps, nd]) fp64 on the device -> fp32 centred rolling mean (min_periods = 1), simulation_4dof.py:31-42."""
    _need_cuda(base, "smooth_force: base")
    squeeze = base.dim() == 2
    b = (base.unsqueeze(0) if squeeze else base).to(torch.float64).contiguous()
    R, steps, nd = b.shape
    out = torch.empty(R, steps, nd, dtype=torch.float32, device=b.device)
    check(lib.shm_sim_smooth_force(C.c_void_p(b.data_ptr()), R, steps, nd, max(int(window), 1), C.c_void_p(out.data_ptr()), _stream()),
          "shm_sim_smooth_force")
    return out[0] if squeeze else out


def init_force(T_total: float, dt: float, num_dofs: int, rms: float, seed: int, n_runs: int | None = None, device="cuda",
               base: torch.Tensor | None = None) -> torch.Tensor:
    """Smoothed Gaussian excitation [steps, num_dofs] fp32 (or [n_runs, steps, num_dofs]); the half-second smoothing window of the
    reference.  `base` (fp64, already scaled by rms) replaces the draw."""
    steps = int(T_total / dt) + 1
    if base is None:
        g = torch.Generator(device=device).manual_seed(int(seed))
        shape = (steps, num_dofs) if n_runs is None else (int(n_runs), steps, num_dofs)
        base = torch.randn(shape, generator=g, dtype=torch.float64, device=device) * float(rms)
    return smooth_force(base, max(int(0.5 / dt), 1))


# ------------------------------------------------------------------------------------------------ Newmark-beta
def run_simulation(cfg, force: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """cfg: one SystemConfig or a sequence of R of them; force fp32 [steps, nd] (shared by all runs) or [R, steps, nd].
    Returns the response table(s) [steps, 3 nd] / [R, steps, 3 nd] with columns x1..xN, v1..vN, a1..aN (columns())."""
    single = isinstance(cfg, SystemConfig)
    cfgs = [cfg] if single else list(cfg)
    _need_cuda(force, "run_simulation: force")
    if dtype not in (torch.float32, torch.float64):
        raise ValueError("run_simulation: dtype must be torch.float32 or torch.float64")
    c0 = cfgs[0]
    R, nd, steps = len(cfgs), c0.num_dofs, c0.steps
    f = force.to(torch.float32).contiguous()
    shared = f.dim() == 2
    if tuple(f.shape[-2:]) != (steps, nd) or (not shared and f.shape[0] != R):
        raise ValueError(f"run_simulation: force shape {tuple(f.shape)} does not match [{'' if shared else str(R) + ', '}{steps}, {nd}]")
    sys_d = torch.from_numpy(pack_systems(cfgs)).to(f.device)
    out = torch.empty(R, steps, 3 * nd, dtype=dtype, device=f.device)
    check(lib.shm_sim_newmark(C.c_void_p(sys_d.data_ptr()), C.c_void_p(f.data_ptr()), int(shared), R, steps, nd,
                              C.c_double(c0.dt), C.c_double(c0.beta), C.c_double(c0.gamma),
                              C.c_void_p(out.data_ptr()), int(dtype == torch.float32), _stream()), "shm_sim_newmark")
    return out[0] if single else out


# ------------------------------------------------------------------------------------------------ dataset generators
def normal_runs(n_runs: int = 10, base_seed: int = 2025, force_rms: float = 50.0, jitter_seed: int | None = None, device="cuda",
                dtype=torch.float32):
    """01_generate_normal_runs.py:113-131 as one launch: per run the masses and stiffnesses are jittered by U(0.98, 1.02) and the
    damping ratio drawn from U(0.015, 0.025); returns (tables [n_runs, steps, 12], configs)."""
    rng = np.random.RandomState(jitter_seed)
    base = default_system_config()
    cfgs = []
    for _ in range(n_runs):
        c = default_system_config()
        c.mass = (np.array(base.mass) * rng.uniform(0.98, 1.02, size=base.num_dofs)).tolist()
        c.stiffness = (np.array(base.stiffness) * rng.uniform(0.98, 1.02, size=base.num_dofs)).tolist()
        c.damping_ratio = float(rng.uniform(0.015, 0.025))
        cfgs.append(c)
    force = init_force(base.T_total, base.dt, base.num_dofs, force_rms, base_seed, n_runs=n_runs, device=device)
    return run_simulation(cfgs, force, dtype=dtype), cfgs


def structural_faults(base_cfg: SystemConfig, force: torch.Tensor, reductions=(0.9, 0.8, 0.70, 0.60), dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """02_generate_fault_datasets.py:152-170: every stiffness scaled by `perc`, same excitation; one launch for all cases."""
    cfgs, labels = [], []
    for perc in reductions:
        c = SystemConfig(mass=list(base_cfg.mass), stiffness=(np.array(base_cfg.stiffness) * perc).tolist(),
                         damping_ratio=base_cfg.damping_ratio, beta=base_cfg.beta, gamma=base_cfg.gamma, num_dofs=base_cfg.num_dofs,
                         dt=base_cfg.dt, T_total=base_cfg.T_total)
        cfgs.append(c)
        labels.append(f"stiff_red_{int((1.0 - perc) * 100)}pct")
    tables = run_simulation(cfgs, force, dtype=dtype)
    return {lab: tables[i] for i, lab in enumerate(labels)}


def inject_sensor_fault(table: torch.Tensor, kind, dof: int, rel_mag: float, z: torch.Tensor | None = None,
                        idx: torch.Tensor | None = None, freq: float = 0.01, generator: torch.Generator | None = None) -> torch.Tensor:
    """A corrupted COPY of table [steps, 3 nd] or [R, steps, 3 nd] (fp32 / fp64, device): the x, v, a columns of `dof` (1-based) get
    the fault, magnitude = rel_mag x the column's sample standard deviation (02_generate_fault_datasets.py:128-149,205-209).
    z: standard-normal draws [R, 3, steps] (noise) or [R, 3, n_spikes] (spikes); idx: int32 spike rows [R, 3, n_spikes]; drawn here
    from `generator` when omitted (spikes: int(steps * freq) distinct rows per column)."""
    _need_cuda(table, "inject_sensor_fault: table")
    k = _KINDS[kind] if isinstance(kind, str) else int(kind)
    squeeze = table.dim() == 2
    t = (table.unsqueeze(0) if squeeze else table).contiguous()
    if t.dtype not in (torch.float32, torch.float64):
        raise ValueError("inject_sensor_fault: table must be fp32 or fp64")
    R, steps, ncols = t.shape
    nd = ncols // 3
    nz = 0
    if k == NOISE:
        if z is None:
            z = torch.randn(R, 3, steps, generator=generator, dtype=torch.float64, device=t.device)
        nz = steps
    elif k == SPIKES:
        if idx is None:
            ns = int(steps * freq)
            idx = torch.rand(R, 3, steps, generator=generator, device=t.device).argsort(dim=-1)[..., :ns]
        nz = int(idx.shape[-1])
        if z is None:
            z = torch.randn(R, 3, nz, generator=generator, dtype=torch.float64, device=t.device)
        idx = idx.to(device=t.device, dtype=torch.int32).reshape(R, 3, nz).contiguous()
    if z is not None:
        z = z.to(device=t.device, dtype=torch.float64).reshape(R, 3, nz).contiguous()
    out = torch.empty_like(t)
    check(lib.shm_sim_sensor_fault(C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), int(t.dtype == torch.float32), R, steps, nd, k, int(dof), C.c_double(rel_mag),
                                   C.c_void_p(z.data_ptr()) if z is not None and nz else None,
                                   C.c_void_p(idx.data_ptr()) if idx is not None and nz else None, nz, _stream()), "shm_sim_sensor_fault")
    return out[0] if squeeze else out


# the four sensor-fault datasets of 02_generate_fault_datasets.py:193-198
SENSOR_FAULTS = {"noise_x4": ("noise", 4, 0.50), "spikes_x1": ("spikes", 1, 5.00), "drift_x2": ("drift", 2, 10.0), "bias_x3": ("bias", 3, 2.00)}


def sensor_faults(normal_table: torch.Tensor, specs=None, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=normal_table.device).manual_seed(int(seed))
    return {name: inject_sensor_fault(normal_table, kind, dof, rel, generator=g) for name, (kind, dof, rel) in (specs or SENSOR_FAULTS).items()}


def fault_datasets(force_rms: float = 200.0, force_seed: int = 42, seed: int = 0, device="cuda", dtype=torch.float32):
    """main() of 02_generate_fault_datasets.py:224-234: the baseline run plus its structural and sensor fault variants."""
    cfg = default_system_config()
    force = init_force(cfg.T_total, cfg.dt, cfg.num_dofs, force_rms, force_seed, device=device)
    normal = run_simulation(cfg, force, dtype=dtype)
    return normal, structural_faults(cfg, force, dtype=dtype), sensor_faults(normal, seed=seed)


def save_csv(table: torch.Tensor, path) -> Path:
    """One run's table -> CSV with the reference's header row (x1..,v1..,a1..), readable by dataprep.load_csv_numeric."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    arr = table.detach().cpu().numpy()
    np.savetxt(str(path), arr, delimiter=",", header=",".join(columns(arr.shape[1] // 3)), comments="", fmt="%.17g" if arr.dtype == np.float64 else "%.9g")
    return path


def write_dataset(root, n_normal: int = 10, base_seed: int = 2025, normal_rms: float = 50.0, fault_rms: float = 200.0, fault_seed: int = 42,
                  jitter_seed: int | None = None, seed: int = 0, device="cuda") -> Dict[str, List[Path]]:
    """Everything 01_generate_normal_runs.py and 02_generate_fault_datasets.py leave under <root>/Data/raw (plots excluded), from two
    launches: normal/normal_seed<k>.csv, faults/structural_fault/<label>/<label>.csv, faults/sensor_fault/<name>/<name>.csv.
    Tables are integrated and written in fp64 like the reference's DataFrames."""
    root = Path(root)
    written: Dict[str, List[Path]] = {"normal": [], "structural_fault": [], "sensor_fault": []}
    tables, _ = normal_runs(n_normal, base_seed, normal_rms, jitter_seed, device, dtype=torch.float64)
    for i in range(n_normal):
        written["normal"].append(save_csv(tables[i], root / f"Data/raw/normal/normal_seed{base_seed + i}.csv"))
    _, structural, sensor = fault_datasets(fault_rms, fault_seed, seed, device, dtype=torch.float64)
    for label, tab in structural.items():
        written["structural_fault"].append(save_csv(tab, root / f"Data/raw/faults/structural_fault/{label}/{label}.csv"))
    for name, tab in sensor.items():
        written["sensor_fault"].append(save_csv(tab, root / f"Data/raw/faults/sensor_fault/{name}/{name}.csv"))
    return written


def s2_windows(n_runs: int = 64, win_len: int = 256, stride: int = 4, train_frac=(0.0, 0.7), val_frac=(0.7, 1.0), base_seed: int = 2025,
               jitter_seed: int = 0, force_rms: float = 50.0, device="cuda"):
    """The "S2" realism data of SURVEY.md §8d, built entirely on the device: jittered 4-DOF Newmark runs (displacement channels x1..x4 only,
    1001 steps), windows of win_len rows every `stride` rows from the train / val time fractions of every run, per-channel z-score
    from the train windows.  Returns (Ztr [Ntr, win_len, 4], Zva [Nva, win_len, 4], mean [4], std [4]) as fp32 device tensors."""
    from . import dataprep
    tables, _ = normal_runs(n_runs, base_seed, force_rms, jitter_seed, device)
    steps = tables.shape[1]
    series = tables[:, :, :4].reshape(n_runs * steps, 4).contiguous()          # runs back to back; windows never straddle two runs

    def starts(frac):
        lo, hi = int(steps * frac[0]), int(steps * frac[1])
        local = np.arange(lo, hi - win_len + 1, stride, dtype=np.int64)
        return (np.arange(n_runs, dtype=np.int64)[:, None] * steps + local[None, :]).reshape(-1)

    Wtr = dataprep.device_windows(series, starts(train_frac), win_len)
    flat = Wtr.reshape(-1, 4).double()
    mean = flat.mean(0).float()
    std = flat.std(0, unbiased=False).float()
    std = torch.where(std == 0, torch.full_like(std, 1e-6), std)
    Ztr = dataprep.device_windows(series, starts(train_frac), win_len, mean, std)
    Zva = dataprep.device_windows(series, starts(val_frac), win_len, mean, std)
    return Ztr, Zva, mean, std
