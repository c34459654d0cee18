"""CPU restatement of the 4-DOF synthetic-data generators — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path is the HIP library
(csrc/simulate.hip behind shm_hip/simulate.py) and never routes through here.  fp64 numpy throughout, written for clarity.  Pinned by
tests/golden/simulation_4dof.npz, which tests/golden/make_golden_simulation.py produced with the reference's own functions:

  smooth_force     centred rolling mean of the excitation          reference 4DOF/Scripts/utils/simulation_4dof.py:25-42
  system_matrices  chain stiffness + Rayleigh damping (modes 1, 2) reference simulation_4dof.py:45-77
  newmark          Newmark-beta time stepping, clipped states       reference simulation_4dof.py:80-137
  inject           sensor faults on one DOF's x/v/a columns          reference 4DOF/Scripts/02_generate_fault_datasets.py:128-149,190-213
"""
from __future__ import annotations

import numpy as np

NOISE, SPIKES, DRIFT, BIAS = 0, 1, 2, 3
CLIP = 1e5


def smooth_force(base: np.ndarray, window: int) -> np.ndarray:
    """base [steps, nd] fp64 -> fp32 [steps, nd]: mean over rows [t - window//2, t + (window-1)//2] that exist (pandas
    rolling(window, center=True, min_periods=1).mean()), then the fp32 cast of torch.tensor(..., dtype=float32)."""
    steps = base.shape[0]
    window = max(int(window), 1)
    out = np.empty_like(base, dtype=np.float64)
    for t in range(steps):
        lo = max(0, t - window // 2)
        hi = min(steps, t + (window - 1) // 2 + 1)
        out[t] = base[lo:hi].sum(axis=0) / (hi - lo)
    return out.astype(np.float32)


def system_matrices(mass, stiffness, zeta: float):
    """(M, C, K) for a chain fixed at the first spring; Rayleigh C = alpha M + beta K with both of the two lowest modes at damping
    ratio zeta (alpha clamped >= 0, beta >= 1e-4)."""
    m = np.asarray(mass, dtype=np.float64)
    k = np.asarray(stiffness, dtype=np.float64)
    nd = m.size
    K = np.zeros((nd, nd))
    for i in range(nd):                       # spring i joins mass i-1 (or the ground) to mass i
        K[i, i] += k[i]
        if i + 1 < nd:
            K[i, i] += k[i + 1]
            K[i, i + 1] = K[i + 1, i] = -k[i + 1]
    M = np.diag(m)
    lam = np.linalg.eigvals(K / m[:, None]).real
    w = np.sqrt(np.sort(lam[lam > 0]))
    if w.size < 2:
        alpha, beta = 0.1, 0.001
    else:
        w1, w2 = w[0], w[1]
        alpha, beta = np.linalg.solve(np.array([[0.5 / w1, 0.5 * w1], [0.5 / w2, 0.5 * w2]]), np.array([zeta, zeta]))
        alpha, beta = max(alpha, 0.0), max(beta, 1e-4)
    return M, alpha * M + beta * K, K


def newmark(M, C, K, force: np.ndarray, dt: float, beta: float = 0.25, gamma: float = 0.5) -> np.ndarray:
    """force [steps, nd] -> [steps, 3 nd] (x | v | a), zero initial displacement and velocity."""
    force = np.asarray(force, dtype=np.float64)
    steps, nd = force.shape
    x = np.zeros(nd); v = np.zeros(nd)
    a = np.nan_to_num(np.linalg.inv(M) @ (force[0] - C @ v - K @ x), nan=0.0)
    c0, c1, c2 = 1.0 / (beta * dt * dt), gamma / (beta * dt), 1.0 / (beta * dt)
    c3, c4, c5 = 0.5 / beta - 1.0, gamma / beta - 1.0, 0.5 * dt * (gamma / beta - 2.0)
    Kinv = np.linalg.inv(c0 * M + c1 * C + K)
    out = np.zeros((steps, 3 * nd))
    out[0, 2 * nd:] = a
    for i in range(1, steps):
        rhs = force[i] + M @ (c0 * x + c2 * v + c3 * a) + C @ (c1 * x + c4 * v + c5 * a)
        xn = Kinv @ rhs
        an = c0 * (xn - x) - c2 * v - c3 * a
        vn = v + dt * ((1.0 - gamma) * a + gamma * an)
        x, v, a = np.clip(xn, -CLIP, CLIP), np.clip(vn, -CLIP, CLIP), np.clip(an, -CLIP, CLIP)
        out[i, :nd], out[i, nd:2 * nd], out[i, 2 * nd:] = x, v, a
    return out


def inject(table: np.ndarray, kind: int, dof: int, rel_mag: float, z=None, idx=None) -> np.ndarray:
    """table [steps, 3 nd]; corrupts columns (x_dof, v_dof, a_dof), dof 1-based.  Magnitude = rel_mag x the column's sample standard
    deviation (ddof = 1; 1.0 if that is 0).  z: standard-normal draws ([3, steps] for NOISE, [3, n_spikes] for SPIKES);
    idx [3, n_spikes]: spike positions."""
    out = np.array(table, dtype=np.float64, copy=True)
    steps = out.shape[0]
    nd = out.shape[1] // 3
    for j in range(3):
        col = (dof - 1) + j * nd
        sd = float(np.std(table[:, col], ddof=1))
        mag = (sd if sd > 0 else 1.0) * rel_mag
        if kind == NOISE:
            out[:, col] += mag * z[j]
        elif kind == SPIKES:
            out[idx[j], col] += mag + 0.25 * mag * z[j]
        elif kind == DRIFT:
            out[:, col] += np.linspace(0.0, mag, steps)
        elif kind == BIAS:
            out[:, col] += mag
        else:
            raise ValueError(kind)
    return out
