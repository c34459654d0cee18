#!/usr/bin/env python3
"""Golden vectors for the 4-DOF data generators: smoothed Gaussian excitation + Newmark-beta chain
(reference 4DOF/Scripts/utils/simulation_4dof.py:25-137) and the sensor / structural fault injections
(reference 4DOF/Scripts/02_generate_fault_datasets.py:128-221), produced by the reference's own functions.

Run in the build container (imports from /root/reference; never on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_simulation.py
Writes tests/golden/simulation_4dof.npz:
  base_<c> / force_<c>      raw Gaussian draw (fp64) and the smoothed excitation the reference returns (fp32) for seed / rms of case c
  mass_<c>, stiff_<c>, zeta_<c>, M_<c>, C_<c>, K_<c>   the system and its matrices
  resp_<c>                  [steps, 12] fp64 response x1..x4, v1..v4, a1..a4
  cases: "normal" (default system, seed 2025, rms 50), "jitter" (a jittered system as 01_generate_normal_runs.py:123-126 draws it,
  seed 2027), "fault" (default system, seed 42, rms 200 = the fault script's baseline) and "stiff60" (all stiffnesses x 0.6, same force)
  sf_<kind>_*               sensor-fault injections applied to the "fault" baseline: the standard-normal draws / spike positions used
                            and the corrupted DOF's [steps, 3] columns (x_k, v_k, a_k); all other columns are unchanged
"""
import importlib.util
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/4DOF")
OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, str(path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod                      # dataclasses look the defining module up by name
    spec.loader.exec_module(mod)
    return mod


def main():
    sim = _load("ref_sim4", REF / "Scripts/utils/simulation_4dof.py")
    flt = _load("ref_faults4", REF / "Scripts/02_generate_fault_datasets.py")
    d = {}

    def case(tag, cfg, seed, rms, force=None):
        steps = int(cfg.T_total / cfg.dt) + 1
        if force is None:
            np.random.seed(seed)
            d["base_" + tag] = np.random.randn(steps, cfg.num_dofs) * rms          # the draw init_force starts from
            force = sim.init_force(cfg.T_total, cfg.dt, cfg.num_dofs, rms, seed)
            d["force_" + tag] = force.numpy()
        M, C, K = sim.compute_matrices(np.array(cfg.mass, float), np.array(cfg.stiffness, float), cfg.damping_ratio, cfg.num_dofs)
        df = sim.run_simulation(cfg, force)
        d.update({"mass_" + tag: np.array(cfg.mass), "stiff_" + tag: np.array(cfg.stiffness), "zeta_" + tag: np.float64(cfg.damping_ratio),
                  "M_" + tag: M, "C_" + tag: C, "K_" + tag: K, "resp_" + tag: df.to_numpy()})
        print(tag, df.shape, float(np.abs(df.to_numpy()[:, :4]).max()))
        return force, df

    case("normal", sim.default_system_config(), 2025, 50.0)

    base = sim.default_system_config()
    rng = np.random.RandomState(7)
    cfg = sim.default_system_config()
    cfg.mass = (np.array(base.mass) * rng.uniform(0.98, 1.02, size=4)).tolist()
    cfg.stiffness = (np.array(base.stiffness) * rng.uniform(0.98, 1.02, size=4)).tolist()
    cfg.damping_ratio = float(rng.uniform(0.015, 0.025))
    case("jitter", cfg, 2027, 50.0)

    force, normal_df = case("fault", sim.default_system_config(), flt.FORCE_SEED, flt.FORCE_RMS)
    cfg = sim.default_system_config()
    cfg.stiffness = (np.array(base.stiffness) * 0.6).tolist()
    case("stiff60", cfg, None, None, force=force)

    # sensor faults as generate_sensor_faults() applies them (02_generate_fault_datasets.py:190-221), with the random draws recorded
    specs = {"noise": (flt.inject_noise, 4, 0.50), "spikes": (flt.inject_spikes, 1, 5.00),
             "drift": (flt.inject_drift, 2, 10.0), "bias": (flt.inject_bias, 3, 2.00)}
    n = len(normal_df)
    for kind, (func, dof, rel) in specs.items():
        out = normal_df.copy()
        zs, idxs = [], []
        for c in flt.dof_triplet(dof):
            std = float(normal_df[c].std())
            mag = (std if std > 0 else 1.0) * rel
            seed = 1000 + len(zs) + 10 * dof
            if kind == "noise":
                np.random.seed(seed); zs.append(np.random.normal(0.0, 1.0, size=n))
            elif kind == "spikes":
                np.random.seed(seed); idx = np.random.choice(n, int(n * 0.01), replace=False)
                idxs.append(idx); zs.append(np.random.normal(0.0, 1.0, size=len(idx)))
            np.random.seed(seed)
            out[c] = func(normal_df[c].to_numpy(), magnitude=mag)
        d[f"sf_{kind}_cols"] = out[flt.dof_triplet(dof)].to_numpy()        # only the DOF's x, v, a columns change
        d[f"sf_{kind}_spec"] = np.array([dof, rel], dtype=np.float64)
        if zs:
            d[f"sf_{kind}_z"] = np.stack(zs)
        if idxs:
            d[f"sf_{kind}_idx"] = np.stack(idxs).astype(np.int32)
    np.savez_compressed(OUT / "simulation_4dof.npz", **d)
    print("wrote", OUT / "simulation_4dof.npz", {k: v.shape for k, v in d.items() if k.startswith("sf_")})


if __name__ == "__main__":
    main()
