"""The simulation oracle (oracle/simulation_oracle.py) against goldens produced by the reference's own generator functions
(tests/golden/make_golden_simulation.py; reference 4DOF/Scripts/utils/simulation_4dof.py, 02_generate_fault_datasets.py)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import simulation_oracle as so

G = np.load(Path(__file__).parent / "golden" / "simulation_4dof.npz")
DT, WINDOW = 0.01, 50


@pytest.mark.parametrize("tag", ["normal", "jitter", "fault"])
def test_force_smoothing(tag):
    got = so.smooth_force(G["base_" + tag], WINDOW)
    want = G["force_" + tag]
    assert got.dtype == np.float32 and got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6 * np.abs(want).max())


@pytest.mark.parametrize("tag", ["normal", "jitter", "fault", "stiff60"])
def test_matrices_and_response(tag):
    M, C, K = so.system_matrices(G["mass_" + tag], G["stiff_" + tag], float(G["zeta_" + tag]))
    np.testing.assert_allclose(M, G["M_" + tag], rtol=0, atol=0)
    np.testing.assert_allclose(K, G["K_" + tag], rtol=1e-15)
    np.testing.assert_allclose(C, G["C_" + tag], rtol=1e-10)
    force = G["force_fault" if tag == "stiff60" else "force_" + tag]
    got = so.newmark(M, C, K, force, DT)
    want = G["resp_" + tag]
    for blk in range(3):                       # x, v, a have very different scales
        w = want[:, 4 * blk:4 * blk + 4]
        np.testing.assert_allclose(got[:, 4 * blk:4 * blk + 4], w, rtol=0, atol=1e-9 * np.abs(w).max())


@pytest.mark.parametrize("kind,name", [(so.NOISE, "noise"), (so.SPIKES, "spikes"), (so.DRIFT, "drift"), (so.BIAS, "bias")])
def test_sensor_faults(kind, name):
    dof, rel = G[f"sf_{name}_spec"]
    dof = int(dof)
    z = G[f"sf_{name}_z"] if f"sf_{name}_z" in G.files else None
    idx = G[f"sf_{name}_idx"] if f"sf_{name}_idx" in G.files else None
    base = G["resp_fault"]
    got = so.inject(base, kind, dof, float(rel), z=z, idx=idx)
    cols = [dof - 1, dof + 3, dof + 7]
    want = G[f"sf_{name}_cols"]
    np.testing.assert_allclose(got[:, cols], want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    others = [c for c in range(12) if c not in cols]
    assert np.array_equal(got[:, others], base[:, others])
