"""The C ABI driven from plain C (tests/c_abi/forward_demo.c: hipMalloc'd buffers, no Python objects in the call) must reproduce the
Python drop-in module bit for bit — both are thin callers of the same libshm_hip.so — and the reference golden within tolerance."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT, load_golden, split_prefix
from test_gpu_vae import inject_eps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    exe = tmp_path_factory.mktemp("c_abi") / "forward_demo"
    csrc = PKG / "csrc"
    # plain gcc, C99: the header and the entry points are C; -D__HIP_PLATFORM_AMD__ only selects the platform in HIP's own headers
    cmd = ["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", str(ROOT / "tests/c_abi/forward_demo.c"), "-I", str(ROOT / "include"), "-I", "/opt/rocm/include",
           "-L", str(csrc), "-lshm_hip", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_plain_c_caller_matches_python_module_and_golden(demo, tmp_path, precision):
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip import _capi
    g = load_golden("vae_fwd_ns_h128")
    cfg, p = g["cfg"], split_prefix(g, "p.")
    m = TemporalVAE(cfg["input_dim"], cfg["latent_dim"], cfg["hidden_dim"], cfg["num_layers"], cfg["dropout"], precision=precision)
    m.load_state_dict({k: torch.tensor(v) for k, v in p.items()}, strict=True)
    m = m.cuda().eval()
    x, eps = g["x"], g["eps"]
    B, T, D = x.shape
    with torch.no_grad(), inject_eps([torch.tensor(eps)]):
        recon, mu, lv = m(torch.tensor(x).cuda())
    with inject_eps([torch.tensor(eps)]):
        score = m.score(torch.tensor(x).cuda())
    # flat parameter buffer in shm_vae_param_layout order
    c = _capi.make_config(cfg["input_dim"], cfg["latent_dim"], cfg["hidden_dim"], cfg["num_layers"], cfg["layer_norm"], cfg["dropout"])
    layout, n_flat = _capi.param_layout(c)
    flat = np.zeros(n_flat, dtype=np.float32)
    for name, off, shape in layout:
        flat[off:off + p[name].size] = p[name].reshape(-1)
    hd = np.array([D, cfg["latent_dim"], cfg["hidden_dim"], cfg["num_layers"], int(cfg["layer_norm"]), 0 if precision == "fp32" else 1, B, T, n_flat],
                  dtype=np.int32)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(hd.tobytes()); f.write(flat.tobytes()); f.write(x.astype(np.float32).tobytes()); f.write(eps.astype(np.float32).tobytes())
    r = subprocess.run([str(demo), str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype=np.float32)
    nx, nz = B * T * D, B * cfg["latent_dim"]
    c_recon, c_mu, c_lv, c_score = out[:nx].reshape(B, T, D), out[nx:nx + nz], out[nx + nz:nx + 2 * nz], out[nx + 2 * nz:]
    assert np.array_equal(c_recon, recon.cpu().numpy()) and np.array_equal(c_mu, mu.cpu().numpy().reshape(-1))
    assert np.array_equal(c_lv, lv.cpu().numpy().reshape(-1)) and np.array_equal(c_score, score.cpu().numpy())
    tol = 1e-4 if precision == "fp32" else 2e-3
    assert np.abs(c_recon - g["recon"]).max() < tol
