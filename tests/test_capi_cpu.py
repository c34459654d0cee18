"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/shm_hip.h declares, and its
parameter layout is the reference's state_dict layout (no GPU compute calls here)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, load_golden, split_prefix


def _declared_symbols():
    text = (ROOT / "include" / "shm_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(shm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from shm_hip import _capi
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(_capi.lib, name), f"{name} declared in include/shm_hip.h but not exported"
    assert set(_capi.EXPORTS) == set(declared), set(_capi.EXPORTS) ^ set(declared)
    assert _capi.lib.shm_version() >= 1


@pytest.mark.parametrize("name", ["vae_fwd_ns_h128", "vae_fwd_ref_h128", "vae_fwd_1dof", "vae_fwd_openlab"])
def test_param_layout_is_the_reference_state_dict(name):
    from shm_hip import _capi
    g = load_golden(name)
    cfg = g["cfg"]
    c = _capi.make_config(cfg["input_dim"], cfg["latent_dim"], cfg["hidden_dim"], cfg["num_layers"], cfg["layer_norm"], cfg["dropout"])
    layout, total = _capi.param_layout(c)
    ref = split_prefix(g, "p.")
    assert [n for n, _, _ in layout] == list(ref.keys())          # same keys, same order
    for n, off, shape in layout:
        assert tuple(ref[n].shape) == shape, n
        assert off % 4 == 0
    n_true = sum(v.size for v in ref.values())
    assert n_true <= total < n_true + 4 * len(layout)


def test_invalid_configs_are_rejected_with_a_message():
    from shm_hip import _capi
    bad = _capi.make_config(4, 16, 100, 2, True, 0.3)             # hidden_dim 100 unsupported
    assert _capi.lib.shm_vae_param_count(C.byref(bad)) < 0
    assert b"hidden_dim" in _capi.lib.shm_last_error()
    handle = C.c_void_p(0)
    ok = _capi.make_config(4, 16, 128, 2, True, 0.3)
    assert _capi.lib.shm_vae_plan_create(C.byref(ok), 0, 256, 1, C.byref(handle)) != 0     # empty batch
    assert _capi.lib.shm_vae_plan_create(C.byref(ok), 512, 256, 1, C.byref(handle)) == 0
    nbytes = _capi.lib.shm_vae_plan_workspace_bytes(handle)
    assert 1 << 30 < nbytes < 8 << 30                              # ~2 GB of saved state at the north-star shape
    _capi.lib.shm_vae_plan_destroy(handle)


@pytest.mark.parametrize("field,value,word", [("hidden_dim", 48, "hidden_dim"), ("hidden_dim", 256, "hidden_dim"), ("latent_dim", 33, "latent_dim"),
                                              ("latent_dim", 0, "latent_dim"), ("num_layers", 3, "num_layers"), ("num_layers", 0, "num_layers"),
                                              ("input_dim", 17, "input_dim"), ("precision", 7, "precision"), ("dropout", 1.0, "dropout")])
def test_unsupported_shapes_return_einval_with_a_message(field, value, word):
    """The reference constructor accepts any sizes (temporal_vae.py:14-21); this build supports H in {32, 64, 128}, Z <= 32, L <= 2,
    D <= 16.  Outside that every entry point that takes a config returns SHM_EINVAL (-1) and shm_last_error() names the field;
    the Python mirror turns it into an exception at construction."""
    from shm_hip import _capi
    c = _capi.make_config(4, 16, 128, 2, True, 0.3)
    setattr(c, field, value)
    handle = C.c_void_p(0)
    assert _capi.lib.shm_vae_param_count(C.byref(c)) == -1
    assert _capi.lib.shm_vae_param_layout(C.byref(c), None, 0) == -1
    assert _capi.lib.shm_vae_plan_create(C.byref(c), 8, 16, 0, C.byref(handle)) == -1 and not handle.value
    msg = _capi.lib.shm_last_error().decode()
    assert word in msg and len(msg) > len(word), msg
    if field != "precision":
        from Scripts.Models.temporal_vae import TemporalVAE
        kw = dict(input_dim=4, latent_dim=16, hidden_dim=128, num_layers=2, dropout=0.3)
        kw[field] = value
        with pytest.raises((RuntimeError, ValueError), match=word):     # ValueError: torch's own nn.LSTM check (num_layers 0), as in the reference
            TemporalVAE(**kw)


def test_eps_counter_rng_replica_statistics():
    """shm_eps_normal: host replica of the head kernel's reparameterisation-noise stream — standard normal, stream-separated."""
    from shm_hip import _capi
    n = 1 << 16
    a = np.zeros(n, np.float32); b = np.zeros(n, np.float32); c2 = np.zeros(n, np.float32)
    _capi.lib.shm_eps_normal(C.c_uint64(42), C.c_uint64(1), n, a.ctypes.data_as(C.c_void_p))
    _capi.lib.shm_eps_normal(C.c_uint64(42), C.c_uint64(2), n, b.ctypes.data_as(C.c_void_p))
    _capi.lib.shm_eps_normal(C.c_uint64(42), C.c_uint64(1), n, c2.ctypes.data_as(C.c_void_p))
    assert np.array_equal(a, c2) and not np.array_equal(a, b)
    assert abs(a.mean()) < 0.02 and abs(a.std() - 1.0) < 0.02 and abs(np.corrcoef(a, b)[0, 1]) < 0.02
    assert abs((np.abs(a) > 1.959964).mean() - 0.05) < 0.005 and np.isfinite(a).all() and np.abs(a).max() < 6.5
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.02


def test_dropout_mask_replica_statistics():
    from shm_hip import _capi
    buf = np.zeros(64 * 50 * 32, dtype=np.uint8)
    _capi.lib.shm_dropout_keep_mask(C.c_uint64(77), 0, 64, 50, 32, C.c_float(0.3), buf.ctypes.data_as(C.c_void_p))
    assert abs(buf.mean() - 0.7) < 0.01
    buf2 = np.zeros_like(buf)
    _capi.lib.shm_dropout_keep_mask(C.c_uint64(77), 1, 64, 50, 32, C.c_float(0.3), buf2.ctypes.data_as(C.c_void_p))
    assert 0.3 < (buf != buf2).mean() < 0.5                       # independent streams per stack


def test_one_hip_runtime_whatever_the_import_order():
    """PyTorch bundles its own libamdhip64 with the system library's SONAME.  Importing the binding BEFORE torch must not leave two HIP
    runtimes in the process (the second one finds no device on a GPU box): _capi imports torch first."""
    import subprocess
    import sys

    from conftest import PKG
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from shm_hip import _capi\n"
            "import torch\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps').read().splitlines() if 'libamdhip64' in l})\n"
            "print(len(libs), libs)\n") % str(PKG)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert r.stdout.split()[0] == "1", r.stdout
