"""pytest configuration: markers, import paths, fixture helpers."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "hybrid-vae-cnn-for-shm_amd"          # plays the role of the reference's 4DOF/ root on sys.path
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    d = dict(np.load(GOLDEN / f"{name}.npz", allow_pickle=False))
    if "cfg" in d:
        d["cfg"] = json.loads(str(d["cfg"]))
    return d


def split_prefix(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture
def golden():
    return load_golden


def record_parity(test, case, errors):
    """Append the measured errors of a parity case to gpurun_out/parity_errors.jsonl (scratch; the round's summary is
    aggregated from it into profiles/rNN_parity_errors.json by tools/collect_parity.py)."""
    out = ROOT / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        with open(out / "parity_errors.jsonl", "a") as f:
            clean = {k: (v if isinstance(v, (int, str)) else float(v)) for k, v in errors.items()}      # numpy scalars are not JSON types
            f.write(json.dumps({"test": test, "case": case, "errors": clean}) + "\n")
    except OSError:
        pass
