"""Assembly lint of the fused split-precision forward (csrc/lstm_sp.hip: lstm_fwd_sp_fused).  Its in-launch hand-off loads are inline-asm
`buffer_load_dwordx2 ... sc1`, which the compiler does not track; the kernel waits for them itself with a counted `s_waitcnt vmcnt(N)`.
A register copy between such a load and that wait would copy stale bytes — and would pass every test on an idle chip.  The lint
(tools/lint_pending_loads.py) reads the gfx950 assembly of every instance of the kernel and fails on any read of a pending destination."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "hybrid-vae-cnn-for-shm_amd" / "csrc"


def test_no_read_of_a_pending_handoff_load(tmp_path):
    hipcc = Path("/opt/rocm/bin/hipcc")
    if not hipcc.exists():
        pytest.skip("no hipcc")
    flags = None
    for line in (CSRC / "Makefile").read_text().splitlines():
        if line.startswith("CXXFLAGS"):
            flags = line.split("=", 1)[1].replace("$(ARCH)", "gfx950").split()
    assert flags, "CXXFLAGS not found in the Makefile"
    out = tmp_path / "lstm_sp.s"
    subprocess.run([str(hipcc), *flags, "-S", "--cuda-device-only", str(CSRC / "lstm_sp.hip"), "-o", str(out)], check=True, cwd=CSRC)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "lint_pending_loads.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 problems" in r.stdout, r.stdout[-500:]
