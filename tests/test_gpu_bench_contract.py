"""bench.py prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the roofline and cpu_baseline objects."""
import json
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*extra):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "2", *extra], cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def _check_roofline(rf, peak):
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] == "TFLOP/s" and rf["peak"] == peak
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0.005 < rf["frac"] < 1.0
    # bytes per launch from the committed PMC summary — or null WITH the reason when that summary was collected on other kernel sources
    assert (rf["traffic"] is not None and rf["traffic"] > 1e8) or rf.get("traffic_note")


def test_default_line_has_the_contract_fields():
    d = _run()
    assert d["metric"].startswith("windows/sec LSTM-VAE train") and d["unit"] == "windows/s" and d["value"] > 1e4
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 4, 2) and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 512 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-2 * d["value"]
    _check_roofline(d["roofline"], 2500.0)
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["unit"] == "windows/s" and cb["value"] > 1 and cb["cores"] >= 1 and cb["sample"]
    assert d["step_ms"]["p10"] <= d["step_ms"]["median"] <= d["step_ms"]["p90"]
    # the parity-grade fast path is a first-class leg: >= 50 timed steps, its own roofline object, and it clears 100 k windows/s
    pp = d["parity_path"]
    assert pp["precision"] == "split" and pp["dtype"] == "bf16x2" and pp["steps"] >= 50 and pp["value"] > 1e5, pp
    _check_roofline(pp["roofline"], 2500.0)
    assert d["fp32_exact_path"]["dtype"] == "f32" and d["fp32_exact_path"]["value"] > 1e3
    # inference side of the north star (scoring) and the classifier's train step ride in the same driver-run line
    inf = d["inference"]
    assert inf["bf16"]["value"] > inf["split"]["value"] > 1e5 and inf["split"]["unit"] == "windows/s"
    cn = d["cnn_train"]
    assert cn["metric"].startswith("windows/sec CNN train") and cn["value"] > 1e5 and cn["dtype"] == "f32"
    assert cn["roofline"]["kernel"].startswith("cnn.") and cn["cpu_baseline"]["kind"] == "port"


def test_default_flags_are_the_survey_protocol():
    """20 warm-up + 200 timed steps are bench.py's own defaults (the driver may override them)."""
    import re
    src = (ROOT / "bench.py").read_text()
    assert re.search(r'"--steps", type=int, default=200', src) and re.search(r'"--warmup", type=int, default=20', src)


def test_cnn_line():
    d = _run("--workload", "cnn")
    assert d["metric"].startswith("windows/sec CNN train") and d["value"] > 1e4 and d["eval_forward"]["value"] > d["value"]
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["kernel"].startswith("cnn.") and rf["peak"] == 157.3 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["frac"] > 0
    assert cb["kind"] == "port" and cb["value"] > 1 and cb["cores"] >= 1 and cb["sample"]
