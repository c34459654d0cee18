"""Builds tests/golden/reference_4dof_inputs.npz: the NUMERIC CONTENTS of the reference's 20 shipped 4DOF runs
(/root/reference/4DOF/Data/raw: 10 normal, 4 sensor-fault, 6 structural-fault CSV tables of 1001 x 12) as fp32 — inputs only,
no reference code, no outputs.  The stage tests (tests/test_gpu_reference_artifacts.py) rebuild the reference's directory layout
from it on the GPU box, run stages 00 -> 03 -> 04 -> 05 and compare with the quality numbers the reference holds in
4DOF/Data/processed/{stage1_vae_train_meta,vae_threshold,stage2_cnn_train_meta}.json (copied into this fixture as `held`).

fp32 is what every stage sees: the loaders cast the CSV text to float32 first (4DOF/Scripts/03_train_vae.py:81-88).
Run in the build container only (needs /root/reference):  python tests/golden/make_reference_inputs.py
"""
import json
from pathlib import Path

import numpy as np

REF = Path("/root/reference/4DOF")
OUT = Path(__file__).resolve().parent / "reference_4dof_inputs.npz"


def main():
    files = sorted((REF / "Data/raw").rglob("*.csv"))
    assert len(files) == 20, len(files)
    names, tables, header = [], [], None
    for fp in files:
        with open(fp) as fh:
            hdr = fh.readline().strip()
        header = header or hdr
        assert hdr == header
        X = np.loadtxt(fp, delimiter=",", skiprows=1, dtype=np.float64, ndmin=2).astype(np.float32)
        assert X.shape == (1001, 12), (fp, X.shape)
        names.append(fp.relative_to(REF).as_posix())
        tables.append(X)
    held = {}
    for key, name in (("vae_train", "stage1_vae_train_meta.json"), ("threshold", "vae_threshold.json"), ("cnn_train", "stage2_cnn_train_meta.json")):
        d = json.loads((REF / "Data/processed" / name).read_text())
        held[key] = {k: v for k, v in d.items() if not k.endswith("_path")}
    totals = json.loads((REF / "Data/processed/run_splits.json").read_text())["totals"]
    held["run_splits_totals"] = totals
    np.savez_compressed(OUT, names=np.array(names), tables=np.stack(tables), header=np.array(header), held=np.array(json.dumps(held)))
    print(OUT, OUT.stat().st_size, "bytes;", len(names), "tables")


if __name__ == "__main__":
    main()
