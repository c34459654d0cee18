#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into the per-kernel
HBM-traffic summary bench.py reads (profiles/rNN_<precision>_pmc_traffic.json).

    python profiles/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> ["note on the command"]

Run it in the tree the counters were collected on: it records the sha256 of the kernel sources (csrc/*.hip, *.hpp), and bench.py quotes
`roofline.traffic` from the summary only while those hashes are the hashes of its own tree (a stale summary yields null, with the reason).

Counter unit: KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half of a wide
coalesced read, so it is doubled here; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Values are averages per
launch, in bytes; the first launches of each kernel (warm-up) are included, which changes nothing at this traffic level.
"""
import csv
import hashlib
import json
import sys
from collections import defaultdict
from pathlib import Path


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    fetch, write, out = sys.argv[1:4]
    extra = sys.argv[4] if len(sys.argv) > 4 else ""
    fe, wr = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f2 = 2.0 * fe.get(k, 0.0) * 1024.0
        w = wr.get(k, 0.0) * 1024.0
        kernels[k] = {"fetch_bytes_corrected": int(round(f2)), "write_bytes": int(round(w)), "hbm_bytes": int(round(f2 + w))}
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), 1x MI355X. Counter unit KiB; FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); WRITE_SIZE exact. Averages per launch, bytes. " + extra)
    csrc = Path(__file__).resolve().parent.parent / "hybrid-vae-cnn-for-shm_amd" / "csrc"
    hashes = {f.name: hashlib.sha256(f.read_bytes()).hexdigest()[:16] for f in sorted(csrc.glob("*.h*"))}
    json.dump({"_note": note, "kernel_src_sha256": hashes, "kernels": kernels}, open(out, "w"), indent=1)
    print(f"{len(kernels)} kernels -> {out}")


if __name__ == "__main__":
    main()
