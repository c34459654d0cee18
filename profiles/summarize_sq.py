#!/usr/bin/env python3
"""Per-kernel averages of SQ counters from rocprofv3 --pmc passes (counter_collection.csv files) -> one JSON.

    python profiles/summarize_sq.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]

Derived ratios (where both counters are present in the inputs):
  mfma_busy_frac      = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   (both in cycles; the MFMA counter sums the CU's 4 SIMDs, so this is
                        the share of time a SIMD's matrix pipe is busy.  Cross-check: SQ_BUSY_CU_CYCLES / 256 CUs = kernel duration x ~2.0 GHz)
  lds_conflict_frac   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE       (extra LDS-array cycles per LDS-array cycle, MI355X_MICROARCH.md LDS section)
  wait_any_frac, wait_inst_frac, active_frac = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycle units, disjoint buckets)
Values are sums over all SEs/XCDs as rocprofv3 reports them, averaged per launch.
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                k, c = r["Kernel_Name"], r["Counter_Name"]
                tot[k][c] += float(r["Counter_Value"]); cnt[k][c] += 1
    res = {}
    for k in sorted(tot):
        if "shm::" not in k and "_ZN3shm" not in k:
            continue
        avg = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
        d = {c: round(v, 1) for c, v in avg.items()}
        d["launches"] = max(cnt[k].values())

        def ratio(a, b):
            scale = 4.0 if a == "SQ_VALU_MFMA_BUSY_CYCLES" else 1.0
            return round(avg[a] / (scale * avg[b]), 4) if a in avg and b in avg and avg[b] > 0 else None
        for name, a, b in (("mfma_busy_frac", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"), ("lds_conflict_frac", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
                           ("wait_any_frac", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ("wait_inst_frac", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
                           ("active_frac", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")):
            v = ratio(a, b)
            if v is not None:
                d[name] = v
        res[k.replace('(anonymous namespace)::', '')[:200]] = d
    json.dump({"_note": "rocprofv3 --pmc SQ passes on `python3 bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline --no-fp32-leg "
                        "--profile-steps 0`, 1x MI355X, round 1; per-launch averages", "kernels": res}, open(out, "w"), indent=1)
    print(len(res), "kernels ->", out)


if __name__ == "__main__":
    main()
