#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: per launch start / end relative to the step's first launch, the stream
(queue) it ran on, and the idle gaps of the main stream.  Usage: step_timeline.py <kernel_trace.csv> [step_index_from_end=3]
Collect with:  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --precision split --steps 8 --warmup 4 \\
               --no-cpu-baseline --no-fp32-leg --no-parity-leg --no-inference-leg --no-cnn-leg --profile-steps 0"""
import csv
import re
import sys


def short(n):
    n = n.replace("void shm::(anonymous namespace)::", "").replace("shm::(anonymous namespace)::", "").replace("shm::", "")
    n = re.sub(r"\(shm::.*|\(int,.*|\(float.*", "", n)
    m = re.match(r"_ZN3shm12_GLOBAL__N_1\d+([a-z_0-9]+?)(?:I|E)", n)
    if m:
        n = m.group(1)
    return n[:58]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "pack_all_kernel" in r["Kernel_Name"]]
    a, b = starts[-back - 1], starts[-back]
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    queues = sorted({r["Queue_Id"] for r in step})
    main_q = step[0]["Queue_Id"]
    print(f"{len(step)} launches, step span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, queues {queues} (main {main_q})")
    last_end = t0
    busy = 0
    for r in step:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        q = "main" if r["Queue_Id"] == main_q else "side"
        gap = ""
        if q == "main":
            g = int(r["Start_Timestamp"]) - last_end
            gap = f"gap {g / 1e3:6.1f}" if g > 0 else ""
            last_end = max(last_end, int(r["End_Timestamp"]))
            busy += e - s
        print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  {q}  {short(r['Kernel_Name']):58s} {gap}")
    print(f"main-stream kernel time {busy / 1e3:.1f} us")


if __name__ == "__main__":
    main()
