#!/usr/bin/env python3
"""Assembly lint for lstm_fwd_sp_fused (csrc/lstm_sp.hip): its hand-off loads are inline-asm `buffer_load_dwordx2 ... sc1`, which the compiler
does not track; the kernel waits for them itself (`s_waitcnt vmcnt(N)` inside an asm block, x_ready).  Nothing may READ the destination
registers between a load and the first such wait behind it — a register copy there would copy stale bytes.  Usage: lint_pending_loads.py file.s"""
import re
import sys


def regs(tok):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def main(path):
    s = open(path).read()
    names = re.findall(r"^(_ZN3shm12_GLOBAL__N_123lstm_fwd_pair_sp_kernel\w+):", s, re.M)
    bad = checked = 0
    for name in names:
        i = s.index(name + ":")
        body = [l.strip() for l in s[i:s.index(".Lfunc_end", i)].split("\n")]
        waits = [k for k, l in enumerate(body) if l.startswith("s_waitcnt vmcnt") and k > 0 and "ASMSTART" in body[k - 1]]
        for k, l in enumerate(body):
            m = re.match(r"buffer_load_dwordx2 (v\[\d+:\d+\]).*sc1", l)
            if not m or "ASMSTART" not in body[k - 1]:
                continue
            pend = regs(m.group(1))
            nxt = [w for w in waits if w > k]
            if not nxt:
                print(name, "load at", k, "has no manual wait behind it")
                bad += 1
                continue
            checked += 1
            for l2 in body[k + 1:nxt[0]]:
                if not l2 or l2[0] in ";." or l2.startswith("buffer_load_dwordx2"):
                    continue
                parts = l2.split(None, 1)
                if len(parts) < 2:
                    continue
                _, _, srcs = parts[1].partition(",")
                is_store = parts[0].startswith(("buffer_store", "global_store", "ds_write", "scratch_store"))
                hit = regs(parts[1] if is_store else srcs) & pend
                if hit:
                    print(name, "reads pending", sorted(hit), "in:", l2)
                    bad += 1
    print(f"{len(names)} kernels, {checked} loads checked, {bad} problems")
    return 1 if bad or not checked else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
