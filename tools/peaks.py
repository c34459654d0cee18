"""Measured ceilings of the box: bf16 / fp32 MFMA issue rate with register operands, HBM read / write / copy.
    python tools/peaks.py"""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hybrid-vae-cnn-for-shm_amd"))
from shm_hip._capi import check, lib   # noqa: E402

s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
scratch = torch.zeros(4, device="cuda")
val = C.c_double()
for name, bf in (("bf16 v_mfma_f32_16x16x32_bf16", 1), ("fp32 v_mfma_f32_16x16x4_f32", 0)):
    for bpc in (1, 2, 4):
        check(lib.shm_peak_mfma(bf, bpc, 20000, C.c_void_p(scratch.data_ptr()), C.byref(val), s), "peak_mfma")
        print(f"MFMA {name}: {bpc * 4:2d} waves per CU -> {val.value:8.1f} TFLOP/s", flush=True)
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.zero_()
for mode, nm in ((0, "read"), (1, "write"), (2, "copy (read + write)")):
    check(lib.shm_peak_hbm(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_size_t(n), mode, C.c_void_p(scratch.data_ptr()), C.byref(val), s), "peak_hbm")
    print(f"HBM {nm:20s} of {n >> 30} GiB: {val.value:8.1f} GB/s", flush=True)
