"""Throughput of the openLAB classifier forward (shm_olcnn_forward): windows/s and fp32 TFLOP/s at T = 200 and T = 512, W = 4, B = 512.
Set SHM_OLCNN_VALU=1 to time the vector-ALU convolution instead of the fp32-MFMA one.   python tools/olcnn_timing.py"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hybrid-vae-cnn-for-shm_amd"))
sys.path.insert(0, str(ROOT / "hybrid-vae-cnn-for-shm_amd" / "Codes"))
from Models.cnn_model import CNN   # noqa: E402


def flops_per_window(T, W=4):
    f, h, cin = 0, T, 1
    for cout, kt in ((32, 7), (64, 5), (128, 5), (256, 3)):
        f += 2 * h * W * cout * cin * kt * 3
        h //= 2 if cout != 256 else 1
        cin = cout
    return f + 2 * (256 * 128 + 128 * 2)


def main():
    torch.manual_seed(0)
    m = CNN(1, 2, 0.4).cuda().eval()
    for T in (200, 512):
        x = torch.randn(512, 1, T, 4, device="cuda")
        with torch.no_grad():
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                m(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"T={T}: {dt * 1e3:.3f} ms per 512 windows = {512 / dt:,.0f} windows/s, {512 / dt * flops_per_window(T) / 1e12:.1f} TFLOP/s fp32")


if __name__ == "__main__":
    main()
