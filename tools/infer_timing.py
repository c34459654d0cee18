"""Ad-hoc timing of the inference side (scoring, hybrid gate) at the north-star and openLAB-like shapes; quoted in DESIGN.md."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "hybrid-vae-cnn-for-shm_amd")):
    sys.path.insert(0, p)
import torch
from Scripts.Models.temporal_vae import TemporalVAE
from Scripts.Models.cnn_model import CNN
from shm_hip.pipeline import score_windows, hybrid_predict

def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

for (N, T, D, H, Z, L, prec) in ((8192, 256, 4, 128, 16, 2, "bf16"), (8192, 256, 4, 128, 16, 2, "fp32"), (8192, 512, 4, 64, 8, 1, "bf16"), (8192, 100, 12, 128, 16, 2, "bf16")):
    torch.manual_seed(0)
    vae = TemporalVAE(D, Z, H, L, 0.3, precision=prec).cuda().eval()
    cnn = CNN(2, 2, 0.5, seq_len=T, num_features=D).cuda().eval()
    X = torch.randn(N, T, D, device="cuda")
    n = 3 if prec == "fp32" else 10
    ds = timeit(lambda: score_windows(vae, X, 512), n)
    sc = score_windows(vae, X, 512)
    thr = float(sc.median())
    dh = timeit(lambda: hybrid_predict(vae, cnn, X, thr, 512), n)
    d1 = timeit(lambda: hybrid_predict(vae, cnn, X, thr, 512, reuse_recon=True), n)
    print(f"N={N} T={T} D={D} H={H} L={L} {prec}: score {N/ds:,.0f} windows/s ; hybrid gate+CNN (50% anomalous) {N/dh:,.0f} windows/s ; "
          f"single-pass hybrid {N/d1:,.0f} windows/s", flush=True)

sys.path.insert(0, str(ROOT / "hybrid-vae-cnn-for-shm_amd" / "Codes"))
from Models.cnn_model import CNN as OlCNN
for (B, T) in ((512, 200), (512, 512), (2048, 200)):
    torch.manual_seed(0)
    m = OlCNN(1, 2, 0.4).cuda().eval()
    x = torch.randn(B, 1, T, 4, device="cuda")
    with torch.no_grad():
        d = timeit(lambda: m(x), 5)
    print(f"openLAB CNN eval B={B} T={T}: {d*1e3:.3f} ms = {B/d:,.0f} windows/s ({B/d*135e6*T/200/1e12:.2f} TFLOP/s fp32)", flush=True)

# openLAB hybrid (10_test_hybrid_pipeline.py:266-302): L = 1 VAE gate on z-scored windows, the openLAB CNN on the anomalous half
from shm_hip.pipeline import hybrid_predict_openlab
for T in (200, 512):
    torch.manual_seed(0)
    vae = TemporalVAE(4, 8, 64, 1, 0.0, precision="bf16").cuda().eval()
    m = OlCNN(1, 2, 0.4).cuda().eval()
    N = 8192
    Zv = torch.randn(N, T, 4, device="cuda")
    Xc = torch.randn(N, T, 4, device="cuda")
    thr = float(score_windows(vae, Zv, 512).median())
    d = timeit(lambda: hybrid_predict_openlab(vae, m, Zv, Xc, thr, 0.5, 512), 5)
    print(f"openLAB hybrid N={N} T={T}: VAE gate + CNN on the anomalous half: {N/d:,.0f} windows/s", flush=True)
