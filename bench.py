#!/usr/bin/env python3
"""bench.py — LSTM-VAE training throughput (windows/s) on MI355X, BASELINE.json's metric.

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N=1 default)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (driver, N>1)

Workload (BASELINE.json configs[1]): TemporalVAE(input_dim=4, latent_dim=16, hidden_dim=128,
num_layers=2, dropout=0.3), seed-42 init, per-GPU batch 512 windows of T=256 x 4 channels, synthetic N(0,1)
windows resident in HBM before the timed region.  A step = forward + MSE/KL loss + backward + gradient all-reduce
(N>1) + clip_grad_norm_(2.0) + Adam(lr 1e-3, wd 1e-5), kl_w = 0.5 — 4DOF/Scripts/03_train_vae.py:260-271.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "hybrid-vae-cnn-for-shm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

B_PER_GPU, T, D, H, Z, L = 512, 256, 4, 128, 16, 2
KL_W = 0.5
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "bf16x2": 2500.0}       # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md


def alg_flops_forward_per_window():
    """F_alg = T[2G(D+H) + (L-1)4GH + L*4GH + 2HD] + 6HZ - 2GH(T-1), G = 4H (decoder-L0 projection hoisted)."""
    G = 4 * H
    F = T * (2 * G * (D + H) + (L - 1) * 2 * G * 2 * H + L * 2 * G * 2 * H + 2 * H * D) + 6 * H * Z
    return F - 2 * G * H * (T - 1)


def kernel_alg_flops(name, B, path="fp32"):
    """Algorithmic FLOPs of ONE launch of a profiled kernel (MFMA contractions only; pointwise excluded).
    path "bf16" / "split": dx = dpre.W_ih is a separate time-parallel GEMM (dx_gemm.*); "fp32": it is inside lstm_bwd.
    path "split": the input projections of layers 1 (and of decoder layer 0) are their own launches (xproj.*)."""
    G = 4 * H
    kind, _, inst = name.partition(".")
    if kind == "lstm_fwd":
        if inst in ("enc", "dec"):                                          # layer-pipelined launch: both layers of the stack
            fused = kernel_alg_flops(f"xproj.{inst}1", B, path) if path == "split" else 0     # split pair launch: layer 1's projection is inside
            return kernel_alg_flops(f"lstm_fwd.{inst}0", B, path) + kernel_alg_flops(f"lstm_fwd.{inst}1", B, path) + fused
        k_in = {"enc0": D, "enc1": H, "dec0": 0, "dec1": H}[inst]          # dec0: constant input, projected once
        if path == "split":
            return B * T * 2 * G * ((D if inst == "enc0" else 0) + H)
        return B * T * 2 * G * (k_in + H) + (B * 2 * G * H if inst == "dec0" else 0)
    if kind == "xproj":
        return B * 2 * G * H if inst == "dec0" else B * T * 2 * G * H
    if kind == "lstm_bwd":
        if inst in ("enc", "dec"):                                          # layer-pipelined launch incl. the dx product
            return 2 * B * T * 2 * G * H + B * T * 2 * G * H
        k_dx = 0 if path != "fp32" else {"enc0": 0, "enc1": H, "dec0": 0, "dec1": H}[inst]   # dx only for layers with a layer below
        return B * T * 2 * G * (H + k_dx)
    if kind == "dx_gemm":
        return B * T * 2 * G * H
    if kind == "dw_gemm":
        k_in = {"enc0": D, "enc1": H, "dec0": 0, "dec1": H, "out": 0}[inst]
        if inst == "out":
            return B * T * 2 * H * D
        return B * T * 2 * G * (H + k_in) + (B * 2 * G * H if inst == "dec0" else 0)
    return 0


PMC_SUMMARIES = {"bf16": ROOT / "profiles" / "r03_bf16_pmc_traffic.json", "split": ROOT / "profiles" / "r03_split_pmc_traffic.json",
                 "cnn": ROOT / "profiles" / "r03_cnn_pmc_traffic.json"}

# substrings of the demangled kernel names in the PMC summaries, per profile label
PMC_KERNELS = {
    "bf16": {"lstm_fwd": {"enc": "lstm_fwd_pair_bf16_kernel<128, true", "dec": "lstm_fwd_pair_bf16_kernel<128, false",
                          "enc0": "lstm_fwd_bf16_kernel<128, 1, true", "dec0": "lstm_fwd_bf16_kernel<128, 4, false, false, true",
                          "enc1": "lstm_fwd_bf16_kernel<128, 4, false, true, false", "dec1": "lstm_fwd_bf16_kernel<128, 4, false, true, false"},
             "lstm_bwd": {"enc": "lstm_bwd_pair_bf16_kernel<128, 0", "dec": "lstm_bwd_pair_bf16_kernel<128, 2",
                          "enc0": "lstm_bwd_bf16_kernel<128, 1>", "dec0": "lstm_bwd_bf16_kernel<128, 1>",
                          "enc1": "lstm_bwd_bf16_kernel<128, 0>", "dec1": "lstm_bwd_bf16_kernel<128, 2>"}},
    "cnn": {"cnn": {"conv1_fwd": "conv3x3_fwd_mfma_kernel<2, 16, true, false>", "conv2_fwd": "conv3x3_fwd_mfma_kernel<16, 32, true, false>",
                    "conv2_dx": "conv3x3_fwd_mfma_kernel<32, 16, false, true>", "conv2_dw": "conv3x3_bwd_weight_ksplit_kernel<16, 32>",
                    "conv1_dw": "conv3x3_bwd_weight_mfma_kernel<2, 16, true>", "fc1_dw": "dense_dw_mfma_kernel<8>",
                    "fc1_fwd": "linear_fwd_mfma_kernel", "fc1_dx": "linear_bwd_dx_mfma_kernel"}},
    "split": {"lstm_fwd": {"enc": "lstm_fwd_pair_sp_kernel<128, 2", "dec": "lstm_fwd_pair_sp_kernel<128, 1",
                           "enc0": "lstm_fwd_sp_kernel<128, 2, true>", "enc1": "lstm_fwd_sp_kernel<128, 0, true>",
                           "dec0": "lstm_fwd_sp_kernel<128, 1, true>", "dec1": "lstm_fwd_sp_kernel<128, 0, true>"},
              "lstm_bwd": {"enc0": "lstm_bwd_sp_kernel<128, 1>", "dec0": "lstm_bwd_sp_kernel<128, 1>",
                           "enc1": "lstm_bwd_dx_pair_sp_kernel<128, 0", "dec1": "lstm_bwd_dx_pair_sp_kernel<128, 2"}},
}


def pmc_traffic(label, precision, meta):
    """HBM bytes per launch of the profiled kernel from a rocprofv3 --pmc summary (profiles/summarize_pmc.py)."""
    kind, _, inst = label.partition(".")
    want = PMC_KERNELS.get(precision, {}).get(kind, {}).get(inst)
    if not want:
        return None
    for name, v in meta["kernels"].items():
        if want in name:
            return v["hbm_bytes"]
    return None


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU share of this process (cgroup/affinity aware), capped at the 16 cores a one-GPU box grants."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(sample_windows=128, max_steps=24, budget_s=12.0):
    """The reference CPU path (oracle/torch_cpu.py: ATen LSTM etc.) timed on this box's host cores, bounded sample: about 10 s of CPU
    work (steps of 128 windows until the budget is spent, at most 24)."""
    import torch
    from oracle.torch_cpu import TorchCpuVAE
    from Scripts.Models.temporal_vae import TemporalVAE
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    ref = TemporalVAE(D, Z, H, L, 0.3)
    m = TorchCpuVAE({k: v.detach() for k, v in ref.state_dict().items()}, L, True, dropout=0.3)
    opt = m.make_optimizer(1e-3, 1e-5)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(sample_windows, T, D, generator=g)
    eps = torch.randn(sample_windows, Z, generator=g)
    tw = time.perf_counter()
    m.train_step(opt, x, eps, KL_W)                       # warm-up
    log(f"cpu baseline warm-up step: {time.perf_counter() - tw:.2f}s on {cores} threads")
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps and (steps == 0 or time.perf_counter() - t0 < budget_s):
        m.train_step(opt, x, eps, KL_W)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": round(sample_windows * steps / dt, 2), "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{steps} train steps of {sample_windows} windows (T={T}, D={D}) after 1 warm-up, torch {torch.__version__} CPU fp32, "
                      f"{cores} threads; same step semantics as the GPU leg"}


def cnn_alg_flops(name, B):
    """Algorithmic FLOPs of one launch group of the CNN step (contractions only) at 2 x T x D windows: SURVEY.md section 8d, CNN row."""
    H0, W0, H1, W1 = T, D, T // 2, D // 2
    F = 32 * (T // 4) * (D // 4)
    conv1, conv2, fc1, fc2 = B * H0 * W0 * 16 * 2 * 9 * 2, B * H1 * W1 * 32 * 16 * 9 * 2, B * F * 128 * 2, B * 128 * 2 * 2
    return {"cnn.conv1_fwd": conv1, "cnn.conv2_fwd": conv2, "cnn.fc1_fwd": fc1, "cnn.fc2_fwd": fc2, "cnn.conv1_dw": conv1, "cnn.conv2_dw": conv2,
            "cnn.conv2_dx": conv2, "cnn.fc1_dw": fc1, "cnn.fc1_dx": fc1, "cnn.fc2_dw": fc2, "cnn.fc2_dx": fc2,
            "cnn.bwd_fused2": 2 * conv2, "cnn.bwd_fused1": conv1}.get(name, 0)


def cnn_cpu_baseline(budget_s=6.0, max_steps=120):
    """The reference CNN step on this box's host cores (oracle/torch_cpu_cnn.py: ATen conv2d / batch_norm / max_pool2d / linear / Adam)."""
    import torch
    from oracle.torch_cpu_cnn import TorchCpuCNN
    from Scripts.Models.cnn_model import CNN
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    ref = CNN(2, 2, 0.5, seq_len=T, num_features=D)
    m = TorchCpuCNN({k: v.detach() for k, v in ref.state_dict().items()}, dropout=0.5)
    opt = m.make_optimizer(1e-4, 5e-5)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B_PER_GPU, 2, T, D, generator=g)
    y = torch.randint(0, 2, (B_PER_GPU,), generator=g)
    m.train_step(opt, x, y)                                  # warm-up
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps and (steps == 0 or time.perf_counter() - t0 < budget_s):
        m.train_step(opt, x, y)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": round(B_PER_GPU * steps / dt, 2), "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{steps} train steps of {B_PER_GPU} windows (2 x {T} x {D}) after 1 warm-up, torch {torch.__version__} CPU fp32, {cores} threads; "
                      f"same step semantics as the GPU leg (05_train_cnn.py:270-276)"}


def cnn_workload(args):
    """`--workload cnn`: the CNN line on its own (same contract as the headline)."""
    if args.gpus != 1:
        raise SystemExit("--workload cnn is a single-GPU line")
    print(json.dumps(cnn_measure(args.steps, args.warmup, args.profile_steps, not args.no_cpu_baseline)))


def cnn_measure(steps, warmup, profile_steps, with_cpu_baseline):
    """Secondary line (BASELINE.json configs[2]): train step of the 4DOF fault classifier (window + squared-residual channels, 2 x 256 x 4)
    at 512 windows per step, fp32 (every product on the exact-fp32 MFMA), plus the eval forward.  One GPU.  Same contract as the
    headline: roofline of the dominant kernel from HIP events in this run, cpu_baseline on the host cores."""
    import torch
    from Scripts.Models.cnn_model import CNN
    from shm_hip._capi import lib, check
    from shm_hip.trainer import CNNTrainer

    class _A:
        pass
    args = _A()
    args.steps, args.warmup, args.profile_steps, args.no_cpu_baseline = steps, warmup, profile_steps, not with_cpu_baseline
    torch.manual_seed(42)
    m = CNN(2, 2, 0.5, seq_len=T, num_features=D).cuda().train()
    tr = CNNTrainer(m, lr=1e-4, weight_decay=5e-5)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(16 * B_PER_GPU, 2, T, D, generator=g).cuda()
    y = torch.randint(0, 2, (16 * B_PER_GPU,), generator=g).cuda()

    def run(k0, k):
        for i in range(k0, k0 + k):
            b = (i % 16) * B_PER_GPU
            tr.step(x[b:b + B_PER_GPU], y[b:b + B_PER_GPU])
    run(0, args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roof = None
    if args.profile_steps > 0:
        check(lib.shm_profile_begin(), "shm_profile_begin")
        run(args.warmup + args.steps, max(args.profile_steps, 4))
        buf = C.create_string_buffer(1 << 16)
        check(lib.shm_profile_end(buf, len(buf)), "shm_profile_end")
        prof = json.loads(buf.value.decode())
        per = {k: v["ms"] / v["calls"] for k, v in prof.items() if k.startswith("cnn.")}
        log("cnn kernel profile: " + json.dumps({k: round(v, 4) for k, v in sorted(per.items())}))
        dom = max((k for k in per if cnn_alg_flops(k, B_PER_GPU) > 0), key=lambda k: per[k])
        fl = cnn_alg_flops(dom, B_PER_GPU)
        ach = fl / (per[dom] * 1e-3) / 1e12
        traffic, why = pmc_traffic_checked(dom, "cnn")
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": PEAK_TFLOPS["f32"], "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS["f32"], 5),
                "traffic": traffic, "traffic_note": why, "avg_launch_ms": round(per[dom], 4), "alg_flops_per_launch": fl,
                "launch_groups_per_step": len(per), "kernels_ms": {k: round(v, 4) for k, v in sorted(per.items())}}
    m.eval()
    with torch.no_grad():
        for _ in range(3):
            m(x[:B_PER_GPU])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            m(x[(i % 16) * B_PER_GPU:(i % 16 + 1) * B_PER_GPU])
        torch.cuda.synchronize()
        de = time.perf_counter() - t1
    step_flops = 3 * sum(cnn_alg_flops(k, 1) for k in ("cnn.conv1_fwd", "cnn.conv2_fwd", "cnn.fc1_fwd", "cnn.fc2_fwd"))
    value = B_PER_GPU * args.steps / dt
    out = {"metric": "windows/sec CNN train (4DOF classifier, 2x256x4 windows, bs512)", "value": round(value, 2),
           "unit": "windows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "4DOF CNN train step: conv3x3(2->16)+BN+ReLU+pool, conv3x3(16->32)+BN+ReLU+pool, fc 2048->128 (+dropout 0.5), "
                                  "fc 128->2, cross-entropy, Adam(1e-4, wd 5e-5); 512 windows of 2 x 256 x 4 per step", "global_batch": B_PER_GPU},
           "alg_flops_per_window_step": step_flops, "whole_step_mfma_frac": round(value * step_flops / (PEAK_TFLOPS["f32"] * 1e12), 5),
           "roofline": roof,
           "eval_forward": {"value": round(B_PER_GPU * args.steps / de, 2), "unit": "windows/s", "ms_per_batch": round(1e3 * de / args.steps, 4)}}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cnn_cpu_baseline()
    return out


def kernel_src_hashes():
    """sha256 of the kernel sources a PMC summary was collected on / the sources of this tree (profiles/summarize_pmc.py stores the former)."""
    import hashlib
    d = ROOT / "hybrid-vae-cnn-for-shm_amd" / "csrc"
    return {f.name: hashlib.sha256(f.read_bytes()).hexdigest()[:16] for f in sorted(d.glob("*.h*"))}


def pmc_traffic_checked(label, precision):
    """(bytes per launch or None, reason).  The committed PMC summary is only quoted while the kernel sources it was collected on are the
    sources of this tree: the summary carries their hashes; a mismatch (or a summary without hashes) yields None instead of a stale figure."""
    summary = PMC_SUMMARIES.get(precision)
    if summary is None or not summary.exists():
        return None, "no PMC summary committed for this precision"
    meta = json.loads(summary.read_text())
    have = meta.get("kernel_src_sha256")
    if not have:
        return None, f"{summary.name} carries no source hashes (collected before they were recorded): not quoted"
    now = kernel_src_hashes()
    changed = sorted(k for k in set(have) | set(now) if have.get(k) != now.get(k))
    if changed:
        return None, f"{summary.name} is stale: {', '.join(changed)} changed since it was collected"
    v = pmc_traffic(label, precision, meta)
    return v, (None if v is not None else "kernel not in the summary")


def measure(precision, data, dev, world, rank, steps, warmup, profile_steps, dist):
    """One leg: `warmup` untimed + `steps` timed train steps of the bench workload on `precision`, barrier + synchronize on both sides,
    then (outside the timed region) `profile_steps` steps with per-kernel HIP events for the roofline object."""
    import torch
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip._capi import lib, check
    from shm_hip.trainer import VAETrainer
    torch.manual_seed(42)
    model = TemporalVAE(D, Z, H, L, 0.3, precision=precision).to(dev).train()
    dt_key = DTYPE_KEY[precision]
    trainer = VAETrainer(model, lr=1e-3, weight_decay=1e-5, max_norm=2.0, seed=42)
    n_batches = data.size(0) // B_PER_GPU
    marks = []                         # one event per timed step (enqueued, never waited on inside the loop): per-step spread afterwards

    def run(k0, k, mark=False):
        for i in range(k0, k0 + k):
            b = i % n_batches
            trainer.step(data[b * B_PER_GPU:(b + 1) * B_PER_GPU], KL_W)
            if mark:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)

    def barrier():
        if world > 1:
            dist.barrier()

    log(f"[{precision}] rank {rank}/{world}: {warmup} warm-up steps")
    tw = time.perf_counter()
    run(0, warmup)
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    log(f"[{precision}] warm-up done in {time.perf_counter() - tw:.2f}s; timing {steps} steps")
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    t0 = time.perf_counter()
    run(warmup, steps, mark=True)
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = sorted(a.elapsed_time(b2) for a, b2 in zip([ev0] + marks[:-1], marks))      # GPU time between consecutive step ends, ms
    pct = lambda f: round(per_step[min(len(per_step) - 1, int(f * len(per_step)))], 4)
    step_ms = {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9)} if per_step else None
    log(f"[{precision}] timed region: {dt:.3f}s for {steps} steps")
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    stats = trainer.stats.cpu().tolist()
    trainer.check()                        # sticky hand-off status word: raises if ANY step of this leg ran with a timed-out hand-off
    trainer.engine.check_status()

    # per-kernel durations with HIP events on the launch stream (outside the timed region)
    # EVERY rank runs these steps (a step contains the gradient all-reduce, a collective); only rank 0 records events
    roof = None
    if profile_steps > 0 and rank != 0:
        run(warmup + steps, profile_steps)
        torch.cuda.synchronize()
    if rank == 0 and profile_steps > 0:
        check(lib.shm_profile_begin(), "shm_profile_begin")
        run(warmup + steps, profile_steps)
        buf = C.create_string_buffer(1 << 16)
        check(lib.shm_profile_end(buf, len(buf)), "shm_profile_end")
        prof = json.loads(buf.value.decode())
        log(f"[{precision}] kernel profile: " + json.dumps(prof))
        per = {k: v["ms"] / v["calls"] for k, v in prof.items()}
        # split path: the weight-gradient GEMMs run on a side stream BESIDE the next recurrence, so their event-to-event time includes
        # the time they share the chip with it — not a kernel duration; the dominant kernel is taken from the main-stream launches
        cand = [k for k in per if kernel_alg_flops(k, B_PER_GPU, precision) > 0 and not (precision == "split" and k.startswith("dw_gemm."))]
        dom = max(cand, key=lambda k: per[k])
        fl = kernel_alg_flops(dom, B_PER_GPU, precision)
        if precision == "split" and dom in ("lstm_bwd.enc1", "lstm_bwd.dec1") and f"dx_gemm.{dom[-4:]}" not in per:
            fl += kernel_alg_flops(f"dx_gemm.{dom[-4:]}", B_PER_GPU, precision)     # the top layer's launch also forms dx of the layer below
        achieved = fl / (per[dom] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[dt_key]
        traffic, why = pmc_traffic_checked(dom, precision)
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 5), "traffic": traffic, "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/)",
                "avg_launch_ms": round(per[dom], 4),
                "hbm_achieved_GBps": (round(traffic / (per[dom] * 1e-3) / 1e9, 1) if traffic else None),
                "alg_flops_per_launch": fl, "kernels_ms": {k: round(v, 4) for k, v in sorted(per.items())}}
        if why:
            roof["traffic_note"] = why
    barrier()
    value = world * B_PER_GPU * steps / dt
    step_flops = 3 * alg_flops_forward_per_window()
    out = {"value": round(value, 2), "unit": "windows/s", "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps, "warmup": warmup, "dtype": dt_key,
           "whole_step_mfma_frac": round(value * step_flops / (world * PEAK_TFLOPS[dt_key] * 1e12), 5), "step_ms": step_ms,
           "final_stats": {"loss": stats[0], "recon": stats[1], "kl": stats[2], "grad_norm": stats[3]}, "roofline": roof}
    del trainer, model
    torch.cuda.empty_cache()
    return out


DTYPE_KEY = {"bf16": "bf16", "fp32": "f32", "split": "bf16x2"}
PRECISION_NOTE = {
    "bf16": "bf16 MFMA operands, fp32 accumulate / cell state / master weights (BASELINE.json configs[1]); parity at its own documented bars",
    "split": "every MFMA operand a bf16 pair hi + lo (16 significant bits), products hi*hi + hi*lo + lo*hi in fp32, fp32 state: holds the "
             "reference's fp32 results to ~1e-6 (north_star's 1e-4 bar) — the parity-grade fast path",
    "fp32": "exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), bitwise an fmaf chain: the arithmetic reference of the build, slow",
}


def inference_leg(data, dev, precisions=("bf16", "split")):
    """Scoring side of the same model (04_vae_thresholding.py:232-283 / 06_test_full_pipeline.py: eval-mode forward, per-window mean
    squared reconstruction error) on the resident synthetic windows, 512 per call: windows/s per precision.  Not the headline metric."""
    import torch
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip.pipeline import score_windows
    out = {}
    for prec in precisions:
        torch.manual_seed(42)
        vae = TemporalVAE(D, Z, H, L, 0.3, precision=prec).to(dev).eval()
        score_windows(vae, data[:2 * B_PER_GPU], B_PER_GPU)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 4
        for _ in range(reps):
            score_windows(vae, data, B_PER_GPU)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[prec] = {"value": round(data.size(0) / dt, 1), "unit": "windows/s", "ms_per_512": round(dt * 1e3 * B_PER_GPU / data.size(0), 4),
                     "windows": int(data.size(0)), "what": "score_windows: eval forward + per-window MSE, ε drawn on the device"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (SURVEY.md section 8d protocol: 20 warm-up + 200 timed)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=4, help="steps profiled per leg with per-launch HIP events, right after its timed region")
    ap.add_argument("--precision", choices=["bf16", "fp32", "split"], default="bf16",
                    help="headline leg.  bf16 (default): BASELINE.json configs[1]; split: bf16 hi+lo operand pairs, the 1e-4 parity-grade fast "
                         "path; fp32: exact-f32 MFMA")
    ap.add_argument("--no-parity-leg", action="store_true", help="skip the parity-grade (split) leg that follows a bf16 headline at N = 1")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the short exact-fp32 measurement (N = 1 only)")
    ap.add_argument("--no-inference-leg", action="store_true", help="skip the scoring (inference) measurement (N = 1 only)")
    ap.add_argument("--no-cnn-leg", action="store_true", help="skip the CNN train-step object of the default line (N = 1 only)")
    ap.add_argument("--workload", choices=["vae", "cnn"], default="vae",
                    help="vae (default): the headline LSTM-VAE train step (BASELINE.json metric); cnn: the 4DOF classifier's train step "
                         "(BASELINE.json configs[2], secondary line, N = 1 only)")
    args = ap.parse_args()
    if args.workload == "cnn":
        return cnn_workload(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 through torch.distributed.run")
    # Rehearsal knobs for a ONE-GPU box (never set by the driver): SHM_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and
    # SHM_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device), so the N > 1 control flow can be exercised
    # (tests/test_gpu_multirank.py does).
    if os.environ.get("SHM_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("SHM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # S1 synthetic windows, resident on the device: no H2D inside the timed region
    n_resident = 16 * B_PER_GPU
    gen = torch.Generator().manual_seed(42 + rank)
    data = torch.randn(n_resident, T, D, generator=gen).to(dev)

    head = measure(args.precision, data, dev, world, rank, args.steps, args.warmup, args.profile_steps, dist)
    if rank == 0:
        out = {
            "metric": "windows/sec LSTM-VAE train (4DOF, seq256x4ch, bs512) at 1/2/4/8 GPU",
            "value": head["value"], "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
            "config": {"workload": "4DOF LSTM-VAE train step: TemporalVAE(D=4,Z=16,H=128,L=2,dropout=0.3), T=256, "
                                   "batch 512 windows/GPU, fwd+loss+BPTT+clip(2.0)+Adam(1e-3,wd 1e-5), kl_w=0.5",
                       "global_batch": world * B_PER_GPU, "seq_len": T, "parallelism": f"dp{world}",
                       "precision": args.precision, "precision_note": PRECISION_NOTE[args.precision]},
            "alg_flops_per_window_step": 3 * alg_flops_forward_per_window(),
            "whole_step_mfma_frac": head["whole_step_mfma_frac"], "step_ms": head["step_ms"], "final_stats": head["final_stats"],
            "roofline": head["roofline"],
        }
    if world == 1 and args.precision == "bf16" and not args.no_parity_leg:
        # the parity-grade fast path (recon / mu / logvar within 1e-4 of the reference, measured ~1e-6) on the same workload, same run,
        # with its own roofline object; at least 50 timed steps
        k = max(50, min(args.steps, 100))
        leg = measure("split", data, dev, 1, 0, k, max(5, min(args.warmup, 10)), args.profile_steps, dist)
        leg["precision"] = "split"; leg["precision_note"] = PRECISION_NOTE["split"]
        out["parity_path"] = leg
    if world == 1 and args.precision != "fp32" and not args.no_fp32_leg:
        leg = measure("fp32", data, dev, 1, 0, 6, 2, 0, dist)
        leg["precision"] = "fp32"; leg["precision_note"] = PRECISION_NOTE["fp32"]
        out["fp32_exact_path"] = leg
    if world == 1 and not args.no_inference_leg:
        out["inference"] = inference_leg(data, dev)
    if world == 1 and not args.no_cnn_leg:
        # the classifier's train step (BASELINE.json configs[2]) in the same driver-run line: its own metric / roofline / cpu_baseline
        # object, short CPU sample (the headline's cpu_baseline above keeps the larger share of the CPU budget)
        del data
        torch.cuda.empty_cache()
        out["cnn_train"] = cnn_measure(min(args.steps, 200), min(args.warmup, 20), args.profile_steps, not args.no_cpu_baseline)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
