"""Throughput of the device data generators: runs/s and GB/s of the response table for the Newmark kernel, the
excitation smoothing and a sensor-fault pass, 1001-step 4-DOF runs.   python tools/sim_timing.py [R ...]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hybrid-vae-cnn-for-shm_amd"))
from shm_hip import simulate as sim   # noqa: E402


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [10, 1024, 16384, 65536]
    cfg = sim.default_system_config()
    for R in sizes:
        cfgs = [cfg] * R
        g = torch.Generator(device="cuda").manual_seed(1)
        base = torch.randn(R, cfg.steps, 4, generator=g, dtype=torch.float64, device="cuda") * 50.0
        t_s = timed(lambda: sim.smooth_force(base, 50))
        force = sim.smooth_force(base, 50)
        del base
        import ctypes as C
        from shm_hip._capi import check, lib
        sys_d = torch.from_numpy(sim.pack_systems(cfgs[:1])).cuda().repeat(R, 1).contiguous()
        for dt_, nm in ((torch.float32, "fp32"), (torch.float64, "fp64")):
            out = torch.empty(R, cfg.steps, 12, dtype=dt_, device="cuda")
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            run = lambda: check(lib.shm_sim_newmark(C.c_void_p(sys_d.data_ptr()), C.c_void_p(force.data_ptr()), 0, R, cfg.steps, 4, C.c_double(0.01),
                                                    C.c_double(0.25), C.c_double(0.5), C.c_void_p(out.data_ptr()), int(dt_ == torch.float32), s), "nm")
            t = timed(run)
            byts = out.numel() * out.element_size() + force.numel() * 4
            print(f"R={R:6d} newmark {nm}: {t * 1e3:8.3f} ms  {R / t:12.0f} runs/s  {byts / t / 1e9:8.1f} GB/s (table + excitation)")
            if dt_ == torch.float32:
                z = torch.randn(R, 3, cfg.steps, dtype=torch.float64, device="cuda")
                tf = timed(lambda: sim.inject_sensor_fault(out, "noise", 4, 0.5, z=z))
                print(f"R={R:6d} sensor fault (noise, incl. table copy): {tf * 1e3:8.3f} ms")
                del z
            del out
        print(f"R={R:6d} smoothing: {t_s * 1e3:8.3f} ms  ({R * cfg.steps * 4 * 12 / t_s / 1e9:.1f} GB/s in+out)")


if __name__ == "__main__":
    main()
