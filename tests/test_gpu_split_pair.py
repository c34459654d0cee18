"""Split-precision path, layer-pipelined forward (round 3): lstm_fwd_pair_sp_kernel runs both layers of a stack in ONE launch and
forms layer 1's input projection inside the recurrent workgroup (W_ih as fp16 in LDS + an fp8 residual in registers, four steps per MFMA tile).
Its MFMA sequence per output element is the one of the two-launch form (xproj_sp_kernel + lstm_fwd_sp_kernel), so every result must be
BIT-IDENTICAL to one launch per layer — which is itself pinned to the reference goldens at 1e-4 (tests/test_gpu_vae.py, split legs).
Reference arithmetic: 4DOF/Scripts/Models/temporal_vae.py:51-77 (nn.LSTM stacks), 4DOF/Scripts/03_train_vae.py:260-271 (train step)."""
import numpy as np
import pytest
import torch

from test_gpu_vae import inject_eps

pytestmark = pytest.mark.gpu

SHAPES = ((512, 256, 128), (510, 100, 128), (37, 50, 64), (9, 31, 32), (1, 1, 128), (3, 2, 128), (5, 3, 64), (2, 5, 32), (4, 7, 128),
          (6, 9, 128), (7, 12, 64), (1100, 20, 128))        # ragged last tiles, T % 4 != 0, T < one 4-step group, three ticket groups


@pytest.mark.parametrize("train_mode", [False, True])
def test_split_layer_pipelined_forward_is_bit_identical(train_mode):
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip._capi import lib
    from shm_hip.trainer import VAETrainer
    for (B, T, H) in SHAPES:
        torch.manual_seed(11)
        m = TemporalVAE(4, 16, H, 2, 0.3, precision="split").cuda()
        gen = torch.Generator().manual_seed(B)
        x = torch.randn(B, T, 4, generator=gen).cuda()
        eps = torch.randn(B, 16, generator=gen).cuda()
        ref = None
        for on in (0, 1, 1, 1):
            lib.shm_set_pair_launch(on)
            try:
                if train_mode:
                    torch.manual_seed(11)
                    m2 = TemporalVAE(4, 16, H, 2, 0.3, precision="split").cuda().train()
                    tr = VAETrainer(m2, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
                    stats = tr.step(x, 0.3, eps=eps, drop_seed=99)
                    out = (stats.clone(), tr.grads.clone())
                    tr.engine.check_status()
                else:
                    m.eval()
                    with torch.no_grad(), inject_eps([eps.cpu()]):
                        out = m(x)
                    m._engine.check_status()
            finally:
                lib.shm_set_pair_launch(1)
            if ref is None:
                ref = out
            else:
                for a, b in zip(ref, out):
                    assert torch.equal(a, b), (B, T, H, on, float((a - b).abs().max()))


def test_split_pipelined_forward_without_dropout_matches_too():
    """dropout = 0 takes the staging path that moves whole 16-byte chunks per plane (the dropout path moves half chunks of both planes)."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip._capi import lib
    from shm_hip.trainer import VAETrainer
    for (B, T, H) in ((64, 40, 128), (10, 23, 64), (5, 8, 32)):
        gen = torch.Generator().manual_seed(7 * B)
        x = torch.randn(B, T, 4, generator=gen).cuda()
        eps = torch.randn(B, 16, generator=gen).cuda()
        ref = None
        for on in (0, 1, 1):
            lib.shm_set_pair_launch(on)
            try:
                torch.manual_seed(3)
                m = TemporalVAE(4, 16, H, 2, 0.0, precision="split").cuda().train()
                tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
                out = (tr.step(x, 0.3, eps=eps).clone(), tr.grads.clone())
                tr.engine.check_status()
            finally:
                lib.shm_set_pair_launch(1)
            if ref is None:
                ref = out
            else:
                for a, b in zip(ref, out):
                    assert torch.equal(a, b), (B, T, H, float((a - b).abs().max()))


def test_split_pipelined_handoff_under_memory_load():
    """The in-launch hand-off must not depend on timing: repeat the pipelined train step while a second stream streams 512 MB copies
    through HBM / L2 (uneven load on the CUs' memory queues) and require bit-identity with the one-launch-per-layer result every time."""
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip._capi import lib
    from shm_hip.trainer import VAETrainer
    B, T, H = 512, 128, 128
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(B, T, 4, generator=gen).cuda()
    eps = torch.randn(B, 16, generator=gen).cuda()

    def one(on):
        lib.shm_set_pair_launch(on)
        try:
            torch.manual_seed(21)
            m = TemporalVAE(4, 16, H, 2, 0.3, precision="split").cuda().train()
            tr = VAETrainer(m, lr=1e-3, weight_decay=1e-5, max_norm=2.0)
            st = tr.step(x, 0.3, eps=eps, drop_seed=5).clone()
            tr.engine.check_status()
            return st, tr.grads.clone()
        finally:
            lib.shm_set_pair_launch(1)

    ref = one(0)
    side = torch.cuda.Stream()
    a = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda")      # 512 MB
    b = torch.empty_like(a)
    for rep in range(10):
        with torch.cuda.stream(side):
            for _ in range(3 + rep % 4):
                b.copy_(a, non_blocking=True)
        got = one(1)
        side.synchronize()
        for u, v in zip(ref, got):
            assert torch.equal(u, v), (rep, float((u - v).abs().max()))


@pytest.mark.parametrize("shape", [(4, 7, 128, 4), (6, 9, 128, 4), (9, 24, 128, 12), (5, 33, 64, 3), (3, 40, 32, 1), (13, 100, 128, 4), (2, 4, 128, 16)])
def test_split_train_step_matches_the_oracle_at_odd_shapes(shape):
    """Round-3 kernels against the fp64 oracle (oracle/vae_oracle.py, pinned to the reference goldens) at shapes around their pipeline depths:
    T % 4 != 0 (the top-layer backward + dx launch falls back to two launches, the fused forward runs a partial last group), T % 4 == 0
    (both pair launches), rows that do not fill a 32-row GEMM stage, H < 128 (general GEMM), D = 12 / 16 (the 16-column third product).
    Bars of the split path everywhere else: loss 1e-4, every gradient tensor within 2e-4 of its max."""
    from oracle.vae_oracle import VAEOracle
    from Scripts.Models.temporal_vae import TemporalVAE
    from shm_hip.trainer import VAETrainer
    B, T, H, D = shape
    torch.manual_seed(1000 + B + T)
    m = TemporalVAE(D, 8, H, 2, 0.0, precision="split").cuda().train()
    p0 = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(B * 7 + T)
    x = torch.randn(B, T, D, generator=g)
    eps = torch.randn(B, 8, generator=g)
    tr = VAETrainer(m, lr=1e-3, weight_decay=0.0, max_norm=2.0)
    stats = tr.step(x.cuda(), 0.25, eps=eps.cuda()).cpu().numpy()
    tr.engine.check_status()
    o = VAEOracle(p0, 2, True)
    o.forward(x.numpy(), eps.numpy())
    losses, grads = o.loss_and_grads(0.25)
    got = dict(zip([k for k, _, _ in tr.engine.layout], tr.engine.split(tr.grads)))
    assert abs(stats[0] - losses["loss"]) < 1e-4 * max(1.0, losses["loss"]), (stats, losses)
    worst = {k: float(np.abs(got[k].cpu().numpy() - v).max() / max(1e-3, np.abs(v).max())) for k, v in grads.items()}
    assert max(worst.values()) < 2e-4, (shape, max(worst, key=worst.get), max(worst.values()))
