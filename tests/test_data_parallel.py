"""Data-parallel equivalence: 2 ranks (gloo, CPU, host-emulated kernels) each run the training step on half of the
batch with injected (t, eps); after allreduce_gradients the gradients equal those of the single-process step on
the global batch (per-rank loss = mean over the shard, gradients averaged)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(lib):
    from oracle import pidm_oracle as O
    from physicsinformeddiffusionmodels_amd.denoising_utils import DenoisingDiffusion
    from physicsinformeddiffusionmodels_amd.residuals_darcy import ResidualsDarcy
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D
    dev = torch.device("cpu")
    m = Unet3D(dim=8, channels=2)
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    m._pidm_lib = lib
    diff = DenoisingDiffusion(100, dev, lib=lib)
    res = ResidualsDarcy(model=m, fd_acc=2, pixels_per_dim=16, pixels_at_boundary=True, reverse_d1=True, device=dev, lib=lib)
    return m, diff, res


def _inputs():
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(4, 2, 16, 16, generator=g)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = torch.randn(4, 2, 16, 16, generator=g)
    t = torch.tensor([2, 41, 77, 99])
    return x0, eps, t


def _step(m, diff, res, x0, eps, t):
    orig = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t.clone()
    torch.randn_like = lambda *a, **k: eps.clone()
    try:
        loss, *_ = diff.model_estimation_loss(x0, residual_func=res, c_data=1., c_residual=1e-3)
    finally:
        torch.randint, torch.randn_like = orig
    loss.backward()
    return loss.item()


def _worker(rank, world, port, outdir):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HIPEMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from physicsinformeddiffusionmodels_amd.parallel import allreduce_gradients, shard_batch
    from tests.emu_util import emu_lib
    m, diff, res = _setup(emu_lib())
    x0, eps, t = _inputs()
    loss = _step(m, diff, res, shard_batch(x0, rank, world), shard_batch(eps, rank, world), shard_batch(t, rank, world))
    allreduce_gradients(m, world)
    eng = next(iter(m.__dict__["_engines"].values()))
    np.save(os.path.join(outdir, f"grad_{rank}.npy"), eng.flat_grad.numpy())
    np.save(os.path.join(outdir, f"loss_{rank}.npy"), np.array(loss))
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_step_equals_global_batch_step(tmp_path):
    from tests.emu_util import emu_lib
    lib = emu_lib()   # builds the emulated library once, before forking workers
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0 = np.load(tmp_path / "grad_0.npy")
    g1 = np.load(tmp_path / "grad_1.npy")
    np.testing.assert_array_equal(g0, g1)          # every rank holds the same averaged gradient
    m, diff, res = _setup(lib)
    x0, eps, t = _inputs()
    loss = _step(m, diff, res, x0, eps, t)
    eng = next(iter(m.__dict__["_engines"].values()))
    ref = eng.flat_grad.numpy()
    l0, l1 = float(np.load(tmp_path / "loss_0.npy")), float(np.load(tmp_path / "loss_1.npy"))
    assert abs(0.5 * (l0 + l1) - loss) < 1e-5 * abs(loss)
    # per-tensor comparison: relative to each tensor's own scale, with a floor for ~zero gradients
    gmax = np.abs(ref).max()
    off = 0
    for name, ne in zip(eng.names, eng.numels):
        a, b = g0[off:off + ne], ref[off:off + ne]
        off += ne
        assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max() + 1e-6 * gmax, name
