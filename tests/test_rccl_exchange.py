"""The RCCL side of the data-parallel path on ONE MI355X: backend "nccl" with world_size 1, GradientExchange(force=True) - the
engine's three-phase deferred reduction, the raw hipEvent_t hand-off (pidm_unet_set_grad_events), the side stream and one
ReduceOp.AVG all-reduce per flat range all run for real; with a single rank the average is the identity, so the gradients must be
BIT-identical to a step without any exchange.  Variants: overlapped (default), PIDM_DP_NO_OVERLAP=1 (collectives after backward on
the caller's stream), PIDM_DP_BUCKETS=1 (one range), the two-tape step of x0_estimation='sample' (falls back to the exchange
after backward), and close() / re-creation of the exchange object (the engine must not keep dangling event handles).
(The N-rank arithmetic is covered on CPU by tests/test_data_parallel.py with 2 gloo ranks.)"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_single_rank():
    import torch.distributed as dist
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29700 + os.getpid() % 200)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    yield dist
    if created:
        dist.destroy_process_group()


def _setup(two_tape):
    from oracle import pidm_oracle as O
    from physicsinformeddiffusionmodels_amd.denoising_utils import DenoisingDiffusion
    from physicsinformeddiffusionmodels_amd.residuals_darcy import ResidualsDarcy
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D
    dev = torch.device("cuda:0")
    m = Unet3D(dim=16, channels=2)
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    m = m.to(dev)
    diff = DenoisingDiffusion(100, dev)
    res = ResidualsDarcy(model=m, fd_acc=2, pixels_per_dim=32, pixels_at_boundary=True, reverse_d1=True, device=dev)
    res.use_ddim_x0 = two_tape
    g = torch.Generator().manual_seed(31)
    x0 = torch.randn(6, 2, 32, 32, generator=g)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = torch.randn(6, 2, 32, 32, generator=g)
    t = torch.tensor([0, 13, 40, 41, 77, 99])
    return m, diff, res, x0.to(dev), eps.to(dev), t.to(dev)


def _step(m, diff, res, x0, eps, t, exchange=None):
    orig = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t.clone()
    torch.randn_like = lambda *a, **k: eps.clone()
    for p in m.parameters():
        p.grad = None
    try:
        loss, *_ = diff.model_estimation_loss(x0, residual_func=res, c_data=1., c_residual=1e-3)
    finally:
        torch.randint, torch.randn_like = orig
    loss.backward()
    if exchange is not None:
        exchange.allreduce()
    torch.cuda.synchronize()
    from physicsinformeddiffusionmodels_amd._engine import get_engine
    return get_engine(m, 32).flat_grad.clone()


@pytest.mark.parametrize("two_tape", [False, True])
def test_forced_single_rank_exchange_is_the_identity(nccl_single_rank, monkeypatch, two_tape):
    from physicsinformeddiffusionmodels_amd.parallel import GradientExchange
    m, diff, res, x0, eps, t = _setup(two_tape)
    ref = _step(m, diff, res, x0, eps, t)                      # no exchange object: one deferred reduction, no events
    assert torch.isfinite(ref).all() and ref.abs().max().item() > 0
    # (default on GPUs: the library's own communicator, negotiated and self-checked; native=False / PIDM_DP_NATIVE=0: torch.distributed)
    for env, want_overlap, want_ranges, native in (({}, not two_tape, [1, 1, 2], None), ({}, not two_tape, [1, 1, 2], False),
                                                   ({"PIDM_DP_NO_OVERLAP": "1"}, False, [1, 1, 2], None),
                                                   ({"PIDM_DP_BUCKETS": "1"}, not two_tape, [1], None),
                                                   ({"PIDM_DP_NATIVE": "0"}, not two_tape, [1, 1, 2], None)):
        for k in ("PIDM_DP_NO_OVERLAP", "PIDM_DP_BUCKETS", "PIDM_DP_NATIVE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ex = GradientExchange(m, image_size=32, diffusion=diff, force=True, native=native)
        assert ex.active and ex.world == 1 and [len(r) for r in ex.ranges] == want_ranges
        if native is False or env.get("PIDM_DP_NATIVE") == "0":
            assert ex.native is None and ex.collective == "torch.distributed.all_reduce (nccl)"
        else:
            assert ex.native is not None and ex.collective.startswith("pidm_allreduce_f32") and ex.collective_note is None
        ex.measure = True
        for _ in range(5):                                      # steady state: from the third step on the backward is a replayed hipGraph
                                                                # cut at the phase events (recorded for real between the segments)
            got = _step(m, diff, res, x0, eps, t, ex)
            assert ex.last_overlapped == want_overlap, env
            assert torch.equal(got, ref), env                   # AVG over one rank: bit-identical gradients
        ms = ex.exchange_ms()
        assert ms is not None and 0.0 < ms < 1000.0
        ex.close()
        with pytest.raises(RuntimeError):
            ex.allreduce()
    # after close() the engine is back to a single reduction without events: plain steps still work and agree
    assert torch.equal(_step(m, diff, res, x0, eps, t), ref)


def test_exchange_object_lifetime(nccl_single_rank):
    """Dropping a GradientExchange must detach its events from the engine (the engine held raw hipEvent_t handles of
    torch events owned by the exchange object); a newer exchange of the same model is not disturbed by the older one's __del__."""
    import gc
    from physicsinformeddiffusionmodels_amd.parallel import GradientExchange
    m, diff, res, x0, eps, t = _setup(False)
    ref = _step(m, diff, res, x0, eps, t)
    ex1 = GradientExchange(m, image_size=32, force=True)
    assert torch.equal(_step(m, diff, res, x0, eps, t, ex1), ref)
    ex2 = GradientExchange(m, image_size=32, force=True)        # takes the engine over
    del ex1
    gc.collect()
    assert torch.equal(_step(m, diff, res, x0, eps, t, ex2), ref) and ex2.last_overlapped
    del ex2
    gc.collect()
    assert torch.equal(_step(m, diff, res, x0, eps, t), ref)    # no dangling event handles: backward records nothing


def test_c_abi_communicator_single_rank(nccl_single_rank):
    """include/pidm.h pidm_comm_unique_id / pidm_comm_init / pidm_allreduce_f32 / pidm_comm_destroy: RCCL bound inside the library
    (no torch collective on the data path).  One rank: mean and sum are the identity, enqueued on the current (non-default) stream."""
    import ctypes as C
    from physicsinformeddiffusionmodels_amd._lib import get_lib, vp
    lib = get_lib()
    ident = (C.c_ubyte * 128)()
    lib.check(lib.pidm_comm_unique_id(ident), "pidm_comm_unique_id")
    assert any(ident)
    comm = vp()
    lib.check(lib.pidm_comm_init(0, 1, bytes(ident), C.byref(comm)), "pidm_comm_init")
    assert comm.value
    x = torch.randn(1 << 20, device="cuda:0")
    ref = x.clone()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for avg in (1, 0):
            lib.check(lib.pidm_allreduce_f32(comm, vp(x.data_ptr()), x.numel(), avg, vp(st.cuda_stream)), "pidm_allreduce_f32")
    st.synchronize()
    assert torch.equal(x, ref)
    assert lib.pidm_allreduce_f32(None, vp(x.data_ptr()), 4, 1, None) != 0            # null communicator: an error, not a crash
    assert lib.pidm_comm_init(3, 2, bytes(ident), C.byref(vp())) != 0                # rank outside the world
    lib.check(lib.pidm_comm_destroy(comm), "pidm_comm_destroy")


def test_gradient_exchange_over_the_c_abi_communicator(nccl_single_rank):
    """GradientExchange(native=True): the three overlapped range all-reduces go through pidm_allreduce_f32 on the side stream
    instead of torch.distributed.all_reduce - same bit-identical result with one rank, id hand-off over the process group."""
    from physicsinformeddiffusionmodels_amd.parallel import GradientExchange
    m, diff, res, x0, eps, t = _setup(False)
    ref = _step(m, diff, res, x0, eps, t)
    ex = GradientExchange(m, image_size=32, diffusion=diff, force=True, native=True)
    assert ex.native is not None and ex.native.world == 1
    for _ in range(4):
        assert torch.equal(_step(m, diff, res, x0, eps, t, ex), ref) and ex.last_overlapped
    ex.close()
    assert ex.native is None
    assert torch.equal(_step(m, diff, res, x0, eps, t), ref)
