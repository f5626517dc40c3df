"""hipGraph replay of the engine's forward / backward passes (csrc/unet_engine.hip): from the third call with the same (batch,
mode, workspace, bound parameters, knobs) a pass is ONE hipGraphLaunch per segment instead of ~200 kernel launches.  The replayed
passes must be bit-identical to launch-by-launch execution (PIDM_GRAPH=0) on CHANGING inputs, across interleaved inference
forwards, other batch sizes, parameter re-binding (EMA swap in / out), the three-phase gradient reduction of the data-parallel
overlap, the conditioning branch and the two-tape step.  `backend` = host emulator (captures are recorded closures there) or the
real gfx950 library (-m gpu)."""
import ctypes as C

import pytest
import torch

from physicsinformeddiffusionmodels_amd._lib import reload_knobs

from oracle import pidm_oracle as O
from physicsinformeddiffusionmodels_amd._engine import get_engine
from physicsinformeddiffusionmodels_amd.unet_model import Unet3D


@pytest.fixture(autouse=True)
def _same_split_plan(monkeypatch):
    """Bit-identity of replay and launch-by-launch execution is a statement about the SAME arithmetic.  A linear (replayed) backward
    queues its weight-gradient problems for grouped launches with a quarter of the splits per problem by default
    (PIDM_WGRAD_GROUP_SPLITDIV=4: another fixed summation order - bit-identical run to run, tests/test_unet_engine.py::
    test_grouped_weight_gradients), the launch-by-launch backward keeps its weight gradients on the side stream with the full
    split count; with the divisor at 1 both forms run the same split plan and must agree to the bit."""
    monkeypatch.setenv("PIDM_WGRAD_GROUP_SPLITDIV", "1")


def _counts(L):
    a = (C.c_longlong * 4)()
    L.check(L.pidm_debug_launch_counts(a))
    return {"eager": a[0], "graph_launches": a[1], "graph_kernels": a[2], "captures": a[3]}


def _model(dev, L, **kw):
    kw.setdefault("dim_mults", (1, 2))          # two levels: the replay mechanics do not depend on the depth
    kw.setdefault("dim", 8)
    m = Unet3D(channels=2, **kw)
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    m = m.to(dev)
    m._pidm_lib = L if dev.type == "cpu" else None
    return m


def _inputs(dev, n, B=2, P=16, seed=11):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(B, P * P, 2, generator=g).to(dev), torch.randint(0, 100, (B,), generator=g).to(dev),
             torch.randn(B, 2, P, P, generator=g).to(dev)) for _ in range(n)]


def _train_step(m, x, t, w, cond=None):
    for p in m.parameters():
        p.grad = None
    out = m(x, t) if cond is None else m(x, t, cond=cond)
    (out * w).sum().backward()
    eng = get_engine(m, 16, m._pidm_lib)
    return out.detach().clone(), eng.flat_grad.clone()


def test_replayed_steps_equal_eager_steps(backend, monkeypatch):
    L, dev = backend
    data = _inputs(dev, 5)
    monkeypatch.setenv("PIDM_GRAPH", "0")
    m0 = _model(dev, L)
    ref = [_train_step(m0, *d) for d in data]
    monkeypatch.delenv("PIDM_GRAPH")
    m1 = _model(dev, L)
    c0 = _counts(L)
    got = [_train_step(m1, *d) for d in data]
    c1 = _counts(L)
    assert c1["captures"] - c0["captures"] == 2                       # one forward graph, one backward graph
    assert c1["graph_launches"] - c0["graph_launches"] >= 2 * 2        # steps 3.. are replays (the capture call launches too)
    for (o_ref, g_ref), (o, g) in zip(ref, got):
        assert torch.equal(o, o_ref) and torch.equal(g, g_ref)
    # the last steps enqueue no kernel of their own: everything went through the graphs (+ 5 plain device copies per step)
    c2 = _counts(L)
    _train_step(m1, *data[0])
    c3 = _counts(L)
    assert c3["eager"] == c2["eager"] and c3["graph_launches"] - c2["graph_launches"] == 2
    assert c3["graph_kernels"] - c2["graph_kernels"] > 100


def test_replay_survives_interleaved_passes_and_rebinding(backend):
    """Between replayed training steps: an inference forward of the same batch, a training step of ANOTHER batch size (its own
    workspace plan, overwrites the host-side tape record), and a swap of the parameter storage (what EMA.ema()/restore() do)."""
    L, dev = backend
    m = _model(dev, L)
    ref_m = _model(dev, L)
    data = _inputs(dev, 6)
    other = _inputs(dev, 2, B=3, seed=12)
    import os
    for i, d in enumerate(data):
        os.environ["PIDM_GRAPH"] = "0"
        reload_knobs()
        try:
            o_ref, g_ref = _train_step(ref_m, *d)
        finally:
            del os.environ["PIDM_GRAPH"]
            reload_knobs()
        o, g = _train_step(m, *d)
        assert torch.equal(o, o_ref) and torch.equal(g, g_ref), i
        if i == 2:
            with torch.no_grad():
                m(d[0], d[1])
        if i == 3:
            _train_step(m, *other[0])
            _train_step(m, *other[1])
        if i == 4:
            # pointer flip of every parameter to a copy and back (same values): the engine re-binds twice
            backup = {k: p.data for k, p in m.named_parameters()}
            for p in m.parameters():
                p.data = p.data.clone()
            with torch.no_grad():
                o_sw = m(d[0], d[1])
            for k, p in m.named_parameters():
                p.data = backup[k]
            with torch.no_grad():
                assert torch.equal(o_sw, m(d[0], d[1]))          # same values behind both pointer sets


def test_changed_weights_are_seen_by_the_replayed_forward(backend):
    """The weight re-pack is part of the forward graph: an optimizer step between two replays changes the result exactly as it
    does launch by launch."""
    L, dev = backend
    import os
    ma, mb = _model(dev, L), _model(dev, L)
    data = _inputs(dev, 4)
    for i, d in enumerate(data):
        os.environ["PIDM_GRAPH"] = "0"
        reload_knobs()
        try:
            o_ref, g_ref = _train_step(ma, *d)
        finally:
            del os.environ["PIDM_GRAPH"]
            reload_knobs()
        o, g = _train_step(mb, *d)
        assert torch.equal(o, o_ref) and torch.equal(g, g_ref), i
        with torch.no_grad():
            for p, q in zip(ma.parameters(), mb.parameters()):
                if p.grad is not None:
                    p.add_(p.grad, alpha=-1e-3)
                    q.add_(q.grad, alpha=-1e-3)


def test_three_phase_backward_is_cut_into_segments(backend):
    """With the data-parallel phase split on (events or not) the replayed backward gives the same gradients; with phase events the
    graph is cut where they are recorded (external events cannot be recorded from inside a capture)."""
    L, dev = backend
    m = _model(dev, L)
    eng = get_engine(m, 16, m._pidm_lib)
    data = _inputs(dev, 5)
    ref = [_train_step(m, *d)[1] for d in data[:1]]
    events = None
    if dev.type == "cuda":
        events = [torch.cuda.Event() for _ in range(3)]
        for e in events:
            e.record()
        handles = (C.c_void_p * 3)(*[C.c_void_p(e.cuda_event) for e in events])
        L.check(L.pidm_unet_set_grad_events(eng.handle, 3, handles))
    else:
        L.check(L.pidm_unet_set_grad_events(eng.handle, 3, None))
    try:
        c0 = _counts(L)
        for _ in range(4):
            g = _train_step(m, *data[0])[1]
            assert torch.equal(g, ref[0])
        c1 = _counts(L)
        assert c1["captures"] - c0["captures"] >= 1
        if events is not None:
            # forward: 1 launch; backward: 3 segments per replayed step
            per_step = (c1["graph_launches"] - c0["graph_launches"]) / 2
            assert per_step >= 4 - 1e-9
            torch.cuda.synchronize()
            assert all(e.query() for e in events)
    finally:
        L.check(L.pidm_unet_set_grad_events(eng.handle, 1, None))


def test_conditioning_branch_replay(backend):
    """Guidance branch: steps with and without a conditioning field alternate - two forward graphs, two backward graphs, and the
    conditioning gradients are zero-filled by the (never captured) memsets when the branch was not used."""
    L, dev = backend
    import os
    ma, mb = _model(dev, L), _model(dev, L)
    data = _inputs(dev, 8)       # a key is replayed from its fourth use on (first: tables not yet on the device, then sighting, capture)
    g = torch.Generator().manual_seed(21)
    conds = [torch.randn(2, 256, 2, generator=g).to(dev) for _ in range(8)]
    for i, d in enumerate(data):
        c = conds[i] if i % 2 == 0 else None
        os.environ["PIDM_GRAPH"] = "0"
        reload_knobs()
        try:
            o_ref, g_ref = _train_step(ma, *d, cond=c)
        finally:
            del os.environ["PIDM_GRAPH"]
            reload_knobs()
        o, gg = _train_step(mb, *d, cond=c)
        assert torch.equal(o, o_ref) and torch.equal(gg, g_ref), i


def test_frozen_weights_scope_packs_once_and_only_inside_the_scope(backend, monkeypatch):
    """frozen_weights(model): the first inference forward packs / splits the weights, the following ones of the scope do not (one
    kernel less each, same numbers); outside a scope every forward packs again - a raw write to a parameter between two forwards
    (what the fused optimizer and the EMA kernels do) is seen at once."""
    from physicsinformeddiffusionmodels_amd._engine import frozen_weights
    L, dev = backend
    monkeypatch.setenv("PIDM_GRAPH", "0")                   # count kernels launch by launch
    m = _model(dev, L).eval()
    (x, t, _), (x2, t2, _) = _inputs(dev, 2)

    def fwd(xx, tt):
        c0 = _counts(L)["eager"]
        with torch.no_grad():
            o = m(xx, tt).clone()
        return o, _counts(L)["eager"] - c0

    ref, n_pack = fwd(x, t)
    ref2, n2 = fwd(x2, t2)
    assert n2 == n_pack                                      # no scope: every forward re-packs
    with frozen_weights(m):
        a, na = fwd(x, t)
        b, nb = fwd(x2, t2)
        with frozen_weights(m):                              # nested: same scope
            c, nc = fwd(x, t)
    assert na == n_pack and nb == n_pack - 1 and nc == n_pack - 1
    assert torch.equal(a, ref) and torch.equal(b, ref2) and torch.equal(c, ref)
    assert getattr(m, "_pidm_frozen", None) is None
    # a write through the raw storage, invisible to autograd's version counters
    with torch.no_grad():
        dict(m.named_parameters())["final_conv.1.weight"].data.mul_(1.5)
    d, nd = fwd(x, t)
    assert nd == n_pack and not torch.equal(d, ref)
    with frozen_weights(m):                                  # a NEW scope starts with a pack
        e, ne = fwd(x, t)
    assert ne == n_pack and torch.equal(e, d)


def test_wide_models_keep_the_backward_launch_by_launch(backend, monkeypatch):
    """>= 512 channels at the deepest level (the mechanics configuration): forward replayed, backward enqueued kernel by kernel (its
    weight gradients may take the side stream); PIDM_GRAPH_BWD=1 / =0 force either form for any model.  Same numbers in all forms."""
    L, dev = backend
    monkeypatch.setenv("PIDM_GRAPH_BWD_WIDE", "64")          # the rule's threshold (default 512 channels), lowered to this test's size
    data = _inputs(dev, 3, B=1, P=8)

    def run(**kw):
        m = _model(dev, L, **kw)
        c0 = _counts(L)
        outs = []
        for x, t, w in data:
            for p in m.parameters():
                p.grad = None
            out = m(x, t)
            (out * w).sum().backward()
            outs.append((out.detach().clone(), get_engine(m, 8, m._pidm_lib).flat_grad.clone()))
        c1 = _counts(L)
        return outs, c1["captures"] - c0["captures"]

    wide, caps_wide = run(dim=8, dim_mults=(1, 8))
    assert caps_wide == 1                                    # the forward graph only
    monkeypatch.setenv("PIDM_GRAPH_BWD", "1")
    forced, caps_forced = run(dim=8, dim_mults=(1, 8))
    assert caps_forced == 2
    monkeypatch.setenv("PIDM_GRAPH_BWD", "0")
    narrow, caps_narrow = run()
    assert caps_narrow == 1
    monkeypatch.delenv("PIDM_GRAPH_BWD")
    narrow_default, caps_default = run()
    assert caps_default == 2
    for (o1, g1), (o2, g2) in zip(wide, forced):
        assert torch.equal(o1, o2) and torch.equal(g1, g2)
    for (o1, g1), (o2, g2) in zip(narrow, narrow_default):
        assert torch.equal(o1, o2) and torch.equal(g1, g2)
