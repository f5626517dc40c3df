"""Whole-UNet forward + backward through the native engine (csrc/unet_engine.hip) vs the golden vectors that
were produced by running the genuine reference (tests/golden/g5*, g6*).
`backend` = host-emulated build of the same sources on CPU tensors (default) or the real gfx950 library (-m gpu)."""
import os

import numpy as np
import pytest
import torch

from oracle import pidm_oracle as O
from physicsinformeddiffusionmodels_amd._engine import unet_apply
from physicsinformeddiffusionmodels_amd.unet_model import Unet3D

G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def build_model(dim, dev):
    m = Unet3D(dim=dim, channels=2)
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    return m.to(dev)


def run_case(backend, tag, dim, full):
    L, dev = backend
    lib = L if dev.type == "cpu" else None
    g = np.load(os.path.join(G, tag + ".npz"))
    m = build_model(dim, dev)
    x = torch.from_numpy(g["x"]).to(dev)
    t = torch.from_numpy(g["t"]).to(dev)
    B, C, P, _ = x.shape
    x_bxyc = x.permute(0, 2, 3, 1).reshape(B, P * P, C)
    out = unet_apply(m, x_bxyc, t, lib=lib)
    o = out.detach().cpu().numpy()
    if full:
        assert rel(o, g["out"]) < 3e-5
    else:
        assert rel(o[:, :, ::8, ::8], g["out_probe"]) < 3e-5
        assert abs(float(o.astype(np.float64).sum()) - float(g["out_sum"])) < 1e-4 * float(g["out_abs_sum"])
    (out * torch.from_numpy(g["w"]).to(dev)).sum().backward()
    names = [str(s) for s in g["grad_names"]]
    params = dict(m.named_parameters())
    have = sorted(k for k, p in params.items() if p.grad is not None)
    assert have == sorted(names)  # exactly the reference's used-parameter set; the rest keep grad None
    gmax = float(np.max(g["grad_norms"]))
    bad = []
    for k, ref in zip(names, g["grad_norms"]):
        got = params[k].grad.double().norm().item()
        if not abs(got - ref) <= 5e-4 * ref + 2e-6 * gmax:
            bad.append((k, got, float(ref)))
    assert not bad, bad[:8]
    for f in g.files:
        if f.startswith("grad/"):
            k = f[5:]
            assert rel(params[k].grad.cpu().numpy(), g[f]) < 1e-3, k


def test_unet_dim8_p16(backend):
    run_case(backend, "g5_unet_dim8_p16", 8, True)


def test_unet_dim16_p32(backend):
    run_case(backend, "g5b_unet_dim16_p32", 16, True)


def test_tiled_weight_repack_on_the_emulator(monkeypatch):
    """The emulated build packs element by element by default (the tile form's barrier is expensive with fibers; on the GPU the
    tiled form is the default and every -m gpu test runs it): here the tiled form - 32 x 32 x taps tiles transposed through LDS,
    csrc/k_conv.hip: pack_tile - against the same golden vectors, forward and input-gradient packings, pieces included."""
    import torch
    from tests.emu_util import emu_lib
    monkeypatch.setenv("PIDM_PACK_TILED", "1")
    run_case((emu_lib(), torch.device("cpu")), "g5b_unet_dim16_p32", 16, True)


# attention forms of the 64x64 / 32x32 levels: "proj" = no qkv tensor (k_attn_proj.hip, the default), "fused" = qkv tensor with the
# attention fused into the to_out projection, "separate" = qkv tensor, separate projection kernels
ATTN_FORMS = {"proj": {}, "fused": {"PIDM_NO_LAP": "1", "PIDM_LA_FUSED_MIN_WGS": "1"},
              "separate": {"PIDM_NO_LAP": "1", "PIDM_LA_FUSED_MIN_WGS": "1000000"}}


@pytest.mark.parametrize("form", ["proj"])     # (all three forms run on the real GPU below; the fused form at small sizes in test_kernels_attn.py)
def test_unet_dim32_p64_emulated(monkeypatch, form):
    """The full Darcy model (dim=32, 64x64: golden g6 from the genuine reference) through the host emulator, ~15 s
    (the same golden runs on the real GPU in test_unet_dim32_p64_gpu)."""
    from tests.emu_util import emu_lib
    for k, v in ATTN_FORMS[form].items():
        monkeypatch.setenv(k, v)
    # (two images are too little work for the row-streaming 3x3 kernel's default: let it take the 64- and 32-wide levels - strips
    # of 16 and 4 rows - with the epilogues the engine asks for: GroupNorm partials, GroupNorm-backward sums, residuals)
    monkeypatch.setenv("PIDM_CONV_RS_WAVES", "16")
    run_case((emu_lib(), torch.device("cpu")), "g6_unet_dim32_p64", 32, False)


@pytest.mark.gpu
@pytest.mark.parametrize("form", sorted(ATTN_FORMS))
def test_unet_dim32_p64_gpu(monkeypatch, form):
    from physicsinformeddiffusionmodels_amd._lib import get_lib
    for k, v in ATTN_FORMS[form].items():
        monkeypatch.setenv(k, v)
    run_case((get_lib(), torch.device("cuda:0")), "g6_unet_dim32_p64", 32, False)


def run_dim128_case(L, dev):
    """Golden g19 (genuine reference): Unet3D(dim=128, channels=10, out_dim=3, sigmoid_last_channel=True), the model main.py:126
    builds for topology optimisation - channel widths 128..1024 on 64x64..8x8 maps, i.e. the conv / wgrad tile variants that
    no dim<=32 golden reaches.  Output probes and sum, the exact 259-tensor used set, every gradient norm, 20 strided gradient
    samples."""
    lib = L if dev.type == "cpu" else None
    g = np.load(os.path.join(G, "g19_unet_dim128_mech.npz"))
    m = Unet3D(dim=128, channels=10, out_dim=3, sigmoid_last_channel=True)
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    m = m.to(dev)
    x = torch.from_numpy(g["x"]).to(dev)
    t = torch.from_numpy(g["t"]).to(dev)
    out = unet_apply(m, x.permute(0, 2, 3, 1).reshape(1, 64 * 64, 10), t, lib=lib)
    o = out.detach().cpu().numpy()
    assert rel(o[:, :, ::4, ::4], g["out_probe"]) < 3e-5
    assert abs(float(o.astype(np.float64).sum()) - float(g["out_sum"])) < 1e-4 * float(g["out_abs_sum"])
    (out * torch.from_numpy(g["w"]).to(dev)).sum().backward()
    names = [str(s) for s in g["grad_names"]]
    params = dict(m.named_parameters())
    assert sorted(k for k, p in params.items() if p.grad is not None) == sorted(names)
    gmax = float(np.max(g["grad_norms"]))
    bad = []
    for k, ref in zip(names, g["grad_norms"]):
        got = params[k].grad.double().norm().item()
        if not abs(got - ref) <= 5e-4 * ref + 2e-6 * gmax:
            bad.append((k, got, float(ref)))
    assert not bad, bad[:8]
    n = 0
    for f in g.files:
        if f.startswith("grad/"):
            k = f[5:]
            assert rel(params[k].grad.reshape(-1)[::int(g["gstride/" + k])].cpu().numpy(), g[f]) < 1e-3, k
            n += 1
    assert n == 20


@pytest.mark.gpu
def test_unet_dim128_mechanics_shape_gpu():
    from physicsinformeddiffusionmodels_amd._lib import get_lib
    run_dim128_case(get_lib(), torch.device("cuda:0"))


def test_self_conditioning_vs_reference(backend):
    """Unet3D(self_condition=True) (golden g15, genuine reference): init_conv reads cat(x_self_cond, x); a missing
    x_self_cond means zeros; gradients of all used parameters."""
    import numpy as np
    L, dev = backend
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g15_selfcond_dim8_p16.npz"))
    m = Unet3D(dim=8, channels=2, self_condition=True)
    assert tuple(m.init_conv.weight.shape) == tuple(g["init_conv_shape"])
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    m = m.to(dev)
    m._pidm_lib = L if dev.type == "cpu" else None
    x, sc, t = (torch.from_numpy(g[k]).to(dev) for k in ("x", "sc", "t"))
    out_sc = m(x, t, x_self_cond=sc)
    with torch.no_grad():
        out_none = m(x, t)
    for mine, ref in ((out_sc, g["out_sc"]), (out_none, g["out_none"])):
        assert (mine.detach().cpu() - torch.from_numpy(ref)).abs().max().item() < 3e-5 * np.abs(ref).max()
    (out_sc * torch.from_numpy(g["w"]).to(dev)).sum().backward()
    params = dict(m.named_parameters())
    names = [str(s) for s in g["grad_names"]]
    assert sorted(k for k, v in params.items() if v.grad is not None) == sorted(names)
    gmax = float(g["grad_norms"].max())
    for k, n in zip(names, g["grad_norms"]):
        assert abs(params[k].grad.double().norm().item() - n) <= 5e-4 * n + 1e-6 * gmax, k
    # the oracle restatement agrees as well
    p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = O.unet_forward(p, x.cpu(), t.cpu(), O.UnetCfg(dim=8, channels=2, self_condition=True), x_self_cond=sc.cpu())
    assert (ref - torch.from_numpy(g["out_sc"])).abs().max().item() < 3e-5 * np.abs(g["out_sc"]).max()


def test_inference_forward_between_training_forward_and_backward(backend):
    """An evaluation-mode forward issued while a training forward still waits for its backward (EMA evaluation inside a
    step, a debugging print, ...) must neither unbind the gradient buffers nor touch the activation tape."""
    L, dev = backend
    m = Unet3D(dim=8, channels=2, dim_mults=(1, 2))
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    m = m.to(dev)
    m._pidm_lib = L if dev.type == "cpu" else None
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 256, 2, generator=g).to(dev)
    x2 = torch.randn(3, 256, 2, generator=g).to(dev)
    t = torch.tensor([4, 70], device=dev)

    def grads(interleave):
        for p in m.parameters():
            p.grad = None
        out = m(x, t)
        if interleave:
            with torch.no_grad():
                m(x2, torch.tensor([1, 2, 3], device=dev))      # different batch size: new workspace on the sibling engine
        out.square().sum().backward()
        return {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}

    a, b = grads(False), grads(True)
    assert a.keys() == b.keys() and len(a) == 147
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_gradient_accumulation_across_backward_calls(backend):
    """Two backward passes without zero_grad accumulate (torch semantics), although the engine writes its flat buffer."""
    L, dev = backend
    m = Unet3D(dim=8, channels=2, dim_mults=(1, 2))
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    m = m.to(dev)
    m._pidm_lib = L if dev.type == "cpu" else None
    g = torch.Generator().manual_seed(4)
    xa, xb = (torch.randn(2, 256, 2, generator=g).to(dev) for _ in range(2))
    t = torch.tensor([10, 60], device=dev)

    def one(x):
        for p in m.parameters():
            p.grad = None
        m(x, t).square().sum().backward()
        return {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}

    ga, gb = one(xa), one(xb)
    for p in m.parameters():
        p.grad = None
    m(xa, t).square().sum().backward()
    m(xb, t).square().sum().backward()          # no zero_grad in between
    for k, v in m.named_parameters():
        if v.grad is None:
            continue
        ref = ga[k] + gb[k]
        assert (v.grad - ref).abs().max().item() <= 1e-6 * max(ref.abs().max().item(), 1e-12), k


def test_steady_state_backward_uploads_no_reduction_table(backend):
    """The deferred-reduction descriptor table is uploaded once; identical later backward passes find the device copy unchanged
    (the descriptors are value-initialised, so the struct's padding bytes do not take part in the comparison) - with the
    three-phase split of the data-parallel overlap as well."""
    L, dev = backend
    m = Unet3D(dim=8, channels=2, dim_mults=(1, 2))
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    m = m.to(dev)
    m._pidm_lib = L if dev.type == "cpu" else None
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 256, 2, generator=g).to(dev)
    t = torch.tensor([3, 77], device=dev)

    def step():
        for p in m.parameters():
            p.grad = None
        m(x, t).square().sum().backward()

    from physicsinformeddiffusionmodels_amd._engine import get_engine
    eng = get_engine(m, 16, m._pidm_lib)
    for phases in (1, 3):
        L.check(L.pidm_unet_set_grad_events(eng.handle, phases, None))
        step()                                      # may upload (first pass / phase split changed)
        n0 = L.pidm_debug_reduce_table_uploads()
        step()
        step()
        with torch.no_grad():
            m(x, t)                                 # an inference forward in between does not disturb the table either
        step()
        assert L.pidm_debug_reduce_table_uploads() == n0, phases
    L.check(L.pidm_unet_set_grad_events(eng.handle, 1, None))


def test_attention_form_is_latched_per_forward(backend, monkeypatch):
    """PIDM_NO_LAP / PIDM_LAP_MIN_N flipped on a live handle between a forward and its backward: the backward replays the form
    the forward took (same gradients as an undisturbed step), and the next forward re-plans its workspace for the new form."""
    L, dev = backend
    m = Unet3D(dim=8, channels=2)
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    m = m.to(dev)
    m._pidm_lib = L if dev.type == "cpu" else None
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 1024, 2, generator=g).to(dev)
    t = torch.tensor([9, 41], device=dev)

    def grads(flip_to=None):
        for p in m.parameters():
            p.grad = None
        out = m(x, t)
        if flip_to is not None:
            for k, v in flip_to.items():
                monkeypatch.setenv(k, v)
        out.square().sum().backward()
        return {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}

    monkeypatch.setenv("PIDM_LAP_MIN_N", "64")              # projected form on the 32x32 ... 8x8 levels of this small model
    monkeypatch.delenv("PIDM_NO_LAP", raising=False)
    ref_proj = grads()
    got = grads(flip_to={"PIDM_NO_LAP": "1"})                # forward projected, knobs flipped before backward
    for k in ref_proj:
        assert torch.equal(got[k], ref_proj[k]), k
    ref_qkv = grads()                                        # now planned and run in the qkv form
    monkeypatch.delenv("PIDM_NO_LAP")
    got = grads()                                            # and back
    gmax = max(v.abs().max().item() for v in ref_proj.values())
    for k in ref_proj:
        assert torch.equal(got[k], ref_proj[k]), k
        # the two forms agree to rounding (mathematically-zero gradients, e.g. conv biases under a GroupNorm, are noise in both)
        assert (ref_qkv[k] - ref_proj[k]).abs().max().item() <= 2e-3 * ref_proj[k].abs().max().item() + 1e-5 * gmax, k


@pytest.mark.parametrize("form", ["bxyc", "bcpp", "bc1pp"])
def test_input_forms_vs_reference(backend, form):
    """Unet3D.forward accepts [B, P*P, C], [B, C, P, P] and [B, C, 1, P, P] (src/unet_model.py:554-562) and returns [B, out, P, P]
    - with the frame axis kept for the 5-D form (:616-618).  Output and input gradient of each form against the genuine reference
    (golden g23)."""
    L, dev = backend
    g = np.load(os.path.join(G, "g23_unet_input_forms.npz"))
    m = build_model(8, dev)
    m._pidm_lib = L if dev.type == "cpu" else None
    x = torch.from_numpy(g["x"]).to(dev)
    t = torch.from_numpy(g["t"]).to(dev)
    w = torch.from_numpy(g["w"]).to(dev)
    B, C, P, _ = x.shape
    xin = {"bxyc": x.permute(0, 2, 3, 1).reshape(B, P * P, C), "bcpp": x, "bc1pp": x.unsqueeze(2)}[form].clone().requires_grad_(True)
    out = m(xin, t)
    ref = g["out_" + form]
    assert tuple(out.shape) == ref.shape == ((B, 2, 1, P, P) if form == "bc1pp" else (B, 2, P, P))
    assert rel(out.detach().cpu().numpy(), ref) < 3e-5
    (out * (w.unsqueeze(2) if form == "bc1pp" else w)).sum().backward()
    assert tuple(xin.grad.shape) == g["gx_" + form].shape
    assert rel(xin.grad.cpu().numpy(), g["gx_" + form]) < 2e-4


def test_input_form_errors(backend):
    L, dev = backend
    m = build_model(8, dev)
    m._pidm_lib = L if dev.type == "cpu" else None
    t = torch.tensor([1, 2], device=dev)
    with pytest.raises(ValueError):
        m(torch.zeros(2, 16, device=dev), t)                     # neither image nor sequence (src/unet_model.py:561-562)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 2, 3, 16, 16, device=dev), t)           # more than one frame: outside the engine's (image) path
    with pytest.raises(ValueError):
        m(torch.zeros(2, 3, 16, 16, device=dev), t)              # wrong channel count


def test_grouped_weight_gradients(backend, monkeypatch):
    """Round 5: the weight-gradient problems of a backward pass that one of the three streaming families takes (3x3 / stride-1,
    4x4 / stride-2, 1x1 pixel streams) are queued and run by ONE launch per family and flush (conv_wgrad_rs_multi_kernel,
    conv_wgrad_rs4_multi_kernel, conv_wgrad_1x1_multi_kernel; PIDM_WGRAD_GROUP = problems per flush, default 1 << 20: one flush per
    reduction phase, 0: off).  Same kernel bodies, same partial slabs, same fixed-order reduction: the gradients are BIT-identical
    to the problem-by-problem launches (PIDM_WGRAD_GROUP=0) for any group size, also with the three-phase reduction of the
    data-parallel exchange; fewer splits per problem (PIDM_WGRAD_GROUP_SPLITDIV) change only the summation order; and a
    steady-state pass uploads neither table."""
    L, dev = backend
    from physicsinformeddiffusionmodels_amd._engine import get_engine
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 256, 2, generator=g).to(dev)
    t = torch.tensor([3, 77, 50], device=dev)
    w = torch.randn(3, 2, 16, 16, generator=g).to(dev)

    def grads(env, phases=1, steps=1):
        for k in ("PIDM_WGRAD_GROUP", "PIDM_WGRAD_GROUP_SPLITDIV"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = Unet3D(dim=32, channels=2, dim_mults=(1, 2))          # 32 / 64 channels at 16x16 and 8x8: every 3x3 layer takes the kernel
        m.load_state_dict(O.fill_state_dict(m.state_dict()))
        m = m.to(dev)
        m._pidm_lib = L if dev.type == "cpu" else None
        eng = get_engine(m, 16, m._pidm_lib)
        L.check(L.pidm_unet_set_grad_events(eng.handle, phases, None))
        n0 = None
        for s in range(steps):
            for p in m.parameters():
                p.grad = None
            (m(x, t) * w).sum().backward()
            if s == 0:
                n0 = L.pidm_debug_reduce_table_uploads()
        if steps > 1:
            assert L.pidm_debug_reduce_table_uploads() == n0          # replayed / repeated passes find both tables unchanged
        out = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        L.check(L.pidm_unet_set_grad_events(eng.handle, 1, None))
        return out
    ref = grads({"PIDM_WGRAD_GROUP": "0"})
    assert len(ref) > 50
    same_splits = {"PIDM_WGRAD_GROUP_SPLITDIV": "1"}
    for env, phases, steps in ((same_splits, 1, 4), (dict(same_splits, PIDM_WGRAD_GROUP="3"), 1, 1), (same_splits, 3, 4),
                               (dict(same_splits, PIDM_WGRAD_GROUP="1"), 3, 1)):
        got = grads(env, phases, steps)
        assert got.keys() == ref.keys()
        for k in ref:
            assert torch.equal(got[k], ref[k]), (env, phases, k)
    # the default (a quarter of the splits per problem) and another divisor: the same sums in another order
    for env, steps in (({}, 4), ({"PIDM_WGRAD_GROUP_SPLITDIV": "2"}, 1)):
        got = grads(env, 1, steps)
        for k in ref:
            assert (got[k] - ref[k]).abs().max().item() <= 2e-5 * max(ref[k].abs().max().item(), 1e-6), k
    a, b = grads({}, 1, 2), grads({}, 3, 2)                  # and the default is bit-identical run to run, one phase or three
    for k in ref:
        assert torch.equal(a[k], b[k]), k


def test_grouped_weight_gradients_toggled_on_a_live_handle(backend, monkeypatch):
    """Grouping switched on -> off -> on through pidm_reload_knobs on ONE engine and ONE workspace: the pass without grouping
    recycles the arena's frames, so its partial slabs may land on the grouped tables the first pass uploaded; the third pass must
    upload them again (not trust its host copies) and produce the same gradients as the first."""
    L, dev = backend
    g = torch.Generator().manual_seed(12)
    x = torch.randn(3, 256, 2, generator=g).to(dev)
    t = torch.tensor([5, 60, 99], device=dev)
    w = torch.randn(3, 2, 16, 16, generator=g).to(dev)
    m = Unet3D(dim=32, channels=2, dim_mults=(1, 2))
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    m = m.to(dev)
    m._pidm_lib = L if dev.type == "cpu" else None

    def grads(group):
        monkeypatch.setenv("PIDM_WGRAD_GROUP", group)
        out = None
        for _ in range(3):                                   # (the third call with one key replays a graph on the GPU)
            for p in m.parameters():
                p.grad = None
            (m(x, t) * w).sum().backward()
            out = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        return out
    on1, up1 = grads("1048576"), L.pidm_debug_reduce_table_uploads()
    off = grads("0")
    on2, up2 = grads("1048576"), L.pidm_debug_reduce_table_uploads()
    assert up2 > up1                                         # the grouped tables were uploaded again
    assert len(on1) > 50 and on1.keys() == off.keys() == on2.keys()
    for k in on1:
        assert torch.equal(on1[k], on2[k]), k
        assert (off[k] - on1[k]).abs().max().item() <= 2e-5 * max(on1[k].abs().max().item(), 1e-6), k


def test_layernorm_fused_into_groupnorm_apply_is_bit_identical(backend, monkeypatch):
    """Round 6: the PreNorm LayerNorm of every attention block (src/unet_model.py:139-145,207-210) rides in the last gn_apply of the
    ResnetBlock in front of it (same operation order as the separate layernorm kernel: one launch and one read of the activation
    less per attention block).  Output, input gradient and every parameter gradient must equal the unfused path
    (PIDM_NO_GN_LN_FUSE=1) bit for bit, in training and in inference mode."""
    L, dev = backend
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 256, 2, generator=g).to(dev)
    t = torch.tensor([7, 91], device=dev)
    w = torch.randn(2, 2, 16, 16, generator=g).to(dev)

    def run(fused):
        if fused:
            monkeypatch.delenv("PIDM_NO_GN_LN_FUSE", raising=False)
        else:
            monkeypatch.setenv("PIDM_NO_GN_LN_FUSE", "1")
        m = Unet3D(dim=16, channels=2, dim_mults=(1, 2, 4))      # 16 / 32 / 64 channels: 4, 8 and 16 lanes per pixel
        m.load_state_dict(O.fill_state_dict(m.state_dict()))
        m = m.to(dev)
        m._pidm_lib = L if dev.type == "cpu" else None
        xin = x.clone().requires_grad_(True)
        out = m(xin, t)
        (out * w).sum().backward()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        m.eval()
        with torch.no_grad():
            out_eval = m(x, t)
            # (inference passes run GroupNorm + FiLM + SiLU in place; PIDM_NO_GN_INPLACE=1 keeps the separate buffers)
            monkeypatch.setenv("PIDM_NO_GN_INPLACE", "1")
            out_eval2 = m(x, t)
            monkeypatch.delenv("PIDM_NO_GN_INPLACE")
        assert torch.equal(out_eval, out_eval2) and torch.equal(out_eval, out.detach())
        return out.detach().clone(), xin.grad.detach().clone(), grads, out_eval.detach().clone()
    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    assert a[2].keys() == b[2].keys() and len(a[2]) > 100
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k


def test_groupnorm_backward_sums_from_the_layernorm_backward(backend, monkeypatch):
    """Round 6: the LayerNorm backward that ends an attention block's backward also takes the first pass of the GroupNorm backward
    of the ResnetBlock in front (per-(image, chunk, channel) sums S1 = sum dv, S2 = sum dv xhat) - gn_bwd_reduce_kernel is not
    launched there - when PIDM_LN_GN_SUMS=1 asks for it (measured slower per step, so it is off by default).  Same values summed in
    another grouping: gradients agree with the separate pass to fp32 rounding, and are bit-identical run to run."""
    L, dev = backend
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 256, 2, generator=g).to(dev)
    t = torch.tensor([2, 40, 88], device=dev)
    w = torch.randn(3, 2, 16, 16, generator=g).to(dev)

    def run(fused):
        if fused:
            monkeypatch.setenv("PIDM_LN_GN_SUMS", "1")
        else:
            monkeypatch.delenv("PIDM_LN_GN_SUMS", raising=False)
        m = Unet3D(dim=16, channels=2, dim_mults=(1, 2, 4))
        m.load_state_dict(O.fill_state_dict(m.state_dict()))
        m = m.to(dev)
        m._pidm_lib = L if dev.type == "cpu" else None
        xin = x.clone().requires_grad_(True)
        (m(xin, t) * w).sum().backward()
        return xin.grad.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    a, a2, b = run(True), run(True), run(False)
    assert torch.equal(a[0], a2[0]) and all(torch.equal(a[1][k], a2[1][k]) for k in a[1])
    assert a[1].keys() == b[1].keys() and len(a[1]) > 100
    assert (a[0] - b[0]).abs().max().item() <= 2e-5 * b[0].abs().max().item()
    for k in a[1]:
        assert (a[1][k] - b[1][k]).abs().max().item() <= 2e-5 * max(b[1][k].abs().max().item(), 1e-6), k
