"""Linear-attention kernels (csrc/k_attn.hip) through pidm_linear_attention_forward/backward vs the oracle's restatement
of SpatialLinearAttention (reference src/unet_model.py:287-297) and its autograd.  `backend` = host-emulated build of the
same sources (default run) or the gfx950 library (-m gpu)."""
import pytest
import torch

from oracle import pidm_oracle as O
from physicsinformeddiffusionmodels_amd._lib import ptr, stream_ptr


def rel(a, b):
    a, b = a.detach().cpu(), b.detach().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


CASES = [
    # B, H, heads
    (2, 16, 4),     # N = 256: matrix-core backward (N % 128 == 0), one k-statistics segment
    (1, 32, 2),     # N = 1024: two pixel splits, four k segments
    (3, 8, 4),      # N = 64: generic per-pixel backward, aligned forward
    (5, 4, 1),      # N = 16 < 32: a wave's pixels straddle images (non-aligned forward)
    (2, 64, 1),     # N = 4096: the full-resolution level of the Darcy model (8 pixel splits, 16 k segments)
]


@pytest.mark.parametrize("B,H,heads", CASES)
def test_linear_attention_fwd_bwd(backend, B, H, heads):
    L, dev = backend
    st = stream_ptr(dev)
    N, HD = H * H, heads * 32
    g = torch.Generator().manual_seed(11 + H)
    qkv = torch.randn(B, 3 * HD, H, H, generator=g) * 1.5
    qkv[:, HD:2 * HD] += 2.0 * torch.randn(B, HD, 1, 1, generator=g)          # per-column offsets: the k-softmax max matters
    d_out = torch.randn(B, HD, H, H, generator=g)
    qr = qkv.clone().requires_grad_(True)
    ref = O.linear_attention_core(qr, heads, 32)
    (gref,) = torch.autograd.grad(ref, qr, d_out)

    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    qd, dod = nhwc(qkv), nhwc(d_out)
    out = torch.empty(B, N, HD, device=dev)
    kstat = torch.empty(B * HD * 2, device=dev)
    ctx = torch.empty(B * heads * 1024, device=dev)
    qstat = torch.empty(B * N * heads * 2, device=dev)
    ws = torch.empty(L.pidm_linear_attention_ws(B, N, heads), dtype=torch.uint8, device=dev)
    L.check(L.pidm_linear_attention_forward(ptr(qd), ptr(out), ptr(kstat), ptr(ctx), ptr(qstat), B, N, heads, ptr(ws), st))
    got = out.reshape(B, H, H, HD).permute(0, 3, 1, 2)
    assert rel(got, ref) < 5e-6
    dqkv = torch.empty(B, N, 3 * HD, device=dev)
    L.check(L.pidm_linear_attention_backward(ptr(qd), ptr(kstat), ptr(qstat), ptr(ctx), ptr(dod), ptr(dqkv), B, N, heads,
                                             ptr(ws), st))
    gg = dqkv.reshape(B, H, H, 3 * HD).permute(0, 3, 1, 2)
    for c in range(3):                                    # dq, dk, dv have very different magnitudes: compare each to its own scale
        sl = slice(c * HD, (c + 1) * HD)
        assert rel(gg[:, sl], gref[:, sl]) < 2e-5, "qkv"[c]


FUSED_CASES = [
    # B, H, heads, Cout
    (2, 16, 4, 32),     # N = 256, one pixel split
    (1, 32, 2, 64),     # N = 1024, two pixel splits, two output-channel tiles
    (2, 16, 3, 128),    # four output-channel tiles (the 16x16 level of the Darcy model)
    (1, 64, 2, 32),     # N = 4096: eight pixel splits
]


@pytest.mark.parametrize("B,H,heads,Cout", FUSED_CASES)
def test_linear_attention_fused_with_projection(backend, B, H, heads, Cout):
    """forward: to_out(attention) + bias + residual without the attention output; backward: dqkv and the to_out weight gradient
    from the gradient of the projection's OUTPUT (d_out = d_y W never stored), on the statistics the fused forward saved."""
    import torch.nn.functional as F
    L, dev = backend
    st = stream_ptr(dev)
    N, HD = H * H, heads * 32
    g = torch.Generator().manual_seed(23 + H + Cout)
    qkv = torch.randn(B, 3 * HD, H, H, generator=g) * 1.5
    qkv[:, HD:2 * HD] += 2.0 * torch.randn(B, HD, 1, 1, generator=g)
    w_out = torch.randn(Cout, HD, 1, 1, generator=g) * 0.2
    d_y = torch.randn(B, Cout, H, H, generator=g)
    qr, wr = qkv.clone().requires_grad_(True), w_out.clone().requires_grad_(True)
    y = F.conv2d(O.linear_attention_core(qr, heads, 32), wr)
    gq_ref, gw_ref = torch.autograd.grad(y, (qr, wr), d_y)

    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    qd, dyd, wd = nhwc(qkv), nhwc(d_y), w_out.reshape(Cout, HD).contiguous().to(dev)
    kstat = torch.empty(B * HD * 2, device=dev)
    ctx = torch.empty(B * heads * 1024, device=dev)
    qstat = torch.empty(B * N * heads * 2, device=dev)
    ws = torch.empty(max(L.pidm_linear_attention_ws(B, N, heads), L.pidm_linear_attention_out_backward_ws(B, N, heads, Cout)),
                     dtype=torch.uint8, device=dev)
    bias = torch.randn(Cout, generator=g)
    resid = torch.randn(B, Cout, H, H, generator=g)
    yd = torch.empty(B, N, Cout, device=dev)
    bd, rd = bias.to(dev), nhwc(resid)
    L.check(L.pidm_linear_attention_out_forward(ptr(qd), ptr(wd), ptr(bd), ptr(rd), ptr(yd), Cout, ptr(kstat), ptr(ctx), ptr(qstat),
                                                B, N, heads, ptr(ws), st))
    y_ref = y.detach() + bias.reshape(1, -1, 1, 1) + resid
    assert rel(yd.reshape(B, H, H, Cout).permute(0, 3, 1, 2), y_ref) < 5e-6
    dqkv = torch.empty(B, N, 3 * HD, device=dev)
    dw = torch.empty(Cout, HD, device=dev)
    L.check(L.pidm_linear_attention_out_backward(ptr(qd), ptr(kstat), ptr(qstat), ptr(ctx), ptr(dyd), Cout, ptr(wd), Cout,
                                                 ptr(dqkv), ptr(dw), B, N, heads, ptr(ws), st))
    gg = dqkv.reshape(B, H, H, 3 * HD).permute(0, 3, 1, 2)
    for c in range(3):
        sl = slice(c * HD, (c + 1) * HD)
        assert rel(gg[:, sl], gq_ref[:, sl]) < 2e-5, "qkv"[c]
    assert rel(dw, gw_ref.reshape(Cout, HD)) < 1e-5


def test_linear_attention_vs_reference_module_golden(backend):
    """g18 (the genuine SpatialLinearAttention module run by oracle/make_golden.py): both forms of the kernels - attention with a
    separate projection, and attention fused with to_out - against the reference's output and its gradients of qkv and
    to_out.weight.  The two 1x1 convolutions around the attention are plain matrix products here."""
    import os
    import numpy as np
    L, dev = backend
    st = stream_ptr(dev)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g18_linear_attention.npz"))
    heads = int(z["heads"])
    x, gy = torch.tensor(z["x"]), torch.tensor(z["gy"])
    w_qkv, w_out, b_out = torch.tensor(z["w_qkv"]), torch.tensor(z["w_out"]), torch.tensor(z["b_out"])
    B, C, H, _ = x.shape
    N, HD = H * H, heads * 32
    xs = x.permute(0, 2, 3, 1).reshape(B, N, C)                              # channels-last pixels
    qd = (xs @ w_qkv.t()).contiguous().to(dev)                               # to_qkv
    gyd = gy.permute(0, 2, 3, 1).reshape(B, N, C).contiguous().to(dev)
    wd, bd = w_out.contiguous().to(dev), b_out.to(dev)
    kstat = torch.empty(B * HD * 2, device=dev)
    ctx = torch.empty(B * heads * 1024, device=dev)
    qstat = torch.empty(B * N * heads * 2, device=dev)
    ws = torch.empty(max(L.pidm_linear_attention_ws(B, N, heads), L.pidm_linear_attention_out_backward_ws(B, N, heads, C)),
                     dtype=torch.uint8, device=dev)
    y_ref = torch.tensor(z["y"]).permute(0, 2, 3, 1).reshape(B, N, C)
    dq_ref = torch.tensor(z["d_qkv"]).permute(0, 2, 3, 1).reshape(B, N, 3 * HD)

    # (a) separate: attention kernels, projection as a matrix product
    out = torch.empty(B, N, HD, device=dev)
    L.check(L.pidm_linear_attention_forward(ptr(qd), ptr(out), ptr(kstat), ptr(ctx), ptr(qstat), B, N, heads, ptr(ws), st))
    y_a = out.cpu() @ w_out.t() + b_out
    assert rel(y_a, y_ref) < 5e-6
    d_out = (gyd.cpu() @ w_out).contiguous().to(dev)
    dqkv = torch.empty(B, N, 3 * HD, device=dev)
    L.check(L.pidm_linear_attention_backward(ptr(qd), ptr(kstat), ptr(qstat), ptr(ctx), ptr(d_out), ptr(dqkv), B, N, heads, ptr(ws), st))
    for c in range(3):
        sl = slice(c * HD, (c + 1) * HD)
        assert rel(dqkv[..., sl], dq_ref[..., sl]) < 2e-5, "qkv"[c]

    # (b) fused with the projection
    yd = torch.empty(B, N, C, device=dev)
    L.check(L.pidm_linear_attention_out_forward(ptr(qd), ptr(wd), ptr(bd), None, ptr(yd), C, ptr(kstat), ptr(ctx), ptr(qstat), B, N, heads,
                                                ptr(ws), st))
    assert rel(yd, y_ref) < 5e-6
    dqkv2 = torch.empty(B, N, 3 * HD, device=dev)
    dw = torch.empty(C, HD, device=dev)
    L.check(L.pidm_linear_attention_out_backward(ptr(qd), ptr(kstat), ptr(qstat), ptr(ctx), ptr(gyd), C, ptr(wd), C, ptr(dqkv2), ptr(dw),
                                                 B, N, heads, ptr(ws), st))
    for c in range(3):
        sl = slice(c * HD, (c + 1) * HD)
        assert rel(dqkv2[..., sl], dq_ref[..., sl]) < 2e-5, "qkv"[c]
    assert rel(dw, torch.tensor(z["d_w_out"])) < 1e-5
    # the gradients the reference reports for the surrounding convolutions follow from dqkv by plain products
    d_x = dqkv2.cpu() @ w_qkv
    assert rel(d_x, torch.tensor(z["d_x"]).permute(0, 2, 3, 1).reshape(B, N, C)) < 2e-5
    d_wq = dqkv2.cpu().reshape(B * N, 3 * HD).t() @ xs.reshape(B * N, C)
    assert rel(d_wq, torch.tensor(z["d_w_qkv"])) < 2e-5
