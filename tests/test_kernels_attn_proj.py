"""Linear attention without a qkv tensor (csrc/k_attn_proj.hip: pidm_lap_forward / pidm_lap_backward) vs
  * golden g18 = the genuine SpatialLinearAttention module (to_qkv -> attention -> to_out, src/unet_model.py:281-299) with its
    autograd gradients of x and of both projections, and
  * the oracle's restatement on shapes of the Darcy model's 64x64 and 32x32 levels (C = 32 / 64, 8 heads).
`backend` = host-emulated build of the same sources (default run) or the gfx950 library (-m gpu)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pidm_oracle as O
from physicsinformeddiffusionmodels_amd._lib import ptr, stream_ptr

G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def run_lap(L, dev, xn, w_qkv, w_out, b_out, resid, gy, heads):
    """xn, resid, gy: [B,C,H,W] cpu; returns y, d_xn (NCHW), d_w_qkv, d_w_out"""
    st = stream_ptr(dev)
    B, C, H, W = xn.shape
    N = H * W
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(B, N, -1).contiguous().to(dev)
    xd, rd, gd = nhwc(xn), nhwc(resid), nhwc(gy)
    wq, wo, bo = w_qkv.contiguous().to(dev), w_out.contiguous().to(dev), b_out.contiguous().to(dev)
    y = torch.empty(B, N, C, device=dev)
    saved = torch.empty(L.pidm_lap_saved_floats(B, heads, C), device=dev)
    qstat = torch.empty(B * N * heads * 2, device=dev)
    ws = torch.empty(L.pidm_lap_ws(B, N, heads, C), dtype=torch.uint8, device=dev)
    L.check(L.pidm_lap_forward(ptr(xd), ptr(wq), ptr(wo), ptr(bo), ptr(rd), ptr(y), ptr(saved), ptr(qstat), C, B, N, heads, ptr(ws), st),
            "pidm_lap_forward")
    dxn = torch.empty(B, N, C, device=dev)
    dwq = torch.empty_like(wq)
    dwo = torch.empty_like(wo)
    L.check(L.pidm_lap_backward(ptr(xd), ptr(gd), ptr(wq), ptr(wo), ptr(saved), ptr(qstat), ptr(dxn), ptr(dwq), ptr(dwo), C, B, N, heads,
                                ptr(ws), st), "pidm_lap_backward")
    img = lambda t: t.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return img(y), img(dxn), dwq, dwo


def test_projected_attention_vs_reference_module(backend):
    """golden g18: y, d_x, d_w_qkv, d_w_out of the genuine module (B=2, 16x16, C=32, 2 heads; the module has no LayerNorm and no
    residual, so xn = x and the residual input is zero)."""
    L, dev = backend
    z = np.load(os.path.join(G, "g18_linear_attention.npz"))
    heads = int(z["heads"])
    t = lambda k: torch.tensor(z[k])
    x = t("x")
    y, dx, dwq, dwo = run_lap(L, dev, x, t("w_qkv"), t("w_out"), t("b_out"), torch.zeros_like(x), t("gy"), heads)
    assert rel(y, t("y")) < 5e-6
    assert rel(dx, t("d_x")) < 2e-5
    assert rel(dwq, t("d_w_qkv")) < 2e-5
    assert rel(dwo, t("d_w_out")) < 2e-5


CASES = [
    # B, H, heads, C
    (2, 16, 8, 32),     # N = 256, one pixel range
    (1, 32, 8, 64),     # 32x32 level of the Darcy model (C = 64): two 64-channel blocks per tile
    (1, 64, 8, 32),     # 64x64 level: eight forward ranges, 16 / 32 backward ranges, online rescaling across tiles
    (2, 8, 3, 32),      # N = 64, heads < 8: idle waves (a single range per workgroup: no tile groups in lap_bwd)
    (1, 64, 4, 32),     # attn_heads = 4 at the 64x64 level: lap_bwd runs two tile groups and stages two ranges at once
    (1, 32, 4, 64),     # ... at the 32x32 level: two tile groups over the two tiles of a C = 64 slab
    (2, 32, 2, 32),     # 2 heads, 1 head: still two groups (a slab holds two tiles), the other wave slots idle
    (1, 32, 1, 32),
]


@pytest.mark.parametrize("form", ["split", "split_fp32proj", "fp32"])
@pytest.mark.parametrize("B,H,heads,C", CASES)
def test_projected_attention_vs_oracle(backend, monkeypatch, form, B, H, heads, C):
    """form: the pixel sums of the forward / backward and (C = 32) the four per-pixel projections of the backward on the bf16 matrix pipe
    with 3-piece operands (default); the same with the projections on the fp32 MFMA; everything on the fp32 MFMA"""
    monkeypatch.setenv("PIDM_LAP_SPLIT", "0" if form == "fp32" else "1")
    monkeypatch.setenv("PIDM_LAP_SPLIT_PROJ", "0" if form == "split_fp32proj" else "1")
    L, dev = backend
    HD = heads * 32
    g = torch.Generator().manual_seed(5 + H + C)
    xn = torch.randn(B, C, H, H, generator=g)
    resid = torch.randn(B, C, H, H, generator=g)
    w_qkv = torch.randn(3 * HD, C, generator=g) * (2.0 / C ** 0.5)
    w_qkv[HD:2 * HD] *= 2.0                                  # spread the k logits: the running column max changes across tiles
    w_out = torch.randn(C, HD, generator=g) * 0.2
    b_out = torch.randn(C, generator=g)
    gy = torch.randn(B, C, H, H, generator=g)
    xr, wq, wo = xn.clone().requires_grad_(True), w_qkv.clone().requires_grad_(True), w_out.clone().requires_grad_(True)
    qkv = F.conv2d(xr, wq[:, :, None, None])
    ref = F.conv2d(O.linear_attention_core(qkv, heads, 32), wo[:, :, None, None], b_out) + resid
    gx, gwq, gwo = torch.autograd.grad(ref, (xr, wq, wo), gy)
    y, dx, dwq, dwo = run_lap(L, dev, xn, w_qkv, w_out, b_out, resid, gy, heads)
    assert rel(y, ref) < 1e-5
    assert rel(dx, gx) < 3e-5
    for c, name in enumerate("qkv"):                         # dWq, dWk, dWv have very different magnitudes
        sl = slice(c * HD, (c + 1) * HD)
        assert rel(dwq[sl], gwq[sl]) < 3e-5, name
    assert rel(dwo, gwo) < 3e-5
    if heads <= 4 and form == "split":
        # PIDM_LAP_GROUPS=1: lap_bwd with one wave per head and the other wave slots idle (the A/B knob) - same sums, another order
        monkeypatch.setenv("PIDM_LAP_GROUPS", "1")
        y1, dx1, dwq1, dwo1 = run_lap(L, dev, xn, w_qkv, w_out, b_out, resid, gy, heads)
        assert rel(y1, y) < 2e-6 and rel(dx1, dx) < 2e-6 and rel(dwq1, dwq) < 5e-6 and rel(dwo1, dwo) < 5e-6
