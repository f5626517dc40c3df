"""Darcy residual / adjoint / fused loss kernels (csrc/k_darcy.hip) run through the host emulator and
compared with the oracle.  CPU only; the same comparisons run on the real GPU in test_gpu_*.py."""
import numpy as np
import pytest
import torch

from oracle import pidm_oracle as O
from physicsinformeddiffusionmodels_amd._lib import ptr
from tests.emu_util import emu_lib


def rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize("P,B", [(16, 3), (64, 2)])
def test_darcy_residual_fwd_bwd(P, B):
    L = emu_lib()
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(B, 2, P, P, generator=g)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    fs = O.darcy_source_field(P).reshape(-1).contiguous()
    inv_h = float(P - 1)
    res = torch.empty(B, P * P, 3)
    L.check(L.pidm_darcy_residual_fwd(ptr(x0), ptr(fs), inv_h, -inv_h, ptr(res), B, P, None))
    xr = x0.clone().requires_grad_(True)
    ref = O.darcy_residual(xr)
    assert rel(res, ref.detach()) < 2e-6
    gr = torch.randn(B, P * P, 3, generator=g)
    (gref,) = torch.autograd.grad(ref, xr, gr)
    gx = torch.empty_like(x0)
    L.check(L.pidm_darcy_residual_bwd(ptr(x0), ptr(gr), inv_h, -inv_h, ptr(gx), B, P, None))
    assert rel(gx, gref) < 5e-6


@pytest.mark.parametrize("P,B", [(16, 4), (64, 2)])
def test_darcy_fused_loss(P, B):
    L = emu_lib()
    g = torch.Generator().manual_seed(6)
    tables = O.diffusion_tables(100)
    x0 = torch.randn(B, 2, P, P, generator=g)
    pred = (x0 + 0.3 * torch.randn(B, 2, P, P, generator=g))
    pred[:, 1] = torch.exp(0.5 * pred[:, 1])
    t = torch.tensor([0, 17, 63, 99][:B])
    p2w = tables["p2_loss_weight"][t].contiguous()
    inv_var = (1.0 / tables["posterior_variance_clipped"][t]).contiguous()
    fs = O.darcy_source_field(P).reshape(-1).contiguous()
    inv_h = float(P - 1)
    res = torch.empty(B, P * P, 3)
    gpred = torch.empty_like(pred)
    out = torch.zeros(4)
    ws = torch.empty(L.pidm_darcy_loss_ws(B, P), dtype=torch.uint8)
    L.check(L.pidm_darcy_loss_fwd_bwd(ptr(x0), ptr(pred), ptr(fs), ptr(p2w), ptr(inv_var), 1.0, 1e-3, inv_h, -inv_h,
                                      ptr(res), ptr(gpred), ptr(out), ptr(ws), B, P, None))
    pr = pred.clone().requires_grad_(True)
    loss, data, rabs, rref = O.darcy_loss_from_pred(tables, x0, pr, t, 1.0, 1e-3)
    loss.backward()
    assert rel(res, rref.detach()) < 2e-6
    assert abs(out[0].item() - loss.item()) < 1e-5 * abs(loss.item())
    assert abs(out[1].item() - data.item()) < 1e-5 * abs(data.item())
    assert abs(out[2].item() - rabs.item()) < 1e-5 * abs(rabs.item())
    assert rel(gpred, pr.grad) < 1e-5
