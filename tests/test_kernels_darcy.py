"""Darcy residual / adjoint / fused-loss kernels (csrc/k_darcy.hip) vs the oracle.
`backend` = host-emulated build of the same sources on CPU (default run) or the real gfx950 library
through the C ABI (-m gpu)."""
import pytest
import torch

from oracle import pidm_oracle as O
from physicsinformeddiffusionmodels_amd._lib import ptr, stream_ptr


def rel(a, b):
    a, b = a.detach().cpu(), b.detach().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


# (P, B) also pick the row-band plan of the kernel: 64 -> 16 bands of 4 rows, 16 -> 4 x 4, 10 -> 2 x 5 (edge stencils reach across
# a band border), 7 -> one band, 21 -> 5 bands the last of which is a single row
# P / 4 a power of two takes the four-pixels-per-thread kernel (8: every quad holds an edge pixel), the others the one-pixel kernel
@pytest.mark.parametrize("P,B", [(16, 3), (64, 2), (10, 3), (7, 2), (21, 2), (8, 2), (32, 1), (12, 2)])
def test_darcy_residual_fwd_bwd(backend, P, B):
    L, dev = backend
    st = stream_ptr(dev)
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(B, 2, P, P, generator=g)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    fs = O.darcy_source_field(P).reshape(-1).contiguous()
    inv_h = float(P - 1)
    x0d, fsd = x0.to(dev), fs.to(dev)
    res = torch.empty(B, P * P, 3, device=dev)
    L.check(L.pidm_darcy_residual_fwd(ptr(x0d), ptr(fsd), inv_h, -inv_h, ptr(res), B, P, st))
    xr = x0.clone().requires_grad_(True)
    ref = O.darcy_residual(xr)
    assert rel(res, ref) < 2e-6
    gr = torch.randn(B, P * P, 3, generator=g)
    (gref,) = torch.autograd.grad(ref, xr, gr)
    grd = gr.to(dev)
    gx = torch.empty_like(x0d)
    L.check(L.pidm_darcy_residual_bwd(ptr(x0d), ptr(grd), inv_h, -inv_h, ptr(gx), B, P, st))
    assert rel(gx, gref) < 5e-6


@pytest.mark.parametrize("P,B", [(16, 4), (64, 2), (10, 3), (21, 4), (8, 3)])
def test_darcy_fused_loss(backend, P, B):
    L, dev = backend
    st = stream_ptr(dev)
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
    x0d, predd, fsd, p2wd, ivd = (z.to(dev) for z in (x0, pred, fs, p2w, inv_var))
    res = torch.empty(B, P * P, 3, device=dev)
    gpred = torch.empty_like(predd)
    out = torch.zeros(4, device=dev)
    ws = torch.empty(L.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
    L.check(L.pidm_darcy_loss_fwd_bwd(ptr(x0d), ptr(predd), ptr(fsd), ptr(p2wd), ptr(ivd), 1.0, 1e-3, inv_h, -inv_h,
                                      ptr(res), ptr(gpred), ptr(out), ptr(ws), B, P, st))
    pr = pred.clone().requires_grad_(True)
    loss, data, rabs, rref = O.darcy_loss_from_pred(tables, x0, pr, t, 1.0, 1e-3)
    loss.backward()
    out = out.cpu()
    assert rel(res, rref) < 2e-6
    assert abs(out[0].item() - loss.item()) < 1e-5 * abs(loss.item())
    assert abs(out[1].item() - data.item()) < 1e-5 * abs(data.item())
    assert abs(out[2].item() - rabs.item()) < 1e-5 * abs(rabs.item())
    assert rel(gpred, pr.grad) < 1e-5
    # the variant that looks the weights up from the schedule tables by t inside the kernel: bit-identical
    res2, gpred2, out2 = torch.empty_like(res), torch.empty_like(gpred), torch.zeros(4, device=dev)
    td = t.to(dev)
    tab_w, tab_v = tables["p2_loss_weight"].to(dev), tables["posterior_variance_clipped"].to(dev)
    L.check(L.pidm_darcy_loss_fwd_bwd_t(ptr(x0d), ptr(predd), ptr(fsd), ptr(td), ptr(tab_w), ptr(tab_v), 1.0, 1e-3, inv_h, -inv_h,
                                        ptr(res2), ptr(gpred2), ptr(out2), ptr(ws), B, P, st))
    assert torch.equal(res2, res) and torch.equal(gpred2, gpred) and torch.equal(out2.cpu(), out)


def test_qsample_table_lookup_matches_gathered_form(backend):
    L, dev = backend
    st = stream_ptr(dev)
    g = torch.Generator().manual_seed(8)
    tables = O.diffusion_tables(100)
    B, C, P = 5, 2, 8
    x0, eps = torch.randn(B, C, P, P, generator=g).to(dev), torch.randn(B, C, P, P, generator=g).to(dev)
    t = torch.tensor([0, 99, 3, 50, 50]).to(dev)
    a_tab, am1_tab = tables["alphas_bar_sqrt"].to(dev), tables["one_minus_alphas_bar_sqrt"].to(dev)
    a, am1 = a_tab[t].contiguous(), am1_tab[t].contiguous()
    y1, y2 = torch.empty(B, P * P, C, device=dev), torch.empty(B, P * P, C, device=dev)
    L.check(L.pidm_qsample_nhwc(ptr(x0), ptr(eps), ptr(a), ptr(am1), ptr(y1), B, C, P * P, st))
    L.check(L.pidm_qsample_nhwc_t(ptr(x0), ptr(eps), ptr(t), ptr(a_tab), ptr(am1_tab), ptr(y2), B, C, P * P, st))
    assert torch.equal(y1, y2)
    ref = (a.view(B, 1, 1, 1) * x0 + am1.view(B, 1, 1, 1) * eps).permute(0, 2, 3, 1).reshape(B, P * P, C)
    assert rel(y1, ref) < 1e-6


def test_darcy_one_pixel_kernel_at_64(backend, monkeypatch):
    """PIDM_DARCY_QUAD=0 keeps the one-pixel-per-thread kernel reachable at the sizes the quad kernel normally takes."""
    monkeypatch.setenv("PIDM_DARCY_QUAD", "0")
    test_darcy_residual_fwd_bwd(backend, 64, 2)
    test_darcy_fused_loss(backend, 16, 4)



@pytest.mark.parametrize("variant", ["stream", "stream_2wgs", "full"])
def test_darcy_one_workgroup_per_sample_kernels_are_bit_identical(backend, monkeypatch, variant):
    """Batches >= 512 take one workgroup per 64 x 64 sample (row neighbours by shuffles) - the fused loss as a persistent workgroup
    per CU that has the next sample copied into LDS while it computes (darcy_stream_kernel), the plain adjoint as
    darcy_full_kernel; PIDM_DARCY_FULL lowers the batch threshold.  Same taps in the same order: residual and gradients
    bit-identical to the band kernel; the loss scalars group the same doubles per sample instead of per band."""
    L, dev = backend
    st = stream_ptr(dev)
    g = torch.Generator().manual_seed(16)
    tables = O.diffusion_tables(100)
    B, P = 5, 64
    x0 = torch.randn(B, 2, P, P, generator=g)
    pred = (x0 + 0.3 * torch.randn(B, 2, P, P, generator=g))
    pred[:, 1] = torch.exp(0.5 * pred[:, 1])
    t = torch.tensor([5, 99, 40, 0, 71])
    fs = O.darcy_source_field(P).reshape(-1).contiguous()
    inv_h = float(P - 1)
    x0d, predd, fsd, td = (z.to(dev) for z in (x0, pred, fs, t))
    tab_w, tab_v = tables["p2_loss_weight"].to(dev), tables["posterior_variance_clipped"].to(dev)
    gr = torch.randn(B, P * P, 3, generator=g).to(dev)

    def run():
        res, gpred, out = torch.empty(B, P * P, 3, device=dev), torch.empty_like(predd), torch.zeros(4, device=dev)
        ws = torch.empty(L.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
        L.check(L.pidm_darcy_loss_fwd_bwd_t(ptr(x0d), ptr(predd), ptr(fsd), ptr(td), ptr(tab_w), ptr(tab_v), 1.0, 1e-3, inv_h, -inv_h,
                                            ptr(res), ptr(gpred), ptr(out), ptr(ws), B, P, st))
        gx = torch.empty_like(predd)
        L.check(L.pidm_darcy_residual_bwd(ptr(predd), ptr(gr), inv_h, -inv_h, ptr(gx), B, P, st))
        return res.cpu(), gpred.cpu(), out.cpu(), gx.cpu()

    monkeypatch.setenv("PIDM_DARCY_FULL", "0")
    ref = run()
    monkeypatch.setenv("PIDM_DARCY_FULL", "2")
    if variant == "stream_2wgs":
        monkeypatch.setenv("PIDM_DARCY_STREAM_WGS", "2")     # workgroup 0 walks samples 0, 2, 4; workgroup 1 samples 1, 3
    elif variant == "full":
        monkeypatch.setenv("PIDM_DARCY_STREAM", "0")
    got = run()
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[3], ref[3])
    assert torch.allclose(got[2], ref[2], rtol=1e-6, atol=0)
