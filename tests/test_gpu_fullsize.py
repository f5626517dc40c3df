"""Full-size (BASELINE.json configs) checks on the real GPU through size-independent properties: analytic residuals,
adjoint (dot-product) identity, batch independence, run-to-run determinism, shard/average equivalence."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _darcy_setup(dim=32, P=64):
    from physicsinformeddiffusionmodels_amd.denoising_utils import DenoisingDiffusion
    from physicsinformeddiffusionmodels_amd.residuals_darcy import ResidualsDarcy
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D
    dev = _dev()
    torch.manual_seed(0)
    m = Unet3D(dim=dim, channels=2).to(dev)
    diff = DenoisingDiffusion(100, dev)
    res = ResidualsDarcy(model=m, fd_acc=2, pixels_per_dim=P, pixels_at_boundary=True, reverse_d1=True, device=dev)
    return m, diff, res, dev


def test_darcy_residual_analytic_quadratic_b256():
    """p = x0^2 + 2 x1^2 (exactly differentiated by the acc-2 stencils, edges included), K = const:
    eq = -K (2 + 4) - f_s everywhere; boundary rows carry -/+ dp/dx."""
    m, diff, res, dev = _darcy_setup()
    B, P = 256, 64
    xs = torch.linspace(0, 1, P, dtype=torch.float64)
    X0, X1 = torch.meshgrid(xs, xs, indexing="ij")
    x1c = 1.0 - X1   # reverse_d1: axis 1 runs with spacing -1/63
    p = (X0 ** 2 + 2 * x1c ** 2)
    K = 1.7
    x0 = torch.stack([p, torch.full_like(p, K)]).float().unsqueeze(0).repeat(B, 1, 1, 1).to(dev)
    r = res.compute_residual(x0, pass_through=True)["residual"].double().cpu().reshape(B, P, P, 3)
    fs = res.f_s.double().cpu().reshape(P, P)
    eq_ref = -K * 6.0 - fs
    assert (r[..., 0] - eq_ref).abs().max() < 2e-2          # fp32 cancellation at h^-2 = 3969: abs error, values ~10
    assert (r[3:-3, :, :, 0] - r[0, :, :, 0]).abs().max() == 0.0   # every sample identical (no cross-sample coupling)
    bc0 = r[0, :, :, 1]
    assert bc0[1:-1].abs().max() == 0.0
    assert (bc0[0] - (-2 * X0[0])).abs().max() < 1e-3 and (bc0[-1] - (2 * X0[-1])).abs().max() < 1e-3


def test_darcy_adjoint_dot_product_b256():
    """<J v, w> == <v, J^T w> with J the Jacobian of the residual (directional derivative by central difference in fp64
    on the host oracle is not needed: J v is obtained from the linearity in p for fixed K)."""
    m, diff, res, dev = _darcy_setup()
    B, P = 256, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 2, P, P, generator=g).to(dev)
    v = torch.zeros_like(x)
    v[:, 0] = torch.randn(B, P, P, generator=g).to(dev)      # perturb p only: residual is linear in p
    w = torch.randn(B, P * P, 3, generator=g).to(dev)
    xr = x.clone().requires_grad_(True)
    r0 = res.compute_residual(xr, pass_through=True)["residual"]
    (jtw,) = torch.autograd.grad(r0, xr, w)
    with torch.no_grad():
        r1 = res.compute_residual(x + v, pass_through=True)["residual"]
    lhs = ((r1 - r0.detach()).double() * w.double()).sum().item()
    rhs = (v.double() * jtw.double()).sum().item()
    assert abs(lhs - rhs) < 2e-4 * max(abs(lhs), abs(rhs))


def test_darcy_large_batch_kernels_b1024(monkeypatch):
    """Batches >= 512 run the fused loss as darcy_stream_kernel and the plain adjoint as darcy_full_kernel (one workgroup per
    sample): the oracle on the whole batch (north_star's 1e-5 on the loss terms), the band kernel bit for bit, the adjoint
    identity, and run-to-run determinism."""
    from oracle import pidm_oracle as O
    from physicsinformeddiffusionmodels_amd._lib import ptr, stream_ptr
    m, diff, res, dev = _darcy_setup()
    lib = res.lib
    B, P = 1024, 64
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(B, 2, P, P, generator=g)
    pred = x0 + 0.3 * torch.randn(B, 2, P, P, generator=g)
    pred[:, 1] = torch.exp(0.5 * pred[:, 1])
    t = torch.randint(0, 100, (B,), generator=g)
    w = torch.randn(B, P * P, 3, generator=g).to(dev)
    x0d, predd, td = x0.to(dev), pred.to(dev), t.to(dev)
    dd = diff.diff_dict

    def run():
        rbuf, gbuf, sc = torch.empty(B, P * P, 3, device=dev), torch.empty_like(predd), torch.empty(4, device=dev)
        ws = torch.empty(lib.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
        lib.check(lib.pidm_darcy_loss_fwd_bwd_t(ptr(x0d), ptr(predd), ptr(res._f_s_flat), ptr(td), ptr(dd['p2_loss_weight']),
                                                ptr(dd['posterior_variance_clipped']), 1.0, 1e-3, res.inv_h0, res.inv_h1, ptr(rbuf),
                                                ptr(gbuf), ptr(sc), ptr(ws), B, P, stream_ptr(dev)), "darcy loss")
        jtw = torch.empty_like(predd)
        lib.check(lib.pidm_darcy_residual_bwd(ptr(predd), ptr(w), res.inv_h0, res.inv_h1, ptr(jtw), B, P, stream_ptr(dev)), "adjoint")
        return rbuf, gbuf, sc.cpu(), jtw

    got = run()
    again = run()
    assert all(torch.equal(a, b) for a, b in zip(got, again))
    monkeypatch.setenv("PIDM_DARCY_FULL", "0")
    band = run()
    monkeypatch.delenv("PIDM_DARCY_FULL")
    assert torch.equal(got[0], band[0]) and torch.equal(got[1], band[1]) and torch.equal(got[3], band[3])
    assert torch.allclose(got[2], band[2], rtol=1e-6, atol=0)
    pr = pred.clone().requires_grad_(True)
    o_loss, o_data, o_rabs, o_res = O.darcy_loss_from_pred(O.diffusion_tables(100), x0, pr, t, 1., 1e-3)
    o_loss.backward()
    k_loss, k_data, k_rabs = (float(v) for v in got[2][:3])
    assert abs(k_loss - o_loss.item()) < 1e-5 * abs(o_loss.item())
    assert abs(k_data - o_data.item()) < 1e-5 * abs(o_data.item())
    assert abs(k_rabs - o_rabs.item()) < 1e-5 * abs(o_rabs.item())
    assert (got[0].cpu() - o_res.detach()).abs().max().item() < 2e-6 * o_res.abs().max().item()
    assert (got[1].cpu() - pr.grad).abs().max().item() < 1e-5 * pr.grad.abs().max().item()
    # adjoint identity on the kernels themselves: the residual is linear in p for fixed K
    v = torch.zeros_like(predd)
    v[:, 0] = torch.randn(B, P, P, generator=g).to(dev)
    r0, r1 = torch.empty(B, P * P, 3, device=dev), torch.empty(B, P * P, 3, device=dev)
    pv = (predd + v).contiguous()
    lib.check(lib.pidm_darcy_residual_fwd(ptr(predd), ptr(res._f_s_flat), res.inv_h0, res.inv_h1, ptr(r0), B, P, stream_ptr(dev)), "fwd")
    lib.check(lib.pidm_darcy_residual_fwd(ptr(pv), ptr(res._f_s_flat), res.inv_h0, res.inv_h1, ptr(r1), B, P, stream_ptr(dev)), "fwd")
    lhs = ((r1 - r0).double() * w.double()).sum().item()
    rhs = (v.double() * got[3].double()).sum().item()
    assert abs(lhs - rhs) < 2e-4 * max(abs(lhs), abs(rhs))


def test_unet_batch_independence_and_determinism_b64():
    m, diff, res, dev = _darcy_setup()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(64, 4096, 2, generator=g).to(dev)
    t = torch.randint(0, 100, (64,), generator=g).to(dev)
    with torch.no_grad():
        full = m(x, t)
        sub = m(x[5:9].contiguous(), t[5:9].contiguous())
        again = m(x, t)
    assert torch.equal(full, again)                               # bit-exact run to run
    assert (full[5:9] - sub).abs().max().item() <= 1e-5 * full.abs().max().item()   # samples do not interact


def test_training_step_determinism_and_shard_average_b64():
    """Two identical steps give bit-identical gradients (no atomics); the gradient of the batch-mean loss equals the
    average of the gradients of the two half-batch mean losses (what the data-parallel all-reduce computes)."""
    m, diff, res, dev = _darcy_setup()
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(64, 2, 64, 64, generator=g).to(dev)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = torch.randn(64, 2, 64, 64, generator=g).to(dev)
    t = torch.randint(0, 100, (64,), generator=g).to(dev)

    def grads(lo, hi):
        orig = torch.randint, torch.randn_like
        torch.randint = lambda *a, **k: t[lo:hi].clone()
        torch.randn_like = lambda *a, **k: eps[lo:hi].clone()
        try:
            loss, *_ = diff.model_estimation_loss(x0[lo:hi].contiguous(), residual_func=res, c_data=1., c_residual=1e-3)
        finally:
            torch.randint, torch.randn_like = orig
        for p in m.parameters():
            p.grad = None
        loss.backward()
        eng = next(iter(m.__dict__["_engines"].values()))
        return loss.item(), eng.flat_grad.clone()

    l1, g1 = grads(0, 64)
    l2, g2 = grads(0, 64)
    assert l1 == l2 and torch.equal(g1, g2)
    la, ga = grads(0, 32)
    lb, gb = grads(32, 64)
    avg = 0.5 * (ga + gb)
    assert abs(0.5 * (la + lb) - l1) < 1e-5 * abs(l1)
    scale = g1.abs().max().item()
    assert (avg - g1).abs().max().item() < 2e-4 * scale


def test_training_step_b256_equals_average_of_four_b64_shards():
    """The north-star configuration (per-GPU batch 256) through the whole step - q-sample, UNet forward, fused residual + loss,
    UNet backward: the gradient of the batch-256 mean loss equals the average of the gradients of its four batch-64 shards (the
    quantity the batch-64 parity tests pin against the reference goldens and what a 4-rank data-parallel step computes), the loss
    terms average the same way, and the step is bit-exact run to run."""
    m, diff, res, dev = _darcy_setup()
    g = torch.Generator().manual_seed(15)
    x0 = torch.randn(256, 2, 64, 64, generator=g).to(dev)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = torch.randn(256, 2, 64, 64, generator=g).to(dev)
    t = torch.randint(0, 100, (256,), generator=g).to(dev)

    def grads(lo, hi):
        orig = torch.randint, torch.randn_like
        torch.randint = lambda *a, **k: t[lo:hi].clone()
        torch.randn_like = lambda *a, **k: eps[lo:hi].clone()
        try:
            loss, data_l, res_l, _, _ = diff.model_estimation_loss(x0[lo:hi].contiguous(), residual_func=res, c_data=1., c_residual=1e-3)
        finally:
            torch.randint, torch.randn_like = orig
        for p in m.parameters():
            p.grad = None
        loss.backward()
        eng = next(iter(m.__dict__["_engines"].values()))
        return (loss.item(), float(data_l), float(res_l)), eng.flat_grad.clone()

    s_full, g_full = grads(0, 256)
    s_again, g_again = grads(0, 256)
    assert s_full == s_again and torch.equal(g_full, g_again)
    assert torch.isfinite(g_full).all() and g_full.abs().max().item() > 0
    parts = [grads(64 * k, 64 * (k + 1)) for k in range(4)]
    avg = sum(p[1] for p in parts) / 4.0
    for i in range(3):
        mean_i = sum(p[0][i] for p in parts) / 4.0
        assert abs(mean_i - s_full[i]) < 1e-5 * abs(s_full[i]), (i, mean_i, s_full[i])
    scale = g_full.abs().max().item()
    assert (avg - g_full).abs().max().item() < 2e-4 * scale


def test_attention_forms_agree_b64(monkeypatch):
    """Three executions of the same attention blocks at batch 64: without a qkv tensor (k_attn_proj.hip, the default of the 64x64
    and 32x32 levels), with the qkv tensor and the attention fused into the to_out projection, and with separate kernels.
    The engine's fused attention x to_out kernels (64x64 and 32x32 levels at batch 64: the heads*32-channel attention output
    and its gradient are never stored; dctx and dW_out come from one q x dY reduction) give the same loss and the same
    gradient - every parameter, in particular to_out.weight / to_out.bias / to_qkv.weight - as the separate attention and
    projection kernels, up to fp32 reassociation."""
    dev = _dev()
    g = torch.Generator().manual_seed(6)
    x0 = torch.randn(64, 2, 64, 64, generator=g).to(dev)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = torch.randn(64, 2, 64, 64, generator=g).to(dev)
    t = torch.randint(0, 100, (64,), generator=g).to(dev)

    def grads(env):
        # one engine per setting (same seed-0 parameters): the activation arena is sized for the path chosen at first use
        for k in ("PIDM_NO_LAP", "PIDM_LA_FUSED_MIN_WGS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m, diff, res, _ = _darcy_setup()
        orig = torch.randint, torch.randn_like
        torch.randint = lambda *a, **k: t.clone()
        torch.randn_like = lambda *a, **k: eps.clone()
        try:
            loss, *_ = diff.model_estimation_loss(x0, residual_func=res, c_data=1., c_residual=1e-3)
        finally:
            torch.randint, torch.randn_like = orig
        for p in m.parameters():
            p.grad = None
        loss.backward()
        eng = next(iter(m.__dict__["_engines"].values()))
        per = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None and ("to_out" in n or "to_qkv" in n)}
        return loss.item(), eng.flat_grad.clone(), per

    ls, gs, ps = grads({"PIDM_NO_LAP": "1", "PIDM_LA_FUSED_MIN_WGS": "1000000000"})     # qkv tensor, separate kernels
    scale = gs.abs().max().item()
    for env in ({"PIDM_NO_LAP": "1", "PIDM_LA_FUSED_MIN_WGS": "1"},       # qkv tensor, fused wherever eligible (64x64, 32x32, 16x16)
                {}):                                                         # default: no qkv tensor at 64x64 / 32x32
        lf, gf, pf = grads(env)
        assert abs(lf - ls) < 1e-5 * abs(ls), env
        assert (gf - gs).abs().max().item() < 1e-4 * scale, env
        assert len(pf) >= 16
        for n in ps:                     # the tensors the forms compute differently, each against its own scale
            assert (pf[n] - ps[n]).abs().max().item() < 2e-4 * max(ps[n].abs().max().item(), 1e-12), (env, n)


def test_sampling_b1024_two_steps_finite_and_deterministic():
    m, diff, res, dev = _darcy_setup()
    torch.manual_seed(11)
    x = torch.randn(1024, 2, 64, 64, device=dev)
    outs = []
    for rep in range(2):
        torch.manual_seed(12)
        (nx, _), _ = diff.p_sample(x, None, 57, save_output=False, surpress_noise=True, residual_func=res)
        outs.append(nx)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])


def test_mechanics_config4_step_dim128_b32_deterministic():
    """BASELINE configs[3] per-GPU shape: Unet3D(dim=128, channels=10, out_dim=3, sigmoid), batch 32, 65x65 fields, the full
    mechanics loss (residual + inequality + compliance terms).  Two identical steps give bit-identical loss and gradients;
    every used gradient is finite and non-trivial; unused parameters keep grad None."""
    from physicsinformeddiffusionmodels_amd.denoising_utils import DenoisingDiffusion
    from physicsinformeddiffusionmodels_amd.residuals_mechanics_K import ResidualsMechanics
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D
    dev = _dev()
    torch.manual_seed(0)
    m = Unet3D(dim=128, channels=10, out_dim=3, sigmoid_last_channel=True).to(dev)
    diff = DenoisingDiffusion(100, dev)
    res = ResidualsMechanics(model=m, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder="/nonexistent/", device=dev)
    B = 32
    g = torch.Generator().manual_seed(8)
    inp = torch.zeros(B, 10, 65, 65)
    inp[:, 0] = torch.rand(B, generator=g).view(B, 1, 1) * 0.3 + 0.2
    inp[:, 1:3] = torch.randn(B, 2, 65, 65, generator=g)
    inp[:, 3:5] = 0.1 * torch.randn(B, 2, 65, 65, generator=g)
    inp[:, 5, :64, :64] = torch.rand(B, 64, 64, generator=g)
    inp[:, 6, :, 0] = 1.0
    inp[:, 7, :, 0] = 1.0
    inp[:, 9, 32, 64] = -1.0
    inp = inp.to(dev)
    eps = torch.randn(B, 3, 65, 65, generator=g).to(dev)
    t = torch.randint(0, 100, (B,), generator=g).to(dev)

    def step():
        orig = torch.randint, torch.randn_like
        torch.randint = lambda *a, **k: t.clone()
        torch.randn_like = lambda *a, **k: eps.clone()
        try:
            loss, data_l, res_l, ineq_l, opt_l = diff.model_estimation_loss(inp, residual_func=res, c_data=1., c_residual=1e-3,
                                                                            c_ineq=0.5, lambda_opt=0.01)
        finally:
            torch.randint, torch.randn_like = orig
        for p in m.parameters():
            p.grad = None
        loss.backward()
        eng = next(iter(m.__dict__["_engines"].values()))
        return loss.item(), (data_l, res_l, ineq_l, opt_l), eng.flat_grad.clone()

    l1, parts1, g1 = step()
    l2, parts2, g2 = step()
    assert l1 == l2 and parts1 == parts2 and torch.equal(g1, g2)
    assert torch.isfinite(g1).all() and g1.abs().max().item() > 0
    assert all(np.isfinite(v) for v in parts1) and parts1[2] > 0 and parts1[3] != 0
    n_used = sum(p.grad is not None for p in m.parameters())
    n_all = sum(1 for _ in m.parameters())
    assert n_used == 259 and n_all > n_used


@pytest.mark.slow
def test_darcy_step_b64_vs_oracle():
    """BASELINE configs[1] at its real batch (64): the large-grid tile variants (persistent 3x3 walk, 256-pixel tiles, permuted
    128-channel 1x1 tiles, fused attention - all gated on >= 512 workgroups and therefore only reached by small-shape unit
    tests otherwise) against the oracle's autograd on the same weights, (t, eps) and data: loss, data loss, mean |residual|
    and EVERY used gradient tensor (norm 1e-3 + element-wise on 24 probes).  ~30 s of host time for the oracle."""
    from oracle import pidm_oracle as O
    m, diff, res, dev = _darcy_setup()
    B, P = 64, 64
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(B, 2, P, P, generator=g)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = torch.randn(B, 2, P, P, generator=g)
    t = torch.randint(0, 100, (B,), generator=g)
    orig = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t.to(dev)
    torch.randn_like = lambda *a, **k: eps.to(dev)
    try:
        loss, data_l, res_l, _, _ = diff.model_estimation_loss(x0.to(dev), residual_func=res, c_data=1., c_residual=1e-3)
    finally:
        torch.randint, torch.randn_like = orig
    loss.backward()
    torch.cuda.synchronize()
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref, rdata, rabs, _ = O.darcy_training_loss(p, O.UnetCfg(dim=32, channels=2), O.diffusion_tables(100), x0, t, eps, 1., 1e-3)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item()), (loss.item(), ref.item())
    assert abs(data_l - rdata.item()) < 1e-4 * abs(rdata.item())
    # end to end each side's residual sees ITS OWN UNet output (38 convolutions deep, 3e-5 apart): 1e-4 is the floor of that
    # comparison, not of the residual kernels ...
    assert abs(res_l - rabs.item()) < 1e-4 * abs(rabs.item())
    print("full step: mean|residual| engine / oracle - 1 =", res_l / rabs.item() - 1.0)
    # ... whose own error is what BASELINE.json's north_star bounds (residual-loss parity <= 1e-5 rel): the fused residual + loss
    # kernel and the oracle's residual on the SAME prediction (the engine's x0_pred of this very step), at the full batch
    from physicsinformeddiffusionmodels_amd._lib import ptr, stream_ptr
    lib = res.lib
    with torch.no_grad():
        was = m.training
        m.eval()
        xt = O.q_sample(O.diffusion_tables(100), x0, t, eps).to(dev)
        pred = m(xt.permute(0, 2, 3, 1).reshape(B, P * P, 2).contiguous(), t.to(dev)).contiguous()
        m.train(was)
    dd = diff.diff_dict
    rbuf = torch.empty(B, P * P, 3, device=dev)
    gbuf = torch.empty_like(pred)
    sc = torch.empty(4, device=dev)
    ws = torch.empty(lib.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
    x0d, td = x0.to(dev).contiguous(), t.to(dev).contiguous()
    lib.check(lib.pidm_darcy_loss_fwd_bwd_t(ptr(x0d), ptr(pred), ptr(res._f_s_flat), ptr(td), ptr(dd['p2_loss_weight']),
                                            ptr(dd['posterior_variance_clipped']), 1.0, 1e-3, res.inv_h0, res.inv_h1, ptr(rbuf), ptr(gbuf),
                                            ptr(sc), ptr(ws), B, P, stream_ptr(dev)), "darcy loss")
    k_loss, k_data, k_rabs = (float(v) for v in sc[:3].cpu())
    o_loss, o_data, o_rabs, _ = O.darcy_loss_from_pred(O.diffusion_tables(100), x0, pred.cpu(), t, 1., 1e-3)
    print("same prediction: loss / data / mean|residual| relative differences",
          k_loss / o_loss.item() - 1, k_data / o_data.item() - 1, k_rabs / o_rabs.item() - 1)
    assert abs(k_rabs - o_rabs.item()) < 1e-5 * abs(o_rabs.item())
    assert abs(k_loss - o_loss.item()) < 1e-5 * abs(o_loss.item())
    assert abs(k_data - o_data.item()) < 1e-5 * abs(o_data.item())
    gmax = max(v.grad.norm().item() for v in p.values() if v.grad is not None)
    n, probes, bad = 0, 0, []
    for k, prm in m.named_parameters():
        if p[k].grad is None:
            assert prm.grad is None, k
            continue
        a = prm.grad.detach().cpu()
        b = p[k].grad
        if not abs(a.norm().item() - b.norm().item()) <= 1e-3 * b.norm().item() + 2e-6 * gmax:
            bad.append((k, a.norm().item(), b.norm().item()))
        n += 1
        if n % 11 == 0:       # element-wise on every 11th tensor (24 of 259)
            assert (a - b).abs().max().item() <= 2e-3 * b.abs().max().item() + 2e-6 * gmax, k
            probes += 1
    assert not bad, bad[:8]
    assert n == 259 and probes >= 20


def _step_grads(m, diff, res, x0, eps, t, lo, hi):
    """model_estimation_loss + backward on samples lo..hi-1 with pinned (t, eps): (loss, flat gradient copy)."""
    orig = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t[lo:hi].clone()
    torch.randn_like = lambda *a, **k: eps[lo:hi].clone()
    try:
        loss, *_ = diff.model_estimation_loss(x0[lo:hi].contiguous(), residual_func=res, c_data=1., c_residual=1e-3)
    finally:
        torch.randint, torch.randn_like = orig
    for p in m.parameters():
        p.grad = None
    loss.backward()
    eng = next(iter(m.__dict__["_engines"].values()))
    return loss.item(), eng.flat_grad.clone()


@pytest.mark.parametrize("B,cut", [(1, 0), (3, 1), (17, 5), (37, 16), (65, 64), (100, 37), (129, 1)])
def test_ragged_batches_full_size(B, cut):
    """The last batch of an epoch is ragged (main.py:116: DataLoader without drop_last) and every batch size picks its own tile
    variants (row-streaming rows per workgroup, 128- / 256-pixel tiles, grouped weight-gradient splits, GroupNorm block plan):
    at the full model size a batch of B samples must give (i) per-sample outputs equal to the ones the same samples get inside
    other batches, (ii) bit-identical gradients run to run, (iii) the gradient of its mean loss = the sample-weighted average of
    the gradients of the two pieces [0, cut) and [cut, B) - pieces that run through different variants again."""
    m, diff, res, dev = _darcy_setup()
    g = torch.Generator().manual_seed(100 + B)
    x0 = torch.randn(B, 2, 64, 64, generator=g).to(dev)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = torch.randn(B, 2, 64, 64, generator=g).to(dev)
    t = torch.randint(0, 100, (B,), generator=g).to(dev)
    x = x0.permute(0, 2, 3, 1).reshape(B, 4096, 2).contiguous()
    with torch.no_grad():
        full = m(x, t)
        one = torch.cat([m(x[i:i + 1].contiguous(), t[i:i + 1].contiguous()) for i in sorted({0, B // 2, B - 1})])
    pick = full[sorted({0, B // 2, B - 1})]
    assert torch.isfinite(full).all()
    assert (pick - one).abs().max().item() <= 1e-5 * full.abs().max().item()
    l1, g1 = _step_grads(m, diff, res, x0, eps, t, 0, B)
    l2, g2 = _step_grads(m, diff, res, x0, eps, t, 0, B)
    assert l1 == l2 and torch.equal(g1, g2)
    assert torch.isfinite(g1).all()
    if cut:
        la, ga = _step_grads(m, diff, res, x0, eps, t, 0, cut)
        lb, gb = _step_grads(m, diff, res, x0, eps, t, cut, B)
        wa, wb = cut / B, (B - cut) / B
        assert abs(wa * la + wb * lb - l1) < 1e-5 * abs(l1)
        assert (wa * ga + wb * gb - g1).abs().max().item() < 2e-4 * g1.abs().max().item()


@pytest.mark.parametrize("B", [1, 3])
def test_small_ragged_batch_step_vs_oracle(B):
    """Batches of 1 and 3 at the full model size against the oracle's autograd (every used gradient tensor)."""
    from oracle import pidm_oracle as O
    m, diff, res, dev = _darcy_setup()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(40 + B)
    x0 = torch.randn(B, 2, 64, 64, generator=g)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = torch.randn(B, 2, 64, 64, generator=g)
    t = torch.randint(0, 100, (B,), generator=g)
    loss, _ = _step_grads(m, diff, res, x0.to(dev), eps.to(dev), t.to(dev), 0, B)
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    ref, _, _, _ = O.darcy_training_loss(p, O.UnetCfg(dim=32, channels=2), O.diffusion_tables(100), x0, t, eps, 1., 1e-3)
    ref.backward()
    assert abs(loss - ref.item()) < 1e-4 * abs(ref.item()), (loss, ref.item())
    gmax = max(v.grad.norm().item() for v in p.values() if v.grad is not None)
    n = 0
    for k, prm in m.named_parameters():
        if p[k].grad is None:
            assert prm.grad is None, k
            continue
        a, b = prm.grad.detach().cpu(), p[k].grad
        assert abs(a.norm().item() - b.norm().item()) <= 1e-3 * b.norm().item() + 2e-6 * gmax, (k, a.norm().item(), b.norm().item())
        n += 1
    assert n == 259


def test_ragged_batch_sampling_step_b7():
    """One ancestral step on 7 samples = the same step on each sample alone with the same noise (torch.randn_like pinned: the
    update is then a function of the sample only); reference: src/denoising_utils.py:388-455."""
    m, diff, res, dev = _darcy_setup()
    torch.manual_seed(21)
    x = torch.randn(7, 2, 64, 64, device=dev)
    z = torch.randn(7, 2, 64, 64, device=dev)

    def step(lo, hi):
        orig = torch.randn_like
        torch.randn_like = lambda a, *aa, **k: z[lo:hi].clone().view(a.shape)
        try:
            (nx, _), _ = diff.p_sample(x[lo:hi].contiguous(), None, 57, save_output=False, surpress_noise=True, residual_func=res)
        finally:
            torch.randn_like = orig
        return nx

    nx = step(0, 7)
    assert torch.isfinite(nx).all()
    assert torch.equal(nx, step(0, 7))
    for i in (0, 3, 6):
        assert (step(i, i + 1)[0] - nx[i]).abs().max().item() <= 1e-5 * nx.abs().max().item()


def test_ragged_batch_mechanics_dim128_b5():
    """The mechanics model (dim 128, 65x65 fields) on a ragged batch of 5 without the sample-coupling inequality term: run-to-run
    bit-identical, and the gradient of the mean loss = the sample-weighted average over the pieces [0, 2) and [2, 5)."""
    from physicsinformeddiffusionmodels_amd.denoising_utils import DenoisingDiffusion
    from physicsinformeddiffusionmodels_amd.residuals_mechanics_K import ResidualsMechanics
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D
    dev = _dev()
    torch.manual_seed(0)
    m = Unet3D(dim=128, channels=10, out_dim=3, sigmoid_last_channel=True).to(dev)
    diff = DenoisingDiffusion(100, dev)
    res = ResidualsMechanics(model=m, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder="/nonexistent/", device=dev)
    B = 5
    g = torch.Generator().manual_seed(9)
    inp = torch.zeros(B, 10, 65, 65)
    inp[:, 0] = torch.rand(B, generator=g).view(B, 1, 1) * 0.3 + 0.2
    inp[:, 1:3] = torch.randn(B, 2, 65, 65, generator=g)
    inp[:, 3:5] = 0.1 * torch.randn(B, 2, 65, 65, generator=g)
    inp[:, 5, :64, :64] = torch.rand(B, 64, 64, generator=g)
    inp[:, 6, :, 0] = 1.0
    inp[:, 7, :, 0] = 1.0
    inp[:, 9, 32, 64] = -1.0
    inp = inp.to(dev)
    eps = torch.randn(B, 3, 65, 65, generator=g).to(dev)
    t = torch.randint(0, 100, (B,), generator=g).to(dev)

    def step(lo, hi):
        orig = torch.randint, torch.randn_like
        torch.randint = lambda *a, **k: t[lo:hi].clone()
        torch.randn_like = lambda *a, **k: eps[lo:hi].clone()
        try:
            loss, *_ = diff.model_estimation_loss(inp[lo:hi].contiguous(), residual_func=res, c_data=1., c_residual=1e-3)
        finally:
            torch.randint, torch.randn_like = orig
        for p in m.parameters():
            p.grad = None
        loss.backward()
        eng = next(iter(m.__dict__["_engines"].values()))
        return loss.item(), eng.flat_grad.clone()

    l1, g1 = step(0, B)
    l2, g2 = step(0, B)
    assert l1 == l2 and torch.equal(g1, g2) and torch.isfinite(g1).all()
    la, ga = step(0, 2)
    lb, gb = step(2, B)
    assert abs(0.4 * la + 0.6 * lb - l1) < 1e-5 * abs(l1)
    assert (0.4 * ga + 0.6 * gb - g1).abs().max().item() < 2e-4 * g1.abs().max().item()


@pytest.mark.parametrize("B", [513, 777])
def test_darcy_stream_kernel_ragged_batches(B, monkeypatch):
    """Batches that do not divide by the 256 persistent workgroups of darcy_stream_kernel (some walk one sample more than others;
    the last round prefetches nothing): residual and gradient bit-identical to the band kernel, loss scalars to 1e-6."""
    from physicsinformeddiffusionmodels_amd._lib import ptr, stream_ptr
    m, diff, res, dev = _darcy_setup()
    lib = res.lib
    P = 64
    g = torch.Generator().manual_seed(B)
    x0 = torch.randn(B, 2, P, P, generator=g)
    pred = x0 + 0.3 * torch.randn(B, 2, P, P, generator=g)
    pred[:, 1] = torch.exp(0.5 * pred[:, 1])
    t = torch.randint(0, 100, (B,), generator=g)
    x0d, predd, td = x0.to(dev), pred.to(dev), t.to(dev)
    dd = diff.diff_dict

    def run():
        rbuf, gbuf, sc = torch.empty(B, P * P, 3, device=dev), torch.empty_like(predd), torch.empty(4, device=dev)
        ws = torch.empty(lib.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
        lib.check(lib.pidm_darcy_loss_fwd_bwd_t(ptr(x0d), ptr(predd), ptr(res._f_s_flat), ptr(td), ptr(dd['p2_loss_weight']),
                                                ptr(dd['posterior_variance_clipped']), 1.0, 1e-3, res.inv_h0, res.inv_h1, ptr(rbuf),
                                                ptr(gbuf), ptr(sc), ptr(ws), B, P, stream_ptr(dev)), "darcy loss")
        return rbuf, gbuf, sc.cpu()

    got = run()
    monkeypatch.setenv("PIDM_DARCY_FULL", "0")
    band = run()
    assert torch.isfinite(got[0]).all() and torch.isfinite(got[1]).all()
    assert torch.equal(got[0], band[0]) and torch.equal(got[1], band[1])
    assert torch.allclose(got[2], band[2], rtol=1e-6, atol=0)
