"""Generate tests/golden/*.npz by RUNNING the genuine reference (/root/reference) in this container.

Usage (authoring container only; the reference does not exist on the GPU box):
    cd /root/repo && python oracle/make_golden.py

The reference needs 7 third-party modules that are not installed here; `oracle/shims/` provides
stand-ins (see oracle/shims/README.md for which of them carry arithmetic and are therefore
"parity unpinned").  Fixtures are DATA ONLY: seeded inputs + the reference's outputs.  Model weights
are not stored: every parameter is formula-filled from its state_dict name
(oracle/pidm_oracle.formula_fill), so the tests can rebuild the identical weights.
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
# The reference's `src` is a namespace package (no __init__.py); this repo's drop-in `src/` is a regular package
# and would shadow it from ANY position on sys.path - so the repo root is NOT put on the path here and the oracle
# helper is loaded by file name.
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("pidm_oracle", os.path.join(HERE, "pidm_oracle.py"))
_oracle = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_oracle)
fill_state_dict = _oracle.fill_state_dict
O = _oracle

# --- genuine reference imports -------------------------------------------------------------------
from src.unet_model import Unet3D  # noqa: E402
from src.denoising_utils import DenoisingDiffusion  # noqa: E402
from src.residuals_darcy import ResidualsDarcy  # noqa: E402
from src.grad_utils import GradientsHelper  # noqa: E402
import src.denoising_utils as du  # noqa: E402

assert du.__file__.startswith("/root/reference/"), "golden vectors must come from the genuine reference: " + du.__file__

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# G1: schedule tables ------------------------------------------------------------------------------
def g1():
    out = {}
    for n in (100, 1000):
        dd = DenoisingDiffusion(n, "cpu").diff_dict
        for k, v in dd.items():
            out[f"n{n}/{k}"] = npy(v)
    np.savez_compressed(os.path.join(OUT, "g1_schedule.npz"), **out)


# G2/G3/G4: stencil engine + Darcy residual --------------------------------------------------------
def g2_g3():
    P = 64
    h = 1.0 / (P - 1)
    gh = GradientsHelper(d0=h, d1=-h, fd_acc=2, periodic=False, device="cpu")
    x8 = seeded((2, 8, 8), 11)
    gh8 = GradientsHelper(d0=0.25, d1=-0.25, fd_acc=2, periodic=False, device="cpu")
    out = {"x8": npy(x8)}
    for mode in ("d_d0", "d_d1", "d_d00", "d_d11"):
        out[f"x8/{mode}"] = npy(gh8.stencil_gradients(x8, mode=mode))
    x64 = seeded((1, 64, 64), 12)
    out["x64"] = npy(x64)
    for mode in ("d_d0", "d_d1", "d_d00", "d_d11"):
        out[f"x64/{mode}"] = npy(gh.stencil_gradients(x64, mode=mode))
    np.savez_compressed(os.path.join(OUT, "g2_stencils.npz"), **out)

    res = ResidualsDarcy(model=None, fd_acc=2, pixels_per_dim=P, pixels_at_boundary=True, reverse_d1=True,
                         device="cpu", bcs="none", domain_length=1.0)
    x0 = seeded((2, 2, P, P), 13)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    x0[:, 0] *= 0.1
    x0.requires_grad_(True)
    r = res.compute_residual(x0, pass_through=True)["residual"]
    (g,) = torch.autograd.grad((r ** 2).sum(), x0)
    np.savez_compressed(os.path.join(OUT, "g3_darcy_residual.npz"), x0=npy(x0), residual=npy(r),
                        grad_sumsq=npy(g), f_s=npy(res.f_s), trap=npy(res.trapezoidal_weights))


# G5/G6: UNet forward + grads with formula-filled weights -----------------------------------------
PROBES = ["init_conv.weight", "downs.0.0.block1.proj.weight", "downs.1.0.block1.proj.weight",
          "downs.1.0.res_conv.weight", "downs.0.2.fn.fn.to_qkv.weight", "downs.0.2.fn.fn.to_out.weight",
          "downs.0.2.fn.norm.gamma", "downs.0.3.weight", "ups.0.3.weight", "time_mlp.3.weight",
          "time_mlp.1.weight", "downs.0.0.mlp.1.weight", "mid_spatial_attn.fn.fn.fn.to_qkv.weight",
          "mid_spatial_attn.fn.fn.fn.to_out.weight", "mid_spatial_attn.fn.norm.gamma",
          "downs.0.0.block1.norm.weight", "downs.0.0.block1.norm.bias", "final_conv.1.weight",
          "final_conv.0.res_conv.weight", "ups.3.0.block1.proj.weight", "mid_block1.block2.proj.weight"]


def unet_case(tag, dim, P, B, tvals, channels=2, out_dim=None, sigmoid=False, full_out=True):
    torch.manual_seed(0)
    m = Unet3D(dim=dim, channels=channels, out_dim=out_dim, sigmoid_last_channel=sigmoid)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    x = seeded((B, channels, P, P), 21)
    t = torch.tensor(tvals, dtype=torch.long)
    x_bxyc = x.permute(0, 2, 3, 1).reshape(B, P * P, channels)
    out = m(x_bxyc, t)
    w = seeded(tuple(out.shape), 22)
    loss = (out * w).sum()
    loss.backward()
    d = {"x": npy(x), "t": npy(t), "w": npy(w)}
    if full_out:
        d["out"] = npy(out)
    else:
        d["out_probe"] = npy(out[:, :, ::8, ::8])
        d["out_sum"] = np.array(out.double().sum().item())
        d["out_abs_sum"] = np.array(out.double().abs().sum().item())
    names, norms = [], []
    for k, p in m.named_parameters():
        if p.grad is not None:
            names.append(k)
            norms.append(p.grad.double().norm().item())
    d["grad_names"] = np.array(names)
    d["grad_norms"] = np.array(norms)
    for k in PROBES:
        p = dict(m.named_parameters())[k]
        if p.grad is not None:
            d["grad/" + k] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **d)
    print(tag, "out", tuple(out.shape), "n_grads", len(names))


# G7: full model_estimation_loss with injected RNG --------------------------------------------------
def g7(tag, dim, P, B, tvals, n_steps=100):
    torch.manual_seed(0)
    m = Unet3D(dim=dim, channels=2)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    diff = DenoisingDiffusion(n_steps, "cpu")
    res = ResidualsDarcy(model=m, fd_acc=2, pixels_per_dim=P, pixels_at_boundary=True, reverse_d1=True,
                         device="cpu", bcs="none", domain_length=1.0)
    x0 = seeded((B, 2, P, P), 31)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = seeded((B, 2, P, P), 32)
    t = torch.tensor(tvals, dtype=torch.long)
    orig_randint, orig_randn_like = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t.clone()
    torch.randn_like = lambda *a, **k: eps.clone()
    try:
        loss, data_l, res_l, ineq_l, opt_l = diff.model_estimation_loss(
            x0, residual_func=res, c_data=1.0, c_residual=1e-3, c_ineq=0.0, lambda_opt=0.0)
    finally:
        torch.randint, torch.randn_like = orig_randint, orig_randn_like
    loss.backward()
    names, norms = [], []
    for k, p in m.named_parameters():
        if p.grad is not None:
            names.append(k)
            norms.append(p.grad.double().norm().item())
    d = dict(x0=npy(x0), eps=npy(eps), t=npy(t), loss=np.array(loss.item()), data_loss=np.array(data_l),
             residual_abs_mean=np.array(res_l), grad_names=np.array(names), grad_norms=np.array(norms))
    for k in PROBES[:6]:
        p = dict(m.named_parameters())[k]
        d["grad/" + k] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **d)
    print(tag, "loss", loss.item(), data_l, res_l)


# G8: sampler ------------------------------------------------------------------------------------------
def g8(tag, dim, P, B, n_steps=5):
    torch.manual_seed(0)
    m = Unet3D(dim=dim, channels=2)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    diff = DenoisingDiffusion(n_steps, "cpu")
    res = ResidualsDarcy(model=m, fd_acc=2, pixels_per_dim=P, pixels_at_boundary=True, reverse_d1=True,
                         device="cpu", bcs="none", domain_length=1.0)
    noises = [seeded((B, 2, P, P), 40 + i) for i in range(n_steps + 1)]
    it = iter(noises)
    orig_randn, orig_randn_like = torch.randn, torch.randn_like
    torch.randn = lambda *a, **k: next(it).clone()
    torch.randn_like = lambda *a, **k: next(it).clone()
    try:
        (x_seq, interm), aux = diff.p_sample_loop(None, (B, 2, P, P), save_output=True, surpress_noise=True,
                                                  residual_func=res, eval_residuals=True)
    finally:
        torch.randn, torch.randn_like = orig_randn, orig_randn_like
    d = dict(noises=np.stack([npy(n) for n in noises]), x_seq=np.stack([npy(x) for x in x_seq]),
             interm=np.stack([npy(x) for x in interm]), residual=npy(aux["residual"]))
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **d)
    print(tag, "x_final abs mean", float(np.abs(d["x_seq"][-1]).mean()))


# G9: mechanics residual / compliance / volume shift on a synthetic mesh (reference with dense K) ------------------
def write_synthetic_mesh(folder, nel=64):
    """SolidsPy text files for the nel x nel unit-square mesh of SURVEY 8(d): node id = row*(nel+1)+col,
    coords (x=col, y=nel-row), all dofs free, element nodes CCW [bl, br, tr, tl]."""
    nn = nel + 1
    os.makedirs(folder, exist_ok=True)
    nodes = np.zeros((nn * nn, 5))
    for r in range(nn):
        for c in range(nn):
            nodes[r * nn + c] = [r * nn + c, c, nel - r, 0, 0]
    eles = np.zeros((nel * nel, 7), dtype=int)
    for r in range(nel):
        for c in range(nel):
            eles[r * nel + c] = [r * nel + c, 1, 0, (r + 1) * nn + c, (r + 1) * nn + c + 1, r * nn + c + 1, r * nn + c]
    np.savetxt(folder + "nodes.txt", nodes, fmt="%d %.6f %.6f %d %d")
    np.savetxt(folder + "eles.txt", eles, fmt="%d")
    np.savetxt(folder + "mater.txt", np.array([[1.0, 0.3]]))
    np.savetxt(folder + "loads.txt", np.array([[32 * nn + 64, 0.0, -1.0]]))


def g9():
    import tempfile
    from src.residuals_mechanics_K import ResidualsMechanics
    folder = tempfile.mkdtemp() + "/"
    write_synthetic_mesh(folder)
    res = ResidualsMechanics(model=None, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder=folder, device="cpu",
                             topopt_eval=False)
    B = 2
    x0 = seeded((B, 3, 64, 64), 51, 0.1)
    x0[:, 2] = torch.sigmoid(seeded((B, 64, 64), 52))
    bcs = torch.zeros(B, 4, 65, 65)
    bcs[:, 0, :, 0] = 1.0      # clamp x-displacement on the left edge
    bcs[:, 1, :, 0] = 1.0      # clamp y-displacement on the left edge
    bcs[0, 3, 32, 64] = -1.0   # point load
    bcs[1, 2, 10, 64] = 0.5
    vf = torch.tensor([0.3, 0.45])
    x0.requires_grad_(True)
    out = res.compute_residual((x0, bcs, vf), reduce="none", return_model_out=True, return_optimizer=True,
                               return_inequality=True, pass_through=True)
    wr = seeded(tuple(out["residual"].shape), 53)
    wm = seeded(tuple(out["model_out"].shape), 54)
    scal = (out["residual"] * wr).sum() + (out["model_out"] * wm).sum() + 0.7 * out["optimizer"].sum() + 1.3 * (out["inequality"] ** 2).sum()
    (g,) = torch.autograd.grad(scal, x0)
    np.savez_compressed(os.path.join(OUT, "g9_mechanics.npz"), x0=npy(x0), bcs=npy(bcs), vf=npy(vf), residual=npy(out["residual"]),
                        model_out=npy(out["model_out"]), compliance=npy(out["optimizer"]), shift=npy(out["inequality"]),
                        wr=npy(wr), wm=npy(wm), grad_x0=npy(g), kloc0=npy(res.stiffs.tot_local_stiffness[0]),
                        elem_dofs=npy(res.stiffs.glob_assembler_idcs[:, :8, 1]).astype(np.int32))
    print("g9 mechanics: |r| mean", float(out["residual"].abs().mean()), "compliance", npy(out["optimizer"]))


# G10: full mechanics model_estimation_loss (tiny UNet, synthetic mesh, c_ineq > 0, lambda > 0) ----------------------
def g10():
    import tempfile
    from src.residuals_mechanics_K import ResidualsMechanics
    folder = tempfile.mkdtemp() + "/"
    write_synthetic_mesh(folder)
    torch.manual_seed(0)
    m = Unet3D(dim=8, channels=10, out_dim=3, sigmoid_last_channel=True)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    diff = DenoisingDiffusion(100, "cpu")
    res = ResidualsMechanics(model=m, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder=folder, device="cpu",
                             topopt_eval=False)
    B = 2
    inp = torch.zeros(B, 10, 65, 65)
    inp[:, 0] = torch.tensor([0.3, 0.45]).view(B, 1, 1)
    inp[:, 1:3] = seeded((B, 2, 65, 65), 61)
    inp[:, 3:5] = seeded((B, 2, 65, 65), 62, 0.1)
    inp[:, 5, :64, :64] = torch.sigmoid(seeded((B, 64, 64), 63))
    inp[:, 6, :, 0] = 1.0
    inp[:, 7, :, 0] = 1.0
    inp[0, 9, 32, 64] = -1.0
    inp[1, 8, 10, 64] = 0.5
    eps = seeded((B, 3, 65, 65), 64)
    t = torch.tensor([4, 71], dtype=torch.long)
    orig_randint, orig_randn_like = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t.clone()
    torch.randn_like = lambda *a, **k: eps.clone()
    try:
        loss, data_l, res_l, ineq_l, opt_l = diff.model_estimation_loss(inp, residual_func=res, c_data=1.0, c_residual=1e-3,
                                                                        c_ineq=0.5, lambda_opt=0.01)
    finally:
        torch.randint, torch.randn_like = orig_randint, orig_randn_like
    loss.backward()
    names, norms = [], []
    for k, p in m.named_parameters():
        if p.grad is not None:
            names.append(k)
            norms.append(p.grad.double().norm().item())
    np.savez_compressed(os.path.join(OUT, "g10_mech_loss_dim8.npz"), inp=npy(inp), eps=npy(eps), t=npy(t), loss=np.array(loss.item()),
                        data_loss=np.array(data_l), residual_abs_mean=np.array(res_l), ineq=np.array(ineq_l), opt=np.array(opt_l),
                        grad_names=np.array(names), grad_norms=np.array(norms))
    print("g10 mech loss", loss.item(), data_l, res_l, ineq_l, opt_l, len(names))


# G11: topology-optimisation evaluation block (reference :276-347 with topopt_eval=True, sample=True): FE solve of the
# binarised prediction (fp32 torch.linalg.solve in the reference), data compliance, volume-fraction error, floating material.
# The data "solution" is manufactured: displacements that solve K_closed(rho_simp) u = f (float64 solve, stored as fp32).
def g11():
    import tempfile
    from src.residuals_mechanics_K import ResidualsMechanics
    folder = tempfile.mkdtemp() + "/"
    write_synthetic_mesh(folder)
    res = ResidualsMechanics(model=None, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder=folder, device="cpu",
                             topopt_eval=True)
    B = 2
    yy, xx = np.meshgrid(np.arange(64), np.arange(64), indexing="ij")
    # predicted densities: sample 0 = one solid cantilever band reaching the clamped edge and the load; sample 1 = the same
    # band plus a detached island (floating material)
    band = (np.abs(yy - 32) < 12).astype(np.float64)
    rho_pred = np.stack([0.1 + 0.8 * band, 0.1 + 0.8 * band])
    rho_pred[1][(np.abs(yy - 6) < 3) & (np.abs(xx - 40) < 5)] = 0.9
    rho_pred += 0.05 * npy(seeded((B, 64, 64), 71))
    x0 = torch.zeros(B, 3, 64, 64)
    x0[:, :2] = seeded((B, 2, 64, 64), 72, 0.1)
    x0[:, 2] = torch.from_numpy(rho_pred).float()
    bcs = torch.zeros(B, 4, 65, 65)
    bcs[:, 0, :, 0] = 1.0
    bcs[:, 1, :, 0] = 1.0
    bcs[0, 3, 32, 64] = -0.01
    bcs[1, 3, 30, 64] = -0.02
    vf = torch.tensor([0.4, 0.35])
    kloc = npy(res.stiffs.tot_local_stiffness[0]).astype(np.float64)
    elem_dofs = npy(res.stiffs.glob_assembler_idcs[:, :8, 1]).astype(np.int64)
    # data solution: SIMP-like density (smooth, in [0.05, 1]) and its exact displacement field
    rho_simp = np.clip(0.05 + 0.95 * np.stack([band, band]) * (0.7 + 0.3 * npy(torch.sigmoid(seeded((B, 64, 64), 73)))), 0.05, 1.0)
    solution = torch.zeros(B, 3, 65, 65)
    for b in range(B):
        u, _ = O.mechanics_fe_solve(rho_simp[b].reshape(-1), npy(bcs[b]), kloc, elem_dofs)
        solution[b, :2] = torch.from_numpy(u.reshape(65, 65, 2).transpose(2, 0, 1)).float()
        solution[b, 2, :64, :64] = torch.from_numpy(rho_simp[b]).float()
    out = res.compute_residual((x0, bcs, vf, solution), reduce="none", return_model_out=False, return_optimizer=True,
                               return_inequality=True, sample=True, pass_through=True)
    truth = O.mechanics_topopt_metrics(rho_pred=npy(x0[:, 2]), bcs=npy(bcs), vf=npy(vf), solution=npy(solution), kloc=kloc,
                                       elem_dofs=elem_dofs)
    np.savez_compressed(os.path.join(OUT, "g11_topopt_eval.npz"), x0=npy(x0), bcs=npy(bcs), vf=npy(vf), solution=npy(solution),
                        rel_CE_error=npy(out["rel_CE_error_full_batch"]), vf_error=npy(out["vf_error_full_batch"]),
                        fm=npy(out["fm_error_full_batch"]).astype(np.int64),
                        rel_CE_error_f64=truth["rel_CE_error"], compliance_true_f64=truth["compliance_true"],
                        compliance_data_f64=truth["compliance_data"])
    print("g11 topopt eval: reference rel_CE", npy(out["rel_CE_error_full_batch"]), "float64 restatement", truth["rel_CE_error"],
          "vf_err", npy(out["vf_error_full_batch"]), truth["vf_error"], "fm", npy(out["fm_error_full_batch"]), truth["fm"])


# G12: CoCoGen residual correction (reference ResidualsDarcy.residual_correction incl. its vmap(jacfwd) Jacobian maximum)
def g12():
    P, B = 16, 2
    res = ResidualsDarcy(model=None, fd_acc=2, pixels_per_dim=P, pixels_at_boundary=True, reverse_d1=True, device="cpu",
                         bcs="none", domain_length=1.)
    x = seeded((B, P * P, 2), 81)
    x[:, :, 1] = torch.exp(0.5 * x[:, :, 1])
    x_in = x.clone()
    from torch.func import jacfwd, vmap
    jac = vmap(jacfwd(res.compute_residual_direct, argnums=0, has_aux=False), in_dims=0, out_dims=0)(x_in.clone()).squeeze(1)[:, :, :, :, 0]
    mx = jac.reshape(B, -1).max(dim=1)[0]
    x_out, r_out = res.residual_correction(x)          # corrects x in place
    np.savez_compressed(os.path.join(OUT, "g12_cocogen_p16.npz"), x_in=npy(x_in), x_out=npy(x_out), delta_p=npy(x_out[:, :, 0] - x_in[:, :, 0]),
                        residual_corrected=npy(r_out), max_dr_dp=npy(mx))
    print("g12 cocogen: max_dr_dp", npy(mx), "max |delta p|", float((x_out - x_in).abs().max()))


# G13: gradient-guidance baseline (residual_grad_guidance=True): training loss with a fixed classifier-free mask, and one
# guided sampler step (forward_with_guidance_scale, scale 3)
def g13():
    import src.unet_model as um
    dim, P, B = 8, 16, 3
    torch.manual_seed(0)
    m = Unet3D(dim=dim, channels=2)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    diff = DenoisingDiffusion(100, "cpu", residual_grad_guidance=True)
    res = ResidualsDarcy(model=m, fd_acc=2, pixels_per_dim=P, pixels_at_boundary=True, reverse_d1=True, device="cpu", bcs="none",
                         domain_length=1., residual_grad_guidance=True)
    x0 = seeded((B, 2, P, P), 91)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = seeded((B, 2, P, P), 92)
    t = torch.tensor([2, 40, 97], dtype=torch.long)
    mask = torch.tensor([False, True, False])
    orig = torch.randint, torch.randn_like, um.prob_mask_like
    torch.randint = lambda *a, **k: t.clone()
    torch.randn_like = lambda *a, **k: eps.clone()
    um.prob_mask_like = lambda shape, prob, device: mask.clone() if 0 < prob < 1 else orig[2](shape, prob, device)
    try:
        loss, data_l, res_l, _, _ = diff.model_estimation_loss(x0, residual_func=res, c_data=1.0, c_residual=1e-3, c_ineq=0., lambda_opt=0.)
        loss.backward()
        names, norms = [], []
        for k, p_ in m.named_parameters():
            if p_.grad is not None:
                names.append(k)
                norms.append(p_.grad.double().norm().item())
        # guided sampler step at t = 5 (sample=True -> forward_with_guidance_scale)
        xs = seeded((B, 2, P, P), 93)
        z = seeded((B, 2, P, P), 94)
        torch.randn_like = lambda *a, **k: z.clone()
        (x_next, x0_pred), _ = diff.p_sample(xs, None, 5, save_output=True, surpress_noise=True, residual_func=res)
    finally:
        torch.randint, torch.randn_like, um.prob_mask_like = orig
    np.savez_compressed(os.path.join(OUT, "g13_guidance_dim8_p16.npz"), x0=npy(x0), eps=npy(eps), t=npy(t), mask=npy(mask),
                        loss=np.array(loss.item()), data_loss=np.array(data_l), residual_abs_mean=np.array(res_l),
                        grad_names=np.array(names), grad_norms=np.array(norms), xs=npy(xs), z=npy(z), x_next=npy(x_next),
                        x0_pred_guided=npy(x0_pred))
    print("g13 guidance: loss", loss.item(), data_l, res_l, "grads", len(names))


# G14: ddim_sample_x0 with ddim_steps in {0, 2}: outputs and RNG consumption (the value of the next torch.rand draw)
def g14():
    dim, P, B = 8, 16, 2
    m = Unet3D(dim=dim, channels=2)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    diff = DenoisingDiffusion(100, "cpu")
    xt = seeded((B, P * P, 2), 95)
    t = torch.tensor([37, 80], dtype=torch.long)
    out = {}
    for k in (0, 2):
        torch.manual_seed(4321)
        with torch.no_grad():
            x0_pred, model_out = diff.ddim_sample_x0(xt, t, m, (B, 2, P, P), k, 0.)
        out[f"x0_pred_k{k}"] = npy(x0_pred)
        out[f"model_out_k{k}"] = npy(model_out)
        out[f"next_rand_k{k}"] = npy(torch.rand(4))
    np.savez_compressed(os.path.join(OUT, "g14_ddim_x0.npz"), xt=npy(xt), t=npy(t), **out)
    print("g14 ddim: |x0_pred_k0 - x0_pred_k2| max", float(np.abs(out["x0_pred_k0"] - out["x0_pred_k2"]).max()), out["next_rand_k0"], out["next_rand_k2"])


# G15: self-conditioning (Unet3D(self_condition=True)): forward with and without x_self_cond, gradient norms
def g15():
    dim, P, B = 8, 16, 2
    m = Unet3D(dim=dim, channels=2, self_condition=True)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    x = seeded((B, 2, P, P), 96)
    sc = seeded((B, 2, P, P), 97)
    t = torch.tensor([9, 66], dtype=torch.long)
    out_sc = m(x, t, x_self_cond=sc.unsqueeze(2))   # the reference concatenates after lifting x to [B,C,1,P,P] (:556-566)
    out_none = m(x, t)
    w = seeded(tuple(out_sc.shape), 98)
    (out_sc * w).sum().backward()
    names, norms = [], []
    for k, p_ in m.named_parameters():
        if p_.grad is not None:
            names.append(k)
            norms.append(p_.grad.double().norm().item())
    np.savez_compressed(os.path.join(OUT, "g15_selfcond_dim8_p16.npz"), x=npy(x), sc=npy(sc), t=npy(t), out_sc=npy(out_sc),
                        out_none=npy(out_none), w=npy(w), grad_names=np.array(names), grad_norms=np.array(norms),
                        init_conv_shape=np.array(m.init_conv.weight.shape))
    print("g15 selfcond:", tuple(m.init_conv.weight.shape), float(out_sc.abs().max()), len(names))


# G16: mechanics sampler steps (p_sample with conditioning_input): t = 3 (no evaluation) and t = 0 with topopt_eval=True
def g16():
    import tempfile
    from src.residuals_mechanics_K import ResidualsMechanics
    folder = tempfile.mkdtemp() + "/"
    write_synthetic_mesh(folder)
    m = Unet3D(dim=8, channels=10, out_dim=3, sigmoid_last_channel=True)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    diff = DenoisingDiffusion(100, "cpu")
    res = ResidualsMechanics(model=m, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder=folder, device="cpu", topopt_eval=True)
    B = 2
    x = seeded((B, 3, 65, 65), 101)
    cond = torch.zeros(B, 3, 65, 65)
    cond[:, 0] = torch.tensor([0.3, 0.45]).view(B, 1, 1)
    cond[:, 1:3] = seeded((B, 2, 65, 65), 102)
    bcs = torch.zeros(B, 4, 65, 65)
    bcs[:, 0, :, 0] = 1.0
    bcs[:, 1, :, 0] = 1.0
    bcs[0, 3, 32, 64] = -0.01
    bcs[1, 3, 30, 64] = -0.02
    yy = np.arange(64).reshape(64, 1) * np.ones((1, 64))
    band = (np.abs(yy - 32) < 12).astype(np.float64)
    rho_simp = np.clip(0.05 + 0.95 * np.stack([band, band]) * (0.7 + 0.3 * npy(torch.sigmoid(seeded((B, 64, 64), 103)))), 0.05, 1.0)
    kloc = npy(res.stiffs.tot_local_stiffness[0]).astype(np.float64)
    elem_dofs = npy(res.stiffs.glob_assembler_idcs[:, :8, 1]).astype(np.int64)
    solution = torch.zeros(B, 3, 65, 65)
    for b in range(B):
        u, _ = O.mechanics_fe_solve(rho_simp[b].reshape(-1), npy(bcs[b]), kloc, elem_dofs)
        solution[b, :2] = torch.from_numpy(u.reshape(65, 65, 2).transpose(2, 0, 1)).float()
        solution[b, 2, :64, :64] = torch.from_numpy(rho_simp[b]).float()
    z = seeded((B, 3, 65, 65), 104)
    orig = torch.randn_like
    torch.randn_like = lambda *a, **k: z.clone()
    try:
        (x3, mo3), aux3 = diff.p_sample(x, (cond, bcs, solution), 3, save_output=True, surpress_noise=True, residual_func=res,
                                        eval_residuals=True, return_optimizer=True, return_inequality=True)
        (x0, mo0), aux0 = diff.p_sample(x, (cond, bcs, solution), 0, save_output=True, surpress_noise=True, residual_func=res,
                                        eval_residuals=True, return_optimizer=True, return_inequality=True)
    finally:
        torch.randn_like = orig
    assert aux3 is None
    np.savez_compressed(os.path.join(OUT, "g16_mech_sampler_dim8.npz"), x=npy(x), cond=npy(cond), bcs=npy(bcs), solution=npy(solution),
                        z=npy(z), x3=npy(x3), mo3=npy(mo3), x0=npy(x0), mo0=npy(mo0), residual0=npy(aux0["residual"]),
                        compliance0=npy(aux0["optimized_quant"]), shift0=npy(aux0["inequality_quant"]),
                        rel_CE=npy(aux0["rel_CE_error_full_batch"]), vf_err=npy(aux0["vf_error_full_batch"]),
                        fm=npy(aux0["fm_error_full_batch"]).astype(np.int64))
    print("g16 mech sampler: rel_CE", npy(aux0["rel_CE_error_full_batch"]), "vf", npy(aux0["vf_error_full_batch"]), "fm", npy(aux0["fm_error_full_batch"]))


# G17: dataset readers (src/data_utils.py:31-119): CSV-per-channel Dataset and .npy-per-sample Dataset_Paths
def g17():
    import tempfile
    from src.data_utils import Dataset, Dataset_Paths
    d = tempfile.mkdtemp()
    rng = np.random.default_rng(7)
    p_csv, k_csv = rng.normal(size=(3, 16)), rng.normal(size=(3, 16))
    np.savetxt(d + "/p.csv", p_csv, delimiter=",")
    np.savetxt(d + "/K.csv", k_csv, delimiter=",")
    ds = Dataset((d + "/p.csv", d + "/K.csv"), use_double=False)
    os.makedirs(d + "/fields")
    arrs = [rng.normal(size=(5, 5, 10)) for _ in range(3)]
    for i, name in enumerate(("10", "2", "33")):          # numeric (not lexicographic) file order
        np.save(d + f"/fields/{name}.npy", arrs[i])
    dp = Dataset_Paths(d + "/fields/", use_double=False)
    np.savez_compressed(os.path.join(OUT, "g17_datasets.npz"), p_csv=p_csv, k_csv=k_csv, ds_data=npy(torch.stack([ds[i] for i in range(len(ds))])),
                        npy_arrays=np.stack(arrs), npy_names=np.array(["10", "2", "33"]),
                        dp_items=npy(torch.stack([dp[i] for i in range(len(dp))])))
    print("g17 datasets:", tuple(ds[0].shape), len(ds), tuple(dp[0].shape), len(dp))


# G18: the SpatialLinearAttention module alone (src/unet_model.py:269-299), forward and all gradients ---------------------
def g18():
    from src.unet_model import SpatialLinearAttention
    heads, dim, B, H = 2, 32, 2, 16
    g = torch.Generator().manual_seed(1801)
    m = SpatialLinearAttention(dim, heads=heads, dim_head=32)
    with torch.no_grad():
        m.to_qkv.weight.copy_(torch.randn(m.to_qkv.weight.shape, generator=g) * 0.25)
        m.to_out.weight.copy_(torch.randn(m.to_out.weight.shape, generator=g) * 0.1)
        m.to_out.bias.copy_(torch.randn(m.to_out.bias.shape, generator=g) * 0.1)
    x = (torch.randn(B, dim, 1, H, H, generator=g) * 1.5).requires_grad_(True)
    gy = torch.randn(B, dim, 1, H, H, generator=g)
    qkv_keep = {}
    def keep(mod, inp, out):
        out.retain_grad()
        qkv_keep["qkv"] = out

    hk = m.to_qkv.register_forward_hook(keep)
    y = m(x)
    y.backward(gy)
    hk.remove()
    qkv = qkv_keep["qkv"]
    np.savez_compressed(os.path.join(OUT, "g18_linear_attention.npz"), heads=heads, x=npy(x[:, :, 0]), gy=npy(gy[:, :, 0]),
                        w_qkv=npy(m.to_qkv.weight[:, :, 0, 0]), w_out=npy(m.to_out.weight[:, :, 0, 0]), b_out=npy(m.to_out.bias),
                        y=npy(y[:, :, 0]), d_qkv=npy(qkv.grad), d_x=npy(x.grad[:, :, 0]),
                        d_w_qkv=npy(m.to_qkv.weight.grad[:, :, 0, 0]), d_w_out=npy(m.to_out.weight.grad[:, :, 0, 0]),
                        d_b_out=npy(m.to_out.bias.grad))
    print("g18 linear attention:", tuple(y.shape), float(y.abs().max()), float(qkv.grad.abs().max()))


# G19: the mechanics-shaped UNet at its FULL width (main.py:126 builds Unet3D(dim=128, channels=10, out_dim=3,
# sigmoid_last_channel=True)): channel widths 128..1024 reach conv tile variants no dim<=32 golden exercises.
# Large gradient tensors are stored as strided samples of the flattened tensor (stride in "gstride/<name>").
BIG_PROBES = ["init_conv.weight", "downs.0.0.block1.proj.weight", "downs.1.0.res_conv.weight", "downs.0.2.fn.fn.to_qkv.weight",
              "downs.0.2.fn.fn.to_out.weight", "downs.0.2.fn.norm.gamma", "downs.0.3.weight", "ups.0.3.weight",
              "time_mlp.3.weight", "downs.0.0.mlp.1.weight", "mid_spatial_attn.fn.fn.fn.to_qkv.weight",
              "mid_spatial_attn.fn.norm.gamma", "downs.3.1.block2.proj.weight", "downs.3.0.block1.norm.weight",
              "final_conv.1.weight", "final_conv.0.res_conv.weight", "ups.3.0.block1.proj.weight", "ups.0.0.block1.proj.weight",
              "mid_block1.block2.proj.weight", "downs.2.2.fn.fn.to_out.bias"]


def g19(tag="g19_unet_dim128_mech", dim=128, P=64, tval=42):
    torch.manual_seed(0)
    m = Unet3D(dim=dim, channels=10, out_dim=3, sigmoid_last_channel=True)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    x = seeded((1, 10, P, P), 71)
    t = torch.tensor([tval], dtype=torch.long)
    out = m(x.permute(0, 2, 3, 1).reshape(1, P * P, 10), t)
    w = seeded(tuple(out.shape), 72)
    (out * w).sum().backward()
    d = {"x": npy(x), "t": npy(t), "w": npy(w), "out_probe": npy(out[:, :, ::4, ::4]),
         "out_sum": np.array(out.double().sum().item()), "out_abs_sum": np.array(out.double().abs().sum().item())}
    names, norms = [], []
    params = dict(m.named_parameters())
    for k, p in params.items():
        if p.grad is not None:
            names.append(k)
            norms.append(p.grad.double().norm().item())
    d["grad_names"] = np.array(names)
    d["grad_norms"] = np.array(norms)
    for k in BIG_PROBES:
        gflat = params[k].grad.reshape(-1)
        stride = max(1, gflat.numel() // 8192)
        if stride > 1 and stride % 2 == 0:
            stride += 1          # odd stride: the sample walks through every tap / channel residue
        d["grad/" + k] = npy(gflat[::stride])
        d["gstride/" + k] = np.array(stride)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **d)
    print(tag, "out", tuple(out.shape), "n_grads", len(names), "out_abs_sum", float(d["out_abs_sum"]))


# G10b: the full mechanics model_estimation_loss at the reference's model width (dim=128), one sample
def g10b():
    import tempfile
    from src.residuals_mechanics_K import ResidualsMechanics
    folder = tempfile.mkdtemp() + "/"
    write_synthetic_mesh(folder)
    torch.manual_seed(0)
    m = Unet3D(dim=128, channels=10, out_dim=3, sigmoid_last_channel=True)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    diff = DenoisingDiffusion(100, "cpu")
    res = ResidualsMechanics(model=m, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder=folder, device="cpu",
                             topopt_eval=False)
    B = 1
    inp = torch.zeros(B, 10, 65, 65)
    inp[:, 0] = 0.4
    inp[:, 1:3] = seeded((B, 2, 65, 65), 81)
    inp[:, 3:5] = seeded((B, 2, 65, 65), 82, 0.1)
    inp[:, 5, :64, :64] = torch.sigmoid(seeded((B, 64, 64), 83))
    inp[:, 6, :, 0] = 1.0
    inp[:, 7, :, 0] = 1.0
    inp[0, 9, 20, 64] = -1.0
    eps = seeded((B, 3, 65, 65), 84)
    t = torch.tensor([33], dtype=torch.long)
    orig_randint, orig_randn_like = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t.clone()
    torch.randn_like = lambda *a, **k: eps.clone()
    try:
        loss, data_l, res_l, ineq_l, opt_l = diff.model_estimation_loss(inp, residual_func=res, c_data=1.0, c_residual=1e-3,
                                                                        c_ineq=0.5, lambda_opt=0.01)
    finally:
        torch.randint, torch.randn_like = orig_randint, orig_randn_like
    loss.backward()
    names, norms = [], []
    for k, p in m.named_parameters():
        if p.grad is not None:
            names.append(k)
            norms.append(p.grad.double().norm().item())
    np.savez_compressed(os.path.join(OUT, "g10b_mech_loss_dim128.npz"), inp=npy(inp), eps=npy(eps), t=npy(t), loss=np.array(loss.item()),
                        data_loss=np.array(data_l), residual_abs_mean=np.array(res_l), ineq=np.array(ineq_l), opt=np.array(opt_l),
                        grad_names=np.array(names), grad_norms=np.array(norms))
    print("g10b mech loss dim128", loss.item(), data_l, res_l, ineq_l, opt_l, len(names))


# G20: state_dict contract of the reference's Unet3D - every key in order, its shape, and float64 checksums of the DEFAULT
# initialisation under torch.manual_seed(0) (the drop-in must build the identical module tree in the identical RNG draw order)
def g20():
    d = {}
    for tag, kw in (("darcy", dict(dim=32, channels=2)),
                    ("mech", dict(dim=32, channels=10, out_dim=3, sigmoid_last_channel=True)),
                    ("selfcond", dict(dim=8, channels=2, self_condition=True))):
        torch.manual_seed(0)
        m = Unet3D(**kw)
        sd = m.state_dict()
        d[tag + "/names"] = np.array(list(sd.keys()))
        d[tag + "/shapes"] = np.array([",".join(str(s) for s in v.shape) for v in sd.values()])
        d[tag + "/sum"] = np.array([v.double().sum().item() for v in sd.values()])
        d[tag + "/abs_sum"] = np.array([v.double().abs().sum().item() for v in sd.values()])
        d[tag + "/requires_grad"] = np.array([k for k, p in m.named_parameters() if p.requires_grad])
        d[tag + "/next_rand"] = npy(torch.rand(4))       # RNG position after construction
    np.savez_compressed(os.path.join(OUT, "g20_state_dict.npz"), **d)
    print("g20 state_dict", {k: len(v) for k, v in d.items() if k.endswith("/names")})


def g21():
    """Thin tensor-algebra members of DenoisingDiffusion that main.py / sample.py never call (src/denoising_utils.py:547-614,57-68):
    normal_kl, predict_start_from_noise, predict_noise_from_start, predict_noise_from_mean, loss_variational (both log bases, a
    batch that contains t == 0), module-level resize_image."""
    dd = DenoisingDiffusion(100, "cpu")
    B, C, P = 4, 2, 8
    x0, xt, out, noise = seeded((B, C, P, P), 211), seeded((B, C, P, P), 212), seeded((B, C, P, P), 213), seeded((B, C, P, P), 214)
    t = torch.tensor([0, 1, 50, 99])
    img = seeded((2, 3, 2, 8, 8), 215)
    np.savez_compressed(
        os.path.join(OUT, "g21_diffusion_algebra.npz"), x0=npy(x0), xt=npy(xt), out=npy(out), noise=npy(noise), t=npy(t), img=npy(img),
        normal_kl=npy(dd.normal_kl(x0, 0.3 * xt, out, 0.2 * noise)),
        start_from_noise=npy(dd.predict_start_from_noise(xt, t, noise)),
        noise_from_start=npy(dd.predict_noise_from_start(xt, t, x0)),
        noise_from_mean=npy(dd.predict_noise_from_mean(xt, t, out)),
        loss_variational=npy(dd.loss_variational(out, x0, xt, t)),
        loss_variational_base2=npy(dd.loss_variational(out, x0, xt, t, base_2=True)),
        resized5=npy(du.resize_image(img, 5)), resized13=npy(du.resize_image(img, 13)))
    print("g21 done")


def g23(tag="g23_unet_input_forms", dim=8, P=16, B=2):
    """Unet3D.forward's three input forms (src/unet_model.py:554-562,616-618): [B, P*P, C], [B, C, P, P] and [B, C, 1, P, P] - outputs
    (the 5-D form keeps its frame axis) and the gradient of (out * w).sum() with respect to the input for each."""
    torch.manual_seed(0)
    m = Unet3D(dim=dim, channels=2)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    x = seeded((B, 2, P, P), 231)
    t = torch.tensor([4, 93], dtype=torch.long)
    w = seeded((B, 2, P, P), 232)
    d = {"x": npy(x), "t": npy(t), "w": npy(w)}
    forms = {"bxyc": x.permute(0, 2, 3, 1).reshape(B, P * P, 2), "bcpp": x.clone(), "bc1pp": x.unsqueeze(2).clone()}
    for name, xin in forms.items():
        xin = xin.clone().requires_grad_(True)
        out = m(xin, t)
        ww = w.unsqueeze(2) if out.dim() == 5 else w
        (out * ww).sum().backward()
        d["out_" + name] = npy(out)
        d["gx_" + name] = npy(xin.grad)
        print(tag, name, "in", tuple(xin.shape), "out", tuple(out.shape))
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **d)


def toy_funcs():
    """The three callables main_toy.py:49-82 defines inline (unit-circle residual, L1-density inequality, x-coordinate objective)."""
    residual = lambda x: torch.sum(x ** 2, dim=1) - 1.0                                        # noqa: E731
    def ineq(x):
        density = torch.sum(torch.abs(x), dim=1)
        return torch.relu(density - 1.0), density
    opt = lambda x: x[:, 0]                                                                    # noqa: E731
    return residual, ineq, opt


def g24(tag="g24_toy_config"):
    """BASELINE configs[0] (main_toy.py:113-130 -> src/denoising_toy_utils.py): default initialisation of ConditionalModel under a
    seed, model_estimation_loss in its three parameterisations with both x0 estimates (injected RNG), a 6-step p_sample_loop with
    save_output, and the schedule dictionary."""
    import src.denoising_toy_utils as toy
    assert toy.__file__.startswith("/root/reference/"), toy.__file__
    d = {}
    torch.manual_seed(5)
    m0 = toy.ConditionalModel(2, 100)
    sd0 = m0.state_dict()
    d["init/names"] = np.array(list(sd0.keys()))
    d["init/sum"] = np.array([v.double().sum().item() for v in sd0.values()])
    d["init/abs_sum"] = np.array([v.double().abs().sum().item() for v in sd0.values()])
    d["init/next_rand"] = npy(torch.rand(3))
    dd = toy.create_diff_dict(100, "cpu")
    d["sched/names"] = np.array(sorted(dd.keys()))
    for k in dd:
        d["sched/" + k] = npy(dd[k])
    residual, ineq, opt = toy_funcs()
    n_steps, B = 100, 9
    m = toy.ConditionalModel(2, n_steps)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    x0 = seeded((B, 2), 241)
    x0 = x0 / x0.norm(dim=1, keepdim=True)
    t_half = torch.tensor([3, 0, 57, 99, 41])             # B // 2 + 1 draws (antithetic completion inside the loss)
    eps = seeded((B, 2), 242)
    d["x0"], d["t_half"], d["eps"] = npy(x0), npy(t_half), npy(eps)
    for mode in ("x0", "eps", "mu"):
        for ddim in (False, True):
            extra = [seeded((B, 2), 243 + i) for i in range(4)]
            it = iter([eps] + extra)
            orig = torch.randint, torch.randn_like
            torch.randint = lambda *a, **k: t_half.clone()
            torch.randn_like = lambda *a, **k: next(it).clone()
            for p_ in m.parameters():
                p_.grad = None
            try:
                out = toy.model_estimation_loss(m, x0, n_steps, dd, model_pred_mode=mode, residual_func=residual, ineq_func=ineq,
                                                opt_func=opt, c_data=1.0, c_residual=0.005, c_ineq=0.1, lambda_opt=0.01,
                                                use_ddim_x0=ddim, reduced_ddim_steps=1 if ddim else 0)
            finally:
                torch.randint, torch.randn_like = orig
            out[0].backward()
            key = f"loss/{mode}/{'sample' if ddim else 'mean'}"
            d[key] = np.array([out[0].item(), out[1], out[2], out[3], out[4]])
            d[key + "/grad_norms"] = np.array([p_.grad.double().norm().item() for p_ in m.parameters()])
            print(tag, key, d[key])
    # sampler, x0 parameterisation, 6-step schedule
    dd6 = toy.create_diff_dict(6, "cpu")
    m6 = toy.ConditionalModel(2, 6)
    m6.load_state_dict(fill_state_dict(m6.state_dict()))
    noises = [seeded((7, 2), 250 + i) for i in range(16)]      # more than the loop draws; the count it consumed is stored
    used = {"n": 0}

    def draw(*a, **k):
        used["n"] += 1
        return noises[used["n"] - 1].clone()
    orig = torch.randn, torch.randn_like
    torch.randn = draw
    torch.randn_like = draw
    try:
        with torch.no_grad():
            x_seq, outs, x0s = toy.p_sample_loop(m6, [7, 2], 6, dd6, model_pred_mode="x0", save_output=True, surpress_noise=True,
                                                 reduced_ddim_steps=0)
    finally:
        torch.randn, torch.randn_like = orig
    d["sampler/noises"] = np.stack([npy(n) for n in noises])
    d["sampler/draws"] = np.array(used["n"])
    d["sampler/x_seq"] = np.stack([npy(x) for x in x_seq])
    d["sampler/model_outputs"] = np.stack([npy(x) for x in outs])
    d["sampler/x0_estimations"] = np.stack([npy(x) for x in x0s])
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **d)
    print(tag, "written")


G22_FRAMES = [0, 1, 2, 10, 100, 500, 900, 990, 999, 1000]


def g22(tag="g22_sampler_1000steps_dim8_p16", dim=8, P=16, B=2, n_steps=1000):
    """The full 1000-step DDPM chain of sample.py:145-150 (DenoisingDiffusion(1000).p_sample_loop, src/denoising_utils.py:494-545) at a
    size the reference runs in seconds, with injected noise.  The 1001 noise fields are NOT stored: draw k is
    seeded((B, 2, P, P), 22000 + k) (the fixture holds the seed base); of the 1001 frames only G22_FRAMES are kept."""
    torch.manual_seed(0)
    m = Unet3D(dim=dim, channels=2)
    m.load_state_dict(fill_state_dict(m.state_dict()))
    diff = DenoisingDiffusion(n_steps, "cpu")
    res = ResidualsDarcy(model=m, fd_acc=2, pixels_per_dim=P, pixels_at_boundary=True, reverse_d1=True,
                         device="cpu", bcs="none", domain_length=1.0)
    seed_base = 22000
    k = {"n": 0}

    orig_randn, orig_randn_like = torch.randn, torch.randn_like

    def draw(*a, **kw):
        z = orig_randn(B, 2, P, P, generator=torch.Generator().manual_seed(seed_base + k["n"]))
        k["n"] += 1
        return z
    torch.randn = draw
    torch.randn_like = draw
    try:
        (x_seq, interm), aux = diff.p_sample_loop(None, (B, 2, P, P), save_output=True, surpress_noise=True,
                                                  residual_func=res, eval_residuals=True)
    finally:
        torch.randn, torch.randn_like = orig_randn, orig_randn_like
    assert len(x_seq) == n_steps + 1 and k["n"] == n_steps + 1, (len(x_seq), k["n"])
    d = dict(seed_base=np.array(seed_base), n_steps=np.array(n_steps), frames=np.array(G22_FRAMES),
             x_seq=np.stack([npy(x_seq[f]) for f in G22_FRAMES]), interm=np.stack([npy(interm[f]) for f in G22_FRAMES]),
             residual=npy(aux["residual"]), x_absmax=np.array([float(x.abs().max()) for x in x_seq]))
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **d)
    print(tag, "x_final abs mean", float(np.abs(d["x_seq"][-1]).mean()), "max |x| over the chain", float(d["x_absmax"].max()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g10b", "g20", "g22", "g23", "g24"):
        {"g24": g24, "g23": g23, "g22": g22, "g19": g19, "g10b": g10b, "g20": g20, "g9": g9, "g10": g10, "g11": g11, "g12": g12, "g13": g13, "g14": g14, "g15": g15, "g16": g16, "g17": g17, "g18": g18}[sys.argv[1]]()
        sys.exit(0)
    g1()
    g2_g3()
    unet_case("g5_unet_dim8_p16", dim=8, P=16, B=2, tvals=[3, 50])
    unet_case("g5b_unet_dim16_p32", dim=16, P=32, B=2, tvals=[0, 99], full_out=True)
    unet_case("g6_unet_dim32_p64", dim=32, P=64, B=2, tvals=[17, 78], full_out=False)
    g7("g7_loss_dim8_p16", dim=8, P=16, B=3, tvals=[0, 37, 99])
    g7("g7b_loss_dim32_p64", dim=32, P=64, B=2, tvals=[5, 60])
    g8("g8_sampler_dim8_p16", dim=8, P=16, B=2, n_steps=5)
    g9()
    g10()
    g11()
    g12()
    g13()
    g14()
    g15()
    g16()
    g17()
    g18()
    g19()
    g10b()
    g20()
    g21()
    g22()
    g23()
    g24()
    print("golden vectors written to", OUT)
