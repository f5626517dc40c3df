"""Linear-elasticity residual callback for topology optimisation: host-side mirror of reference
`src/residuals_mechanics_K.py::{StiffnessMatrix, ResidualsMechanics, resize_image}`.

The reference assembles a dense 8450 x 8450 stiffness matrix PER SAMPLE (285.6 MB, src/residuals_mechanics_K.py:
208-218) and multiplies it with u.  Here `compute_residual` is one matrix-free gfx950 kernel (csrc/k_mech.hip: every
dof gathers its <= 4 incident elements in fixed order) and its hand-written adjoint; the bilinear 64<->65 resizes
(torchvision Resize, antialias=False) are folded into the same kernels.  No CPU fallback.

The `topopt_eval and sample` evaluation block (reference :276-347) runs on the GPU as well: the
per-sample dense `torch.linalg.solve` becomes a matrix-free Jacobi-PCG in fp64 (one workgroup per sample), the OpenCV
connected-components call a label-propagation kernel (`pidm_mech_apply / pidm_mech_solve / pidm_floating_material`).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from ._lib import PidmError, get_lib, ptr, stream_ptr
from .unet_model import generalized_b_xy_c_to_image, generalized_image_to_b_xy_c


def resize_image(tensor, target_size, lib=None):
    """Bilinear resize of [B, c..., X, Y] to target_size x target_size (torchvision Resize(antialias=False) on
    tensors == F.interpolate(mode='bilinear', align_corners=False)); reference :10-21.  Forward only."""
    assert tensor.dim() > 3, f"Expected image, got {tensor.shape}"
    if tensor.requires_grad:
        raise PidmError("resize_image is a forward-only kernel (network inputs); differentiable resizes are fused into "
                        "the mechanics residual kernel")
    lib = lib or get_lib()
    shp = tensor.shape
    x = tensor.contiguous().float()
    bc = int(np.prod(shp[:-2]))
    out = torch.empty(*shp[:-2], target_size, target_size, dtype=torch.float32, device=x.device)
    lib.check(lib.pidm_bilinear_resize(ptr(x), ptr(out), bc, shp[-1], target_size, stream_ptr(x.device)), "pidm_bilinear_resize")
    return out


def quad4_plane_stress_stiffness(coord, E=1.0, nu=0.3):
    """8x8 stiffness of a 4-node bilinear quad (plane stress, 2x2 Gauss, dofs [u1x,u1y,..,u4x,u4y], nodes CCW) -
    what solidspy.uelutil.elast_quad4 computes for the reference (src/residuals_mechanics_K.py:99-103)."""
    C = E / (1.0 - nu ** 2) * np.array([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, (1.0 - nu) / 2.0]])
    gp = 1.0 / np.sqrt(3.0)
    k = np.zeros((8, 8))
    coord = np.asarray(coord, dtype=float)
    for r in (-gp, gp):
        for s in (-gp, gp):
            dN = 0.25 * np.array([[-(1 - s), (1 - s), (1 + s), -(1 + s)], [-(1 - r), -(1 + r), (1 + r), (1 - r)]])
            J = dN @ coord
            dNdx = np.linalg.solve(J, dN)
            Bm = np.zeros((3, 8))
            Bm[0, 0::2] = dNdx[0]
            Bm[1, 1::2] = dNdx[1]
            Bm[2, 0::2] = dNdx[1]
            Bm[2, 1::2] = dNdx[0]
            k += np.linalg.det(J) * (Bm.T @ C @ Bm)
    return k


def synthetic_mesh(nel=64):
    """nodes [nn*nn, 5] and elements [nel*nel, 7] in SolidsPy's text-file layout for the regular mesh the dataset
    uses: node id = row*(nel+1)+col at (x=col, y=nel-row), all dofs free, element nodes CCW [bl, br, tr, tl]."""
    nn = nel + 1
    r, c = np.meshgrid(np.arange(nn), np.arange(nn), indexing="ij")
    nodes = np.stack([r * nn + c, c, nel - r, 0 * r, 0 * r], axis=-1).reshape(-1, 5).astype(float)
    er, ec = np.meshgrid(np.arange(nel), np.arange(nel), indexing="ij")
    eles = np.stack([er * nel + ec, 1 + 0 * er, 0 * er, (er + 1) * nn + ec, (er + 1) * nn + ec + 1, er * nn + ec + 1,
                     er * nn + ec], axis=-1).reshape(-1, 7).astype(int)
    return nodes, eles


class StiffnessMatrix:
    """Element stiffness + assembly tables (reference :23-103).  Reads SolidsPy `nodes.txt` / `eles.txt` from
    `no_BC_folder` when present, otherwise builds the regular nel x nel mesh (the reference's mesh files are not
    shipped with the repository)."""

    def __init__(self, no_BC_folder, nels_per_side=64, ndof=8, device='cpu', dtype=torch.float32):
        self.ndof = ndof
        self.nels = nels_per_side ** 2
        self.nel = nels_per_side
        if no_BC_folder and os.path.exists(os.path.join(no_BC_folder, 'nodes.txt')):
            nodes = np.loadtxt(os.path.join(no_BC_folder, 'nodes.txt'), ndmin=2)
            elements = np.loadtxt(os.path.join(no_BC_folder, 'eles.txt'), ndmin=2, dtype=int)
        else:
            nodes, elements = synthetic_mesh(nels_per_side)
        if np.any(nodes[:, -2:] != 0):
            raise NotImplementedError('the matrix-free kernel expects the "no BC" mesh (all dofs free)')
        nn = nels_per_side + 1
        if nodes.shape[0] != nn * nn or elements.shape[0] != self.nels:
            raise NotImplementedError('mesh must be nels_per_side x nels_per_side quads')
        kloc = np.stack([quad4_plane_stress_stiffness(nodes[elements[e, 3:], 1:3], 1.0, 0.3) for e in range(self.nels)])
        self.neq = 2 * nodes.shape[0]
        elem_dofs = (2 * elements[:, 3:, None] + np.arange(2)[None, None, :]).reshape(self.nels, 8)   # DME, all free
        uniform = bool(np.abs(kloc - kloc[0]).max() < 1e-12)
        # inverse table: for every dof the (element, local index) pairs that touch it, in ascending element order
        dof_elems = -np.ones((self.neq, 4, 2), dtype=np.int32)
        fill = np.zeros(self.neq, dtype=np.int64)
        for e in range(self.nels):
            for a in range(8):
                i = elem_dofs[e, a]
                if fill[i] >= 4:
                    raise NotImplementedError('more than 4 elements share a node')
                dof_elems[i, fill[i]] = (e, a)
                fill[i] += 1
        self.tot_local_stiffness = torch.tensor(kloc, dtype=dtype, device=device)
        self.glob_assembler = torch.tensor(elem_dofs, dtype=torch.int64, device=device)
        self.indices_ext = torch.cartesian_prod(torch.arange(ndof), torch.arange(ndof)).to(device)
        self.glob_assembler_idcs = self.glob_assembler[:, self.indices_ext]
        # kernel-side tables
        self.kloc_dev = self.tot_local_stiffness[:1].contiguous() if uniform else self.tot_local_stiffness.contiguous()
        self.kloc_stride = 0 if uniform else 64
        self.elem_dofs32 = torch.tensor(elem_dofs, dtype=torch.int32, device=device).contiguous()
        self.dof_elems32 = torch.tensor(dof_elems, dtype=torch.int32, device=device).contiguous()

    def to(self, device):
        for k in ('kloc_dev', 'elem_dofs32', 'dof_elems32', 'tot_local_stiffness', 'glob_assembler', 'indices_ext',
                  'glob_assembler_idcs'):
            setattr(self, k, getattr(self, k).to(device))
        return self

    def image_to_stiffness_coord(self, image_coord, dof, tot_dofs=2):
        b, h, w = image_coord.shape
        out = torch.zeros((b, h * w, tot_dofs), dtype=image_coord.dtype, device=image_coord.device)
        out[:, :, dof] = image_coord.reshape(b, h * w)
        return out.reshape(b, h * w * tot_dofs)

    def stiffness_to_image_coord(self, stiffness_flat, dof, tot_dofs=2):
        if stiffness_flat.dim() == 1:
            stiffness_flat = stiffness_flat.unsqueeze(0)
        nodes = stiffness_flat.shape[1] // tot_dofs
        n = int(round(nodes ** 0.5))
        assert n * n == nodes, "The number of nodes is not a perfect square."
        return stiffness_flat.reshape(stiffness_flat.shape[0], n, n, tot_dofs)[:, :, :, dof]


class _MechResidualFn(torch.autograd.Function):
    """(residual [B,ndof], model_out [B,3,65,65], compliance [B], shift [B]) = M(x0_pred [B,3,64,64]; bcs, vf)."""

    @staticmethod
    def forward(ctx, x0_pred, bcs, vf, stiffs, lib):
        x = x0_pred.contiguous().float()
        bc = bcs.contiguous().float()
        v = vf.contiguous().float()
        B, _, nel, _ = x.shape
        nn = nel + 1
        dev = x.device
        res = torch.empty(B, 2 * nn * nn, dtype=torch.float32, device=dev)
        mo = torch.empty(B, 3, nn, nn, dtype=torch.float32, device=dev)
        cs = torch.empty(B, 2, dtype=torch.float32, device=dev)
        lib.check(lib.pidm_mech_residual_fwd(ptr(x), ptr(bc), ptr(v), ptr(stiffs.kloc_dev), stiffs.kloc_stride,
                                             ptr(stiffs.elem_dofs32), ptr(stiffs.dof_elems32), nel, ptr(res), ptr(mo), ptr(cs),
                                             B, stream_ptr(dev)), "pidm_mech_residual_fwd")
        ctx.save_for_backward(x, bc)
        ctx.meta = (stiffs, lib)
        return res, mo, cs[:, 0].contiguous(), cs[:, 1].contiguous()

    @staticmethod
    def backward(ctx, g_res, g_mo, g_comp, g_shift):
        x, bc = ctx.saved_tensors
        stiffs, lib = ctx.meta
        B, _, nel, _ = x.shape
        dev = x.device
        g_res = g_res.contiguous().float() if g_res is not None else torch.zeros(B, stiffs.neq, device=dev)
        g_mo_c = g_mo.contiguous().float() if g_mo is not None else None
        gcs = torch.stack([g_comp if g_comp is not None else torch.zeros(B, device=dev),
                           g_shift if g_shift is not None else torch.zeros(B, device=dev)], dim=1).contiguous().float()
        gx = torch.empty_like(x)
        lib.check(lib.pidm_mech_residual_bwd(ptr(x), ptr(bc), ptr(stiffs.kloc_dev), stiffs.kloc_stride, ptr(stiffs.elem_dofs32),
                                             ptr(stiffs.dof_elems32), nel, ptr(g_res), ptr(g_mo_c), ptr(gcs), ptr(gx), B,
                                             stream_ptr(dev)), "pidm_mech_residual_bwd")
        return gx, None, None, None, None


class _MechLossFn(torch.autograd.Function):
    """(loss, scalars[8]) of the mechanics training loss and, saved for backward, d loss / d x0_pred - one fused kernel
    (`pidm_mech_loss_fwd_bwd`, src/denoising_utils.py:666-708).  scalars = (loss, data loss, mean |r|, mean shift, mean
    compliance, 0, 0, 0)."""

    @staticmethod
    def forward(ctx, x0_pred, target, bcs, vf, p2w, inv_var, inv_var_sum, c_data, c_residual, c_ineq, lambda_opt, stiffs, lib):
        x = x0_pred.contiguous().float()
        B, _, nel, _ = x.shape
        dev = x.device
        if tuple(target.shape) != (B, 3, nel + 1, nel + 1) or tuple(bcs.shape) != (B, 4, nel + 1, nel + 1):
            raise PidmError(f"mechanics loss: target / bcs must be [B,3,{nel + 1},{nel + 1}] / [B,4,{nel + 1},{nel + 1}]")
        grad = torch.empty_like(x)
        out = torch.empty(8, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.pidm_mech_loss_ws(B), dtype=torch.uint8, device=dev)
        lib.check(lib.pidm_mech_loss_fwd_bwd(ptr(x), ptr(target.contiguous().float()), ptr(bcs.contiguous().float()),
                                             ptr(vf.contiguous().float()), ptr(p2w), ptr(inv_var), ptr(inv_var_sum), float(c_data), float(c_residual),
                                             float(c_ineq), float(lambda_opt), ptr(stiffs.kloc_dev), stiffs.kloc_stride,
                                             ptr(stiffs.elem_dofs32), ptr(stiffs.dof_elems32), nel, ptr(grad), ptr(out), ptr(ws), B,
                                             stream_ptr(dev)), "pidm_mech_loss_fwd_bwd")
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        (grad,) = ctx.saved_tensors
        return (grad * g_loss,) + (None,) * 12


class ResidualsMechanics:
    """Drop-in for reference ResidualsMechanics (src/residuals_mechanics_K.py:105-367), matrix-free."""

    def __init__(self, model, pixels_per_dim, pixels_at_boundary, no_BC_folder, device='cpu', bcs='none', E=1.0, nu=0.3,
                 topopt_eval=False, use_ddim_x0=False, ddim_steps=0, lib=None):
        self.gov_eqs = 'mechanics'
        self.model = model
        self.stiffs = StiffnessMatrix(no_BC_folder=no_BC_folder, nels_per_side=pixels_per_dim, device=device, dtype=torch.float32)
        self.deriv_mode = None
        self.pixels_at_boundary = pixels_at_boundary
        self.E, self.nu = E, nu
        self.periodic = bcs == 'periodic'
        self.device = device
        self.pixels_per_dim = pixels_per_dim
        self.use_trapezoid = bool(pixels_at_boundary)
        self.topopt_eval = topopt_eval
        self.use_ddim_x0 = use_ddim_x0
        self.ddim_steps = ddim_steps
        self._lib = lib

    @property
    def lib(self):
        if self._lib is None:
            self._lib = get_lib()
        return self._lib

    # evaluation-only knobs of the FE solve (the reference uses a direct dense solve, :321-323)
    pcg_max_iter = 20000
    pcg_rtol = 1e-9

    def topopt_metrics(self, rho_pred, bcs, vf, solution):
        """Reference :276-347: compliance error of the BINARISED predicted density (true displacements by an FE solve)
        relative to the compliance of the data solution, volume-fraction error, floating-material flag.
        rho_pred [B,nel,nel]; bcs [B,4,nn,nn]; vf [B]; solution [B,3,nn,nn] = (u_x, u_y, rho_simp zero-padded)."""
        lib, st = self.lib, self.stiffs
        dev = rho_pred.device
        B, nel = rho_pred.shape[0], self.pixels_per_dim
        nn = nel + 1
        bc = bcs.contiguous().float()
        opt_disp = solution[:, :2].contiguous().float()
        rho_simp = solution[:, 2, :-1, :-1].contiguous().float()
        mesh = (ptr(st.kloc_dev), st.kloc_stride, ptr(st.elem_dofs32), ptr(st.dof_elems32), nel)
        # compliance of the data and the "residual of opt_disp should be zero" sanity check (:293-296)
        res_data = torch.empty(B, st.neq, dtype=torch.float32, device=dev)
        comp_data = torch.empty(B, dtype=torch.float32, device=dev)
        lib.check(lib.pidm_mech_apply(ptr(rho_simp), ptr(opt_disp), ptr(bc), *mesh, ptr(res_data), ptr(comp_data), B,
                                      stream_ptr(dev)), "pidm_mech_apply")
        assert torch.isclose(res_data.abs().mean(), torch.tensor(0., device=dev), atol=1.e-5), 'Residual of opt_disp is not zero.'
        # FE solve on the binarised prediction (:299-323)
        rho = rho_pred.contiguous().float()
        comp_true = torch.empty(B, dtype=torch.float32, device=dev)
        rho_mean = torch.empty(B, dtype=torch.float32, device=dev)
        iters = torch.empty(B, dtype=torch.int32, device=dev)
        relres = torch.empty(B, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.pidm_mech_solve_ws_bytes(nel, B), dtype=torch.uint8, device=dev)
        lib.check(lib.pidm_mech_solve(ptr(rho), ptr(bc), *mesh, 0.5, 1.0, 1.e-3, int(self.pcg_max_iter), float(self.pcg_rtol),
                                      None, ptr(comp_true), ptr(rho_mean), ptr(iters), ptr(relres), ptr(ws), B,
                                      stream_ptr(dev)), "pidm_mech_solve")
        self.last_solve_info = {'iterations': iters, 'relative_residual': relres}
        ncomp = torch.empty(B, dtype=torch.int32, device=dev)
        lib.check(lib.pidm_floating_material(ptr(rho), 0.5, nel, ptr(ncomp), B, stream_ptr(dev)), "pidm_floating_material")
        vfd = vf.to(dev).float()
        return {'rel_CE_error_full_batch': (comp_true - comp_data) / comp_data,
                'vf_error_full_batch': torch.abs(rho_mean - vfd) / vfd,
                # cv2.connectedComponents counts the background label too: "!= 2" <=> not exactly one foreground component
                'fm_error_full_batch': (ncomp != 1).to(torch.int64).cpu()}

    def compute_residual(self, input_tuple, reduce='none', return_model_out=False, return_optimizer=False,
                         return_inequality=False, sample=False, ddim_func=None, pass_through=False):
        self.deriv_mode = 'stiffness'
        input, bcs, vf = input_tuple[0], input_tuple[1], input_tuple[2]
        lib = self.lib
        if pass_through:
            assert isinstance(input, torch.Tensor), 'Input is assumed to directly be given output.'
            x0_pred = input
            mo_src = None
        else:
            mo_src = None
            assert len(input) == 2 and isinstance(input, tuple), 'Input must be a tuple consisting of noisy signal and time.'
            noisy_in, time = input
            P = self.pixels_per_dim
            noisy_img = generalized_b_xy_c_to_image(noisy_in).detach()
            net_in = torch.cat((resize_image(noisy_img, P, lib), resize_image(bcs.detach(), P, lib)), dim=1)   # 10 channels
            if self.use_ddim_x0:
                # x0_pred = model(x_t, 0) feeds the residual; model_out = model(x_t, t) is what the data loss and the
                # sampler's posterior mean see (reference :192-195, 246-256)
                x0_pred, mo_src = ddim_func(net_in, time, self.model, noisy_img.shape, self.ddim_steps, 0., gov_eqs='mechanics')
            else:
                x0_pred = self.model(net_in, time)
        assert x0_pred.dim() == 4, 'Model output must be a tensor shaped as an image.'
        if self.stiffs.kloc_dev.device != x0_pred.device:
            self.stiffs.to(x0_pred.device)
        residual, model_out, compliance, shift = _MechResidualFn.apply(x0_pred, bcs, vf, self.stiffs, lib)
        if mo_src is not None and return_model_out:
            # displacements resized to the nodal grid + zero-padded density of the FIRST model call (differentiable: the same
            # kernel pair, only its model_out output / input gradient are used)
            _, model_out, _, _ = _MechResidualFn.apply(mo_src, bcs, vf, self.stiffs, lib)
        output = {'residual': residual}
        if return_model_out:
            output['model_out'] = model_out
        if return_optimizer:
            output['optimizer'] = compliance
        if return_inequality:
            output['inequality'] = shift
        if self.topopt_eval and sample:
            with torch.no_grad():
                output.update(self.topopt_metrics(x0_pred[:, -1].detach(), bcs, vf, input_tuple[3]))
        if reduce == 'full':
            return {k: v.mean() for k, v in output.items()}
        elif reduce == 'per-batch':
            return {k: v.mean(dim=tuple(range(1, v.ndim))) if v.ndim > 1 and (k != 'model_out' and k != 'residual') else v
                    for k, v in output.items()}
        elif reduce == 'none':
            return output
        raise ValueError('Unknown reduction method.')
