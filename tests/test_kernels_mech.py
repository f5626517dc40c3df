"""Matrix-free mechanics residual K(rho)u - f, compliance, volume shift, bilinear resizes (csrc/k_mech.hip) vs the
golden vector produced by the reference's DENSE 8450x8450 assembly (tests/golden/g9_mechanics.npz) and vs the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pidm_oracle as O
from physicsinformeddiffusionmodels_amd.residuals_mechanics_K import ResidualsMechanics, resize_image

G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_index_tables_bit_exact():
    g = np.load(os.path.join(G, "g9_mechanics.npz"))
    res = ResidualsMechanics(model=None, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder="/nonexistent/", lib=object())
    np.testing.assert_array_equal(res.stiffs.elem_dofs32.numpy(), g["elem_dofs"])          # DME assembly table
    np.testing.assert_array_equal(res.stiffs.glob_assembler_idcs[:, :8, 1].numpy(), g["elem_dofs"])
    assert rel(res.stiffs.tot_local_stiffness[0], g["kloc0"]) < 1e-6
    assert res.stiffs.neq == 8450 and res.stiffs.kloc_stride == 0


def test_mechanics_residual_vs_reference_dense_K(backend):
    L, dev = backend
    g = np.load(os.path.join(G, "g9_mechanics.npz"))
    res = ResidualsMechanics(model=None, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder="/nonexistent/", device=dev,
                             lib=L if dev.type == "cpu" else None)
    x0 = torch.from_numpy(g["x0"]).to(dev).requires_grad_(True)
    bcs = torch.from_numpy(g["bcs"]).to(dev)
    vf = torch.from_numpy(g["vf"]).to(dev)
    out = res.compute_residual((x0, bcs, vf), reduce="none", return_model_out=True, return_optimizer=True, return_inequality=True,
                               pass_through=True)
    assert rel(out["residual"], g["residual"]) < 2e-6
    assert rel(out["model_out"], g["model_out"]) < 3e-6
    assert rel(out["optimizer"], g["compliance"]) < 1e-5      # residual-loss parity target: 1e-5 rel
    assert rel(out["inequality"], g["shift"]) < 1e-5
    scal = (out["residual"] * torch.from_numpy(g["wr"]).to(dev)).sum() + (out["model_out"] * torch.from_numpy(g["wm"]).to(dev)).sum() \
        + 0.7 * out["optimizer"].sum() + 1.3 * (out["inequality"] ** 2).sum()
    (gx,) = torch.autograd.grad(scal, x0)
    assert rel(gx, g["grad_x0"]) < 5e-6


@pytest.mark.parametrize("hi,ho", [(65, 64), (64, 65), (16, 17)])
def test_bilinear_resize(backend, hi, ho):
    L, dev = backend
    x = torch.randn(2, 3, hi, hi, generator=torch.Generator().manual_seed(1))
    y = resize_image(x.to(dev), ho, L if dev.type == "cpu" else None)
    ref = F.interpolate(x, size=(ho, ho), mode="bilinear", align_corners=False)
    assert rel(y, ref) < 3e-6
