"""Driver entry points: build() compiles every HIP source for gfx950 in-tree; smoke() runs one tiny training step
of the hot path on cuda:0 and checks it against the CPU oracle."""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "physicsinformeddiffusionmodels_amd", "csrc")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    """hipcc --offload-arch=gfx950 on csrc/*.hip -> csrc/libpidm_hip.so (cross-compiles without a GPU).
    The oracle is pure Python (torch-CPU) and the reference is Python, so there is no oracle/_ref to compile."""
    subprocess.run(["make", "-C", CSRC, "-j8", "all"], check=True)
    so = os.path.join(CSRC, "libpidm_hip.so")
    assert os.path.exists(so), so
    import physicsinformeddiffusionmodels_amd  # noqa: F401
    from physicsinformeddiffusionmodels_amd._lib import PidmLib
    lib = PidmLib(so)
    assert lib.backend == "hip"


def smoke() -> None:
    """One training step (q-sample, UNet fwd, Darcy residual + PIDM loss, UNet bwd) of a small UNet on cuda:0,
    loss and gradient norms checked against the oracle."""
    import torch
    from oracle import pidm_oracle as O
    from physicsinformeddiffusionmodels_amd._lib import get_lib
    from physicsinformeddiffusionmodels_amd.denoising_utils import DenoisingDiffusion
    from physicsinformeddiffusionmodels_amd.residuals_darcy import ResidualsDarcy
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D

    assert torch.cuda.is_available(), "smoke() needs an MI355X"
    assert get_lib().backend == "hip"
    dev = torch.device("cuda:0")
    dim, P, B = 16, 32, 4
    m = Unet3D(dim=dim, channels=2)
    sd = O.fill_state_dict(m.state_dict())
    m.load_state_dict(sd)
    m = m.to(dev)
    diff = DenoisingDiffusion(100, dev)
    res = ResidualsDarcy(model=m, fd_acc=2, pixels_per_dim=P, pixels_at_boundary=True, reverse_d1=True, device=dev)
    g = torch.Generator().manual_seed(7)
    x0 = torch.randn(B, 2, P, P, generator=g)
    x0[:, 1] = torch.exp(0.5 * x0[:, 1])
    eps = torch.randn(B, 2, P, P, generator=g)
    t = torch.tensor([0, 33, 66, 99])
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
    ref, rdata, rabs, _ = O.darcy_training_loss(p, O.UnetCfg(dim=dim, channels=2), O.diffusion_tables(100), x0, t, eps, 1., 1e-3)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-4 * abs(ref.item()), (loss.item(), ref.item())
    assert abs(data_l - rdata.item()) < 2e-4 * abs(rdata.item())
    assert abs(res_l - rabs.item()) < 2e-4 * abs(rabs.item())
    gmax = max(v.grad.norm().item() for v in p.values() if v.grad is not None)
    n = 0
    for k, prm in m.named_parameters():
        if p[k].grad is None:
            assert prm.grad is None, k
            continue
        a, b = prm.grad.norm().item(), p[k].grad.norm().item()
        assert abs(a - b) <= 2e-3 * b + 1e-5 * gmax, (k, a, b)
        n += 1
    print(f"smoke ok: loss {loss.item():.6e} (oracle {ref.item():.6e}), {n} gradient tensors match")


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
