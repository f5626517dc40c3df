"""Secondary measurements (not the headline): C4 mechanics training step, C5 DDPM sampling, EMA swap cost.
python tools/bench_secondary.py [mech_batch=32] [sample_batch=1024] [sample_steps=20]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd.data_utils import synthetic_darcy_batch  # noqa: E402
from physicsinformeddiffusionmodels_amd.denoising_utils import EMA, DenoisingDiffusion  # noqa: E402
from physicsinformeddiffusionmodels_amd.residuals_darcy import ResidualsDarcy  # noqa: E402
from physicsinformeddiffusionmodels_amd.residuals_mechanics_K import ResidualsMechanics  # noqa: E402
from physicsinformeddiffusionmodels_amd.unet_model import Unet3D  # noqa: E402

dev = torch.device("cuda:0")
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SB = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
SS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
out = {}


def timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


# ---- C5: sampling, Darcy, per-step cost of p_sample (UNet fwd + residual + ancestral update), no CPU history ----
torch.manual_seed(0)
model = Unet3D(dim=32, channels=2).to(dev)
diff = DenoisingDiffusion(1000, dev)
res = ResidualsDarcy(model=model, fd_acc=2, pixels_per_dim=64, pixels_at_boundary=True, reverse_d1=True, device=dev)
x = torch.randn(SB, 2, 64, 64, device=dev)
state = {"x": x, "i": 999}


def sample_step():
    (nx, _), _ = diff.p_sample(state["x"], None, state["i"], save_output=False, surpress_noise=True, residual_func=res)
    state["x"] = nx
    state["i"] -= 1


dt = timed(sample_step, SS)
out["C5_sampling"] = {"batch": SB, "ms_per_step": round(dt * 1e3, 2), "sample_steps_per_s": round(SB / dt, 1),
                      "projected_1000_step_s": round(dt * 1000, 1), "fwd_tflops": round(SB * 3.98e9 / dt / 1e12, 1)}

# ---- EMA swap (main.py:178-183,316): update + ema + restore every iteration ----
ema = EMA(0.99)
ema.register(model)


def ema_cycle():
    ema.update(model)
    ema.ema(model)
    ema.restore(model)


out["ema_cycle_ms"] = round(timed(ema_cycle, 5) * 1e3, 2)
del model, res, diff, x, state
torch.cuda.empty_cache()

# ---- C4: mechanics training step (dim=128 UNet, 10 -> 3 channels, matrix-free K u residual) ----
torch.manual_seed(0)
model = Unet3D(dim=128, channels=10, out_dim=3, sigmoid_last_channel=True).to(dev)
diff = DenoisingDiffusion(100, dev)
res = ResidualsMechanics(model=model, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder="/nonexistent/", device=dev,
                         topopt_eval=False)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(5)
inp = torch.zeros(MB, 10, 65, 65)
inp[:, 0] = (0.2 + 0.3 * torch.rand(MB, generator=g)).view(MB, 1, 1)
inp[:, 1:3] = torch.randn(MB, 2, 65, 65, generator=g)
inp[:, 3:5] = 0.1 * torch.randn(MB, 2, 65, 65, generator=g)
inp[:, 5, :64, :64] = torch.rand(MB, 64, 64, generator=g)
inp[:, 6:8, :, 0] = 1.0
inp[:, 9, 32, 64] = -1.0
inp = inp.to(dev)


def mech_step():
    loss, *_ = diff.model_estimation_loss(inp, residual_func=res, c_data=1., c_residual=1e-3, c_ineq=0.1, lambda_opt=0.01)
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.)
    opt.step()


dt = timed(mech_step, 5)
out["C4_mechanics_train"] = {"per_gpu_batch": MB, "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(MB / dt, 1),
                             "step_tflops": round(MB * 141.39e9 / dt / 1e12, 1)}
print(json.dumps(out))
