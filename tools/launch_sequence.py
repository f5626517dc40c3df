"""Print the launch sequence of one Darcy training step (forward, loss, backward) - kernel names in order, launch by launch
(PIDM_TRACE_LAUNCHES=1 + the per-kernel profile hooks, which switch graph replay off).  `--emu`: on the host-emulated build
(no GPU; the launchers choose tiles by shape, so use the real image size and width: 64x64, dim 32).
    python tools/launch_sequence.py [--emu] [--batch B] > seq.txt"""
import argparse
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ap = argparse.ArgumentParser()
ap.add_argument("--emu", action="store_true")
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--dim", type=int, default=32)
ap.add_argument("--image", type=int, default=64)
a = ap.parse_args()
os.environ["PIDM_TRACE_LAUNCHES"] = "1"
from physicsinformeddiffusionmodels_amd._lib import get_lib, stream_ptr  # noqa: E402
from physicsinformeddiffusionmodels_amd.data_utils import synthetic_darcy_batch  # noqa: E402
from physicsinformeddiffusionmodels_amd.denoising_utils import DenoisingDiffusion  # noqa: E402
from physicsinformeddiffusionmodels_amd.residuals_darcy import ResidualsDarcy  # noqa: E402
from physicsinformeddiffusionmodels_amd.unet_model import Unet3D  # noqa: E402
if a.emu:
    from tests.emu_util import emu_lib
    lib, dev = emu_lib(), torch.device("cpu")
else:
    lib, dev = get_lib(), torch.device("cuda:0")
klib = lib if a.emu else None
torch.manual_seed(0)
m = Unet3D(dim=a.dim, channels=2).to(dev)
m._pidm_lib = klib
diff = DenoisingDiffusion(100, dev, lib=klib)
res = ResidualsDarcy(model=m, fd_acc=2, pixels_per_dim=a.image, pixels_at_boundary=True, reverse_d1=True, device=dev, bcs='none',
                     domain_length=1., lib=klib)
batch = synthetic_darcy_batch(a.batch, a.image, seed=1, device=dev)
diff.deferred_scalars = True
lib.check(lib.pidm_prof_kernels_begin(stream_ptr(dev)))
print("==== step", file=sys.stderr)
loss, *_ = diff.model_estimation_loss(batch, residual_func=res, c_data=1., c_residual=1e-3, c_ineq=0., lambda_opt=0.)
print("==== backward", file=sys.stderr)
loss.backward()
buf = C.create_string_buffer(1 << 16)
lib.pidm_prof_kernels_collect(buf, len(buf))
print(buf.value.decode())
