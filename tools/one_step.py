"""One Darcy training step at batch B (default 64) after W warm-up steps - for traces (PIDM_TRACE_CONV, rocprofv3 --kernel-trace)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd.data_utils import synthetic_darcy_batch  # noqa: E402
from physicsinformeddiffusionmodels_amd.denoising_utils import DenoisingDiffusion  # noqa: E402
from physicsinformeddiffusionmodels_amd.residuals_darcy import ResidualsDarcy  # noqa: E402
from physicsinformeddiffusionmodels_amd.unet_model import Unet3D  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Unet3D(dim=32, channels=2).to(dev)
diff = DenoisingDiffusion(100, dev)
res = ResidualsDarcy(model=m, fd_acc=2, pixels_per_dim=64, pixels_at_boundary=True, reverse_d1=True, device=dev)
batch = synthetic_darcy_batch(B, 64, seed=1, device=dev)
for i in range(W + 1):
    if i == W:
        torch.cuda.synchronize()
        print("==== traced step ====", file=sys.stderr, flush=True)
    loss, *_ = diff.model_estimation_loss(batch, residual_func=res, c_data=1., c_residual=1e-3)
    for p in m.parameters():
        p.grad = None
    loss.backward()
torch.cuda.synchronize()
print("loss", float(loss))
