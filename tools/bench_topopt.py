"""Timing of the topology-optimisation evaluation block on the GPU (B samples, 64x64 mesh): matrix-free fp64 PCG solve,
floating-material labelling.  python tools/bench_topopt.py [B=64]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import get_lib, ptr, stream_ptr  # noqa: E402
from physicsinformeddiffusionmodels_amd.residuals_mechanics_K import ResidualsMechanics  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0"); L = get_lib(); nel = 64; nn = 65
res = ResidualsMechanics(model=None, pixels_per_dim=nel, pixels_at_boundary=True, no_BC_folder="/nonexistent/", device=dev, topopt_eval=True)
st = res.stiffs
g = torch.Generator().manual_seed(0)
yy = torch.arange(nel).view(1, nel, 1)
rho = (0.1 + 0.8 * ((yy - 32).abs() < 12).float() + 0.05 * torch.randn(B, nel, nel, generator=g)).contiguous().to(dev)
bcs = torch.zeros(B, 4, nn, nn); bcs[:, 0, :, 0] = 1; bcs[:, 1, :, 0] = 1; bcs[:, 3, 32, 64] = -0.01
bcs = bcs.to(dev)
comp = torch.empty(B, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); rr = torch.empty(B, device=dev)
ws = torch.empty(L.pidm_mech_solve_ws_bytes(nel, B), dtype=torch.uint8, device=dev)
def solve():
    L.check(L.pidm_mech_solve(ptr(rho), ptr(bcs), ptr(st.kloc_dev), st.kloc_stride, ptr(st.elem_dofs32), ptr(st.dof_elems32), nel,
                              0.5, 1.0, 1e-3, 20000, 1e-9, None, ptr(comp), None, ptr(it), ptr(rr), ptr(ws), B, stream_ptr(dev)))
solve(); torch.cuda.synchronize()
t0 = time.perf_counter(); solve(); torch.cuda.synchronize(); t1 = time.perf_counter()
n = torch.empty(B, dtype=torch.int32, device=dev)
L.check(L.pidm_floating_material(ptr(rho), 0.5, nel, ptr(n), B, stream_ptr(dev))); torch.cuda.synchronize()
t2 = time.perf_counter(); L.check(L.pidm_floating_material(ptr(rho), 0.5, nel, ptr(n), B, stream_ptr(dev))); torch.cuda.synchronize(); t3 = time.perf_counter()
print(json.dumps({"B": B, "pcg_solve_ms": round((t1 - t0) * 1e3, 2), "pcg_iterations_max": int(it.max()), "pcg_iterations_mean": float(it.float().mean()),
                  "us_per_iteration": round((t1 - t0) * 1e6 / int(it.max()), 2), "relres_max": float(rr.max()),
                  "floating_material_ms": round((t3 - t2) * 1e3, 3), "components": n[:4].tolist()}))
