"""Projected linear attention (pidm_lap_forward / pidm_lap_backward, csrc/k_attn_proj.hip): time per call at one level.
python tools/bench_lap.py [B] [H] [heads] [C]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import get_lib, ptr, stream_ptr  # noqa: E402
a = [int(v) for v in sys.argv[1:]] + [None] * 4
B, H, heads, C = a[0] or 64, a[1] or 64, a[2] or 8, a[3] or 32
L = get_lib(); dev = torch.device("cuda:0"); st = stream_ptr(dev)
N, HD = H * H, heads * 32
xn = torch.randn(B, N, C, device=dev); resid = torch.randn(B, N, C, device=dev); gy = torch.randn(B, N, C, device=dev)
wq = torch.randn(3 * HD, C, device=dev) * 0.3; wo = torch.randn(C, HD, device=dev) * 0.2; bo = torch.randn(C, device=dev)
y = torch.empty(B, N, C, device=dev); dxn = torch.empty(B, N, C, device=dev); dwq = torch.empty_like(wq); dwo = torch.empty_like(wo)
saved = torch.empty(L.pidm_lap_saved_floats(B, heads, C), device=dev); qstat = torch.empty(B * N * heads * 2, device=dev)
ws = torch.empty(L.pidm_lap_ws(B, N, heads, C), dtype=torch.uint8, device=dev)
fwd = lambda: L.check(L.pidm_lap_forward(ptr(xn), ptr(wq), ptr(wo), ptr(bo), ptr(resid), ptr(y), ptr(saved), ptr(qstat), C, B, N, heads, ptr(ws), st), "fwd")
bwd = lambda: L.check(L.pidm_lap_backward(ptr(xn), ptr(gy), ptr(wq), ptr(wo), ptr(saved), ptr(qstat), ptr(dxn), ptr(dwq), ptr(dwo), C, B, N, heads, ptr(ws), st), "bwd")
import time
out, host = [], []
torch.cuda.Event(enable_timing=True).record(); torch.cuda.synchronize()     # the first event record sets up the runtime's pool (~50 ms of host time)
for f in (fwd, bwd):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(20): f()
    e1.record()
    host.append((time.perf_counter() - t0) / 20 * 1e6)      # host time to enqueue one call
    torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"lap B={B} H={H} heads={heads} C={C} groups={os.environ.get('PIDM_LAP_GROUPS', 'auto')}: fwd {out[0]:.1f} us (host {host[0]:.0f})  "
      f"bwd {out[1]:.1f} us (host {host[1]:.0f})")
