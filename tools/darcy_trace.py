"""Cycle stamps of darcy_stream_kernel (workgroup 0, wave 0): where a sample's time goes.
Needs a measurement build of k_darcy.hip with -DPIDM_DARCY_TRACE_BUILD=1 linked into a library of its own, e.g.
  cd physicsinformeddiffusionmodels_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DPIDM_DARCY_TRACE_BUILD=1 \
     -x hip -c k_darcy.hip -o /tmp/k_darcy_trace.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libpidm_darcy_trace.so \
     /tmp/k_darcy_trace.o $(ls build/*.o | grep -v k_darcy) -ldl
python tools/darcy_trace.py <library> [B]"""
import ctypes as C_
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import PidmLib, ptr, stream_ptr  # noqa: E402
from oracle import pidm_oracle as O  # noqa: E402
L = PidmLib(os.path.abspath(sys.argv[1])); dev = torch.device("cuda:0"); P = 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
fs = O.darcy_source_field(P).reshape(-1).contiguous().to(dev)
tab = O.diffusion_tables(100)
tw, tv = tab["p2_loss_weight"].to(dev), tab["posterior_variance_clipped"].to(dev)
g = torch.Generator().manual_seed(3)
x0 = torch.randn(B, 2, P, P, generator=g).to(dev); pred = x0 + 0.3 * torch.randn(B, 2, P, P, generator=g).to(dev)
t = torch.randint(0, 100, (B,), generator=g).to(dev)
res = torch.empty(B, P * P, 3, device=dev); grad = torch.empty_like(pred); sc = torch.empty(4, device=dev)
ws = torch.empty(L.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
for _ in range(4):
    L.check(L.pidm_darcy_loss_fwd_bwd_t(ptr(x0), ptr(pred), ptr(fs), ptr(t), ptr(tw), ptr(tv), 1.0, 1e-3, float(P - 1), -float(P - 1),
                                        ptr(res), ptr(grad), ptr(sc), ptr(ws), B, P, stream_ptr(dev)))
torch.cuda.synchronize()
buf = (C_.c_ulonglong * 256)()
L.lib.pidm_debug_darcy_trace.argtypes = [C_.POINTER(C_.c_ulonglong)]
assert L.lib.pidm_debug_darcy_trace(buf) == 0
s = list(buf)
print(f"darcy_stream_kernel B={B}: cycles of workgroup 0 / wave 0 per sample")
print(" sample | wait+top barrier | pass 0+1 | 2nd barrier | pass 2 + sums | total")
for i in range(64):
    a = s[4 * i:4 * i + 4]
    if a[3] == 0:
        break
    prev_end = s[4 * i - 1] if i else a[0]
    print(f" {i:6d} | {a[0] - prev_end:16d} | {a[1] - a[0]:8d} | {a[2] - a[1]:11d} | {a[3] - a[2]:13d} | {a[3] - prev_end}")
