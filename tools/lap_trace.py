"""Cycle stamps of lap_bwd_kernel (workgroup 0, waves 0 and 4 = one SIMD): where a (tile, head) unit's time goes.
Needs a measurement build of k_attn_proj.hip with -DPIDM_LAP_TRACE_BUILD=1 (the stamps cost the C = 32 kernel its last registers;
e.g. `make -C physicsinformeddiffusionmodels_amd/csrc HIPFLAGS+=-DPIDM_LAP_TRACE_BUILD=1` after touching the file).
python tools/lap_trace.py [B] [H] [heads] [C]"""
import ctypes as C_
import os
import sys
os.environ["PIDM_LAP_TRACE"] = "1"
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
L.check(L.pidm_lap_forward(ptr(xn), ptr(wq), ptr(wo), ptr(bo), ptr(resid), ptr(y), ptr(saved), ptr(qstat), C, B, N, heads, ptr(ws), st), "fwd")
for _ in range(3):
    L.check(L.pidm_lap_backward(ptr(xn), ptr(gy), ptr(wq), ptr(wo), ptr(saved), ptr(qstat), ptr(dxn), ptr(dwq), ptr(dwo), C, B, N, heads, ptr(ws), st), "bwd")
torch.cuda.synchronize()
buf = (C_.c_ulonglong * 256)()
assert L.pidm_debug_lap_trace(buf) == 0
t = list(buf)
names = ["half 1: proj + softmax", "d_xn issued", "dW share", "half 2: proj + softmax", "d_xn issued", "dW share", "write share", "barrier 1", "head sum", "barrier 2"]
print(f"lap_bwd B={B} H={H} heads={heads} C={C}: cycles per phase of a tile round, wave 0 || wave 4")
print("  round | " + " | ".join(f"{n[:14]:>14s}" for n in names) + " | total")
for rnd in range(8):
    rows = []
    for w in (0, 1):
        s = t[128 * w + 16 * rnd:128 * w + 16 * rnd + 11]
        if s[0] == 0 or s[10] == 0:
            rows = None
            break
        rows.append(" | ".join(f"{s[i + 1] - s[i]:14d}" for i in range(10)) + f" | {s[10] - s[0]}")
    if rows is None:
        break
    print(f"  {rnd:5d} | " + rows[0])
    print(f"        | " + rows[1])
