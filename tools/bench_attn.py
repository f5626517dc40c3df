"""Linear attention (pidm_linear_attention_forward/backward) at the Darcy model's four levels: time and effective HBM rate.
python tools/bench_attn.py [B]     algorithmic bytes: fwd = k + (k,v) + q reads + out write; bwd = (q,dA) + (q,k,v,dA) reads + dqkv write"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import get_lib, ptr, stream_ptr  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = get_lib(); dev = torch.device("cuda:0"); st = stream_ptr(dev)
heads, HD = 8, 256
tot = [0.0, 0.0]
LEVELS = (int(sys.argv[2]),) if len(sys.argv) > 2 else (64, 32, 16, 8)     # optional second argument: only this level
for H in LEVELS:
    N = H * H
    qkv = torch.randn(B, N, 3 * HD, device=dev); dA = torch.randn(B, N, HD, device=dev)
    out = torch.empty(B, N, HD, device=dev); dqkv = torch.empty(B, N, 3 * HD, device=dev)
    kstat = torch.empty(B * HD * 2, device=dev); ctx = torch.empty(B * heads * 1024, device=dev); qstat = torch.empty(B * N * heads * 2, device=dev)
    ws = torch.empty(L.pidm_linear_attention_ws(B, N, heads), dtype=torch.uint8, device=dev)
    fwd = lambda: L.check(L.pidm_linear_attention_forward(ptr(qkv), ptr(out), ptr(kstat), ptr(ctx), ptr(qstat), B, N, heads, ptr(ws), st))
    bwd = lambda: L.check(L.pidm_linear_attention_backward(ptr(qkv), ptr(kstat), ptr(qstat), ptr(ctx), ptr(dA), ptr(dqkv), B, N, heads, ptr(ws), st))
    Cout = {64: 32, 32: 64, 16: 128}.get(H)
    fns = [fwd, bwd]
    if Cout:
        w = torch.randn(Cout, HD, device=dev) * 0.1; bias = torch.randn(Cout, device=dev); xres = torch.randn(B, N, Cout, device=dev)
        y = torch.empty(B, N, Cout, device=dev); dy = torch.randn(B, N, Cout, device=dev); dw = torch.empty(Cout, HD, device=dev)
        ws2 = torch.empty(L.pidm_linear_attention_out_backward_ws(B, N, heads, Cout), dtype=torch.uint8, device=dev)
        fns.append(lambda: L.check(L.pidm_linear_attention_out_forward(ptr(qkv), ptr(w), ptr(bias), ptr(xres), ptr(y), Cout, ptr(kstat), ptr(ctx), ptr(qstat), B, N, heads, ptr(ws2), st)))
        fns.append(lambda: L.check(L.pidm_linear_attention_out_backward(ptr(qkv), ptr(kstat), ptr(qstat), ptr(ctx), ptr(dy), Cout, ptr(w), Cout, ptr(dqkv), ptr(dw), B, N, heads, ptr(ws2), st)))
    res = []
    for f in fns:
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 if len(sys.argv) > 2 else 20
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / reps * 1e3)
    px = B * N * 4.0
    fb, bb = px * HD * 5, px * HD * (2 + 4 + 3)
    tot[0] += res[0]; tot[1] += res[1]
    fused = f" | fused(+to_out) fwd {res[2]:7.1f}us bwd {res[3]:7.1f}us" if Cout else ""
    print(f"H={H:3d} fwd {res[0]:7.1f}us {fb / res[0] / 1e6:5.2f}TB/s | bwd {res[1]:7.1f}us {bb / res[1] / 1e6:5.2f}TB/s" + fused)
print(f"TOTAL fwd {tot[0]:.0f}us bwd {tot[1]:.0f}us")
