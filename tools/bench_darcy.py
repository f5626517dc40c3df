"""Fused Darcy residual + loss kernel alone (csrc/k_darcy.hip): us per launch pair and algorithmic GB/s (112 KiB per 64x64 sample)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import get_lib, ptr, stream_ptr
from oracle import pidm_oracle as O
L = get_lib(); dev = torch.device("cuda:0"); P = 64
fs = O.darcy_source_field(P).reshape(-1).contiguous().to(dev)
tab = O.diffusion_tables(100)
tw, tv = tab["p2_loss_weight"].to(dev), tab["posterior_variance_clipped"].to(dev)
for B in ([int(a) for a in sys.argv[1:]] or (64, 256, 1024, 4096)):
    g = torch.Generator().manual_seed(B)
    x0 = torch.randn(B, 2, P, P, generator=g).to(dev)
    pad = [torch.empty(int(os.environ["BENCH_DARCY_PAD"]), dtype=torch.uint8, device=dev)] if os.environ.get("BENCH_DARCY_PAD") else []     # shifts the later buffers
    pred = x0 + (0.3 * torch.randn(B, 2, P, P, generator=g).to(dev) if os.environ.get("BENCH_DARCY_NOISE") else 0.1)
    t = torch.randint(0, 100, (B,), generator=g).to(dev)
    if pad: pad.append(torch.empty(int(os.environ["BENCH_DARCY_PAD"]) + 4096, dtype=torch.uint8, device=dev))
    res = torch.empty(B, P * P, 3, device=dev)
    if pad: pad.append(torch.empty(int(os.environ["BENCH_DARCY_PAD"]) + 8192, dtype=torch.uint8, device=dev))
    grad = torch.empty_like(pred); sc = torch.empty(4, device=dev)
    if os.environ.get("BENCH_DARCY_ADDR"): print("addresses mod 2^21:", [hex(z.data_ptr() % (1 << 21)) for z in (x0, pred, res, grad)], [hex(z.data_ptr()) for z in (x0, pred, res, grad)])
    ws = torch.empty(L.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
    call = lambda: L.check(L.pidm_darcy_loss_fwd_bwd_t(ptr(x0), ptr(pred), ptr(fs), ptr(t), ptr(tw), ptr(tv), 1.0, 1e-3, float(P - 1), -float(P - 1),
                                                      ptr(res), ptr(grad), ptr(sc), ptr(ws), B, P, stream_ptr(dev)))
    for _ in range(int(os.environ.get("BENCH_DARCY_WARMUP", "3"))): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30; e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"PIDM_DARCY_ROWS={os.environ.get('PIDM_DARCY_ROWS','16')} B={B:5d}: {us:8.2f} us  {B*112*1024/us/1e3:8.1f} GB/s")
