"""Micro-benchmark of the implicit-GEMM conv kernels on the real GPU (HIP events on the launch stream).
Usage: python tools/bench_conv.py [B]   -> one line per UNet conv shape: us, TFLOP/s (fwd, dgrad, wgrad)."""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import ConvDesc, get_lib, ptr, stream_ptr  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = get_lib()
dev = torch.device("cuda:0")
st = stream_ptr(dev)
SHAPES = [  # H, C0, C1, Cout, K, stride, pad, transposed
    (64, 2, 0, 32, 7, 1, 3, 0), (64, 32, 0, 32, 3, 1, 1, 0), (64, 32, 32, 32, 3, 1, 1, 0), (64, 32, 0, 768, 1, 1, 0, 0),
    (64, 256, 0, 32, 1, 1, 0, 0), (64, 32, 0, 32, 4, 2, 1, 0), (32, 32, 0, 64, 3, 1, 1, 0), (32, 64, 0, 64, 3, 1, 1, 0),
    (32, 64, 64, 32, 3, 1, 1, 0), (16, 64, 0, 128, 3, 1, 1, 0), (16, 128, 0, 128, 3, 1, 1, 0), (8, 128, 0, 256, 3, 1, 1, 0),
    (8, 256, 0, 256, 3, 1, 1, 0), (8, 256, 256, 128, 3, 1, 1, 0), (8, 128, 0, 128, 4, 2, 1, 1), (1, 128, 0, 4000, 1, 1, 0, 0),
]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


if os.environ.get("BENCH_CONV_SHAPES"):      # e.g. "64,128,0,768,1,1,0,0;16,256,0,128,1,1,0,0"
    SHAPES = [tuple(int(v) for v in sh.split(",")) for sh in os.environ["BENCH_CONV_SHAPES"].split(";")]
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
totf = 0.0
for (H, C0, C1, Cout, K, s, p, tr) in SHAPES:
    Cin = C0 + C1
    Ho = H * 2 if tr else (H + 2 * p - K) // s + 1
    d = ConvDesc(B=B, Hi=H, Wi=H, C0=C0, C1=C1, ld0=C0, ld1=C1, Cout=Cout, KH=K, KW=K, stride=s, pad=p, transposed=tr,
                 out_nchw=0, ldo=Cout)
    x0 = torch.randn(B, H, H, C0, device=dev)
    x1 = torch.randn(B, H, H, C1, device=dev) if C1 else None
    w = torch.randn((Cin, Cout, K, K) if tr else (Cout, Cin, K, K), device=dev) * 0.05
    bias = torch.randn(Cout, device=dev)
    wp = torch.empty(L.pidm_conv_packed_weight_floats(d), device=dev)
    wd = torch.empty(L.pidm_conv_dgrad_packed_weight_floats(d), device=dev)
    L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
    L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wd), 1, st))
    out = torch.empty(B, Ho, Ho, Cout, device=dev)
    dy = torch.randn(B, Ho, Ho, Cout, device=dev)
    dx = torch.empty(B, H, H, Cin, device=dev)
    dw = torch.empty_like(w)
    db = torch.empty(Cout, device=dev)
    ws = torch.empty(L.pidm_conv_wgrad_ws(d), dtype=torch.uint8, device=dev)
    flops = 2.0 * B * (H * H if tr else Ho * Ho) * Cout * Cin * (4 if tr else K * K)
    t_f = timeit(lambda: L.pidm_conv_forward(d, ptr(x0), ptr(x1), ptr(wp), ptr(bias), None, ptr(out), st))
    t_d = timeit(lambda: L.pidm_conv_dgrad(d, ptr(dy), Cout, ptr(wd), None, ptr(dx), Cin, st))
    if tr and C1:
        t_w = float("nan")
    else:
        t_w = timeit(lambda: L.pidm_conv_wgrad(d, ptr(x0), ptr(x1), ptr(dy), Cout, ptr(dw), ptr(db), ptr(ws), st))
    tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w; totf += flops
    print(f"H={H:3d} Cin={Cin:4d} Cout={Cout:4d} K={K} s={s} tr={tr}  GF={flops/1e9:8.2f} | fwd {t_f:9.1f}us {flops/t_f/1e6:6.1f}TF"
          f" | dgrad {t_d:9.1f}us {flops/t_d/1e6:6.1f}TF | wgrad {t_w:9.1f}us {flops/t_w/1e6:6.1f}TF", flush=True)
print("TOTAL", {k: f"{v:.0f}us {totf/v/1e6:.1f}TF" for k, v in tot.items()})
