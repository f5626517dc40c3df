"""Cycle stamps of the split 3x3 weight-gradient kernel (workgroup 0, thread 0).  python tools/wgrad_trace.py [H Cin Cout B]"""
import ctypes as C
import os
import sys
os.environ["PIDM_STREAM_TRACE"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import ConvDesc, get_lib, ptr, stream_ptr  # noqa: E402
H, Cin, Cout, B = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (64, 32, 32, 64)
L = get_lib(); dev = torch.device("cuda:0"); st = stream_ptr(dev)
d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cin, C1=0, ld0=Cin, ld1=0, Cout=Cout, KH=3, KW=3, stride=1, pad=1, transposed=0, out_nchw=0, ldo=Cout)
x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, H, H, Cout, device=dev)
dw = torch.empty(Cout, Cin, 3, 3, device=dev); db = torch.empty(Cout, device=dev)
ws = torch.empty(L.pidm_conv_wgrad_ws(d) // 4 + 64, device=dev)
for _ in range(3):
    L.check(L.pidm_conv_wgrad(d, ptr(x), None, ptr(dy), Cout, ptr(dw), ptr(db), ptr(ws), st))
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert L.pidm_debug_stream_trace(buf) == 0
t = list(buf); n = int(t[255])
print(f"{H}x{H} {Cin}->{Cout} B={B}: prologue {t[1] - t[0]} cycles; per tile [wait barrier 1 | staging | wait barrier 2 | k-steps]; epilogue {t[n] - t[n - 1]}; total {t[n] - t[0]}")
for i in range(1, n - 4, 5):
    a = t[i:i + 5]
    print(f"  tile: {a[1] - a[0]:6d} | {a[2] - a[1]:6d} | {a[3] - a[2]:6d} | {a[4] - a[3]:6d}")
