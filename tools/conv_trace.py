"""Cycle stamps of the split 3x3 convolution (workgroup 0, waves 0 and 4 = the two waves of one SIMD): where a stage's time goes.
python tools/conv_trace.py [H Cin Cout B]"""
import ctypes as C
import os
import sys
os.environ["PIDM_STREAM_TRACE"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import ConvDesc, PidmLib, get_lib, ptr, stream_ptr  # noqa: E402
H, Cin, Cout, B = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (64, 32, 32, 64)
L = PidmLib(os.environ['PIDM_BENCH_LIB']) if os.environ.get('PIDM_BENCH_LIB') else get_lib(); dev = torch.device("cuda:0"); st = stream_ptr(dev)
d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cin, C1=0, ld0=Cin, ld1=0, Cout=Cout, KH=3, KW=3, stride=1, pad=1, transposed=0, out_nchw=0, ldo=Cout)
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
wp = torch.zeros(L.pidm_conv_packed_weight_floats(d), device=dev); L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
out = torch.empty(B, H, H, Cout, device=dev)
for _ in range(3):
    L.check(L.pidm_conv_forward(d, ptr(x), None, ptr(wp), ptr(bias), None, ptr(out), st))
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert L.pidm_debug_stream_trace(buf) == 0
t = list(buf)
print(f"{H}x{H} {Cin}->{Cout} B={B}: per stage, wave 0 | wave 4: [taps | epilogue | barrier wait | total]   (54 MFMAs per wave = 2 x 54 x 32 pipe cycles per SIMD)")
t0 = t[0]
for s in range(32):
    row = []
    for w in (0, 1):
        a, b, c, e = t[128 * w + 4 * s:128 * w + 4 * s + 4]
        if a == 0:
            row = None
            break
        row.append(f"start {a - t0:7d}: {b - a:6d} | {c - b:6d} | {e - c:6d} | {e - a:6d}")
    if row is None:
        break
    print(f"  stage {s:2d}: " + "   ||   ".join(row))
