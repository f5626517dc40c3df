"""Run ONE conv shape (fwd, dgrad, wgrad) a few times - for rocprofv3 --pmc passes.
python tools/bench_one.py H C0 C1 Cout K stride pad transposed [B] [reps]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import ConvDesc, get_lib, ptr, stream_ptr  # noqa: E402
H, C0, C1, Cout, K, s, p, tr = [int(a) for a in sys.argv[1:9]]
B = int(sys.argv[9]) if len(sys.argv) > 9 else 64
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 3
L = get_lib(); dev = torch.device("cuda:0"); st = stream_ptr(dev)
Cin = C0 + C1
Ho = H * 2 if tr else (H + 2 * p - K) // s + 1
d = ConvDesc(B=B, Hi=H, Wi=H, C0=C0, C1=C1, ld0=C0, ld1=C1, Cout=Cout, KH=K, KW=K, stride=s, pad=p, transposed=tr, out_nchw=0, ldo=Cout)
x0 = torch.randn(B, H, H, C0, device=dev); x1 = torch.randn(B, H, H, C1, device=dev) if C1 else None
w = torch.randn((Cin, Cout, K, K) if tr else (Cout, Cin, K, K), device=dev) * 0.05
bias = torch.randn(Cout, device=dev)
wp = torch.empty(L.pidm_conv_packed_weight_floats(d), device=dev); wd = torch.empty(L.pidm_conv_dgrad_packed_weight_floats(d), device=dev)
L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st)); L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wd), 1, st))
out = torch.empty(B, Ho, Ho, Cout, device=dev); dy = torch.randn(B, Ho, Ho, Cout, device=dev); dx = torch.empty(B, H, H, Cin, device=dev)
dw = torch.empty_like(w); db = torch.empty(Cout, device=dev); ws = torch.empty(L.pidm_conv_wgrad_ws(d), dtype=torch.uint8, device=dev)
for _ in range(reps):
    L.check(L.pidm_conv_forward(d, ptr(x0), ptr(x1), ptr(wp), ptr(bias), None, ptr(out), st))
    L.check(L.pidm_conv_dgrad(d, ptr(dy), Cout, ptr(wd), None, ptr(dx), Cin, st))
    L.check(L.pidm_conv_wgrad(d, ptr(x0), ptr(x1), ptr(dy), Cout, ptr(dw), ptr(db), ptr(ws), st))
torch.cuda.synchronize()
