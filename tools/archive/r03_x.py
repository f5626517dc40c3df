"""1x1 split GEMM vs the fp32 kernel on the same inputs at large pixel counts (debugging aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import ConvDesc, get_lib, ptr, stream_ptr
L = get_lib(); dev = torch.device("cuda:0"); st = stream_ptr(dev)
def run(B, H, C0, C1, Cout, split):
    os.environ["PIDM_CONV_SPLIT"] = "1" if split else "0"
    g = torch.Generator().manual_seed(B + H + C0 + Cout)
    Cin = C0 + C1
    x0 = torch.randn(B, H, H, C0, generator=g).to(dev)
    x1 = torch.randn(B, H, H, C1, generator=g).to(dev) if C1 else None
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    res = torch.randn(B, H, H, Cout, generator=g).to(dev)
    d = ConvDesc(B=B, Hi=H, Wi=H, C0=C0, C1=C1, ld0=C0, ld1=C1, Cout=Cout, KH=1, KW=1, stride=1, pad=0, transposed=0, out_nchw=0, ldo=Cout)
    wp = torch.empty(L.pidm_conv_packed_weight_floats(d), device=dev)
    L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
    out = torch.full((B, H, H, Cout), float("nan"), device=dev)
    L.check(L.pidm_conv_forward(d, ptr(x0), ptr(x1), ptr(wp), ptr(bias), ptr(res), ptr(out), st))
    torch.cuda.synchronize()
    return out
for (B, H, C0, C1, Cout) in [(64, 16, 128, 0, 768), (256, 16, 128, 0, 768), (256, 16, 768, 0, 128), (256, 8, 256, 0, 768), (256, 32, 64, 64, 64), (256, 16, 128, 128, 128), (128, 16, 128, 0, 768), (200, 16, 128, 0, 256)]:
    a = run(B, H, C0, C1, Cout, True); b = run(B, H, C0, C1, Cout, False)
    diff = (a - b).abs()
    bad = (diff > 1e-3 * b.abs().max()).nonzero()
    print(B, H, C0, C1, Cout, "max diff", diff.max().item(), "nan", torch.isnan(a).sum().item(), "bad", bad.shape[0], bad[:3].tolist(), bad[-2:].tolist() if bad.shape[0] else "")
