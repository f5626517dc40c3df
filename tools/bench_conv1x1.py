"""1x1 convolutions of the qkv-form attention levels and res_convs: split-form GEMM (conv1x1_split_kernel) vs the fp32 kernel, us and TFLOP/s.
python tools/bench_conv1x1.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import ConvDesc, get_lib, ptr, stream_ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = get_lib(); dev = torch.device("cuda:0"); st = stream_ptr(dev)
SHAPES = [(16, 128, 0, 768), (16, 768, 0, 128), (16, 64, 0, 768), (16, 128, 0, 256), (16, 256, 0, 128), (16, 128, 128, 128), (16, 64, 0, 256),
          (8, 256, 0, 768), (8, 768, 0, 256), (8, 128, 0, 768), (8, 256, 256, 256), (32, 64, 64, 64), (64, 64, 0, 128)]
for (H, C0, C1, Cout) in SHAPES:
    Cin = C0 + C1
    x0 = torch.randn(B, H, H, C0, device=dev); x1 = torch.randn(B, H, H, C1, device=dev) if C1 else None
    w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5; bias = torch.randn(Cout, device=dev); res = torch.randn(B, H, H, Cout, device=dev)
    out = torch.empty(B, H, H, Cout, device=dev)
    d = ConvDesc(B=B, Hi=H, Wi=H, C0=C0, C1=C1, ld0=C0, ld1=C1, Cout=Cout, KH=1, KW=1, stride=1, pad=0, transposed=0, out_nchw=0, ldo=Cout)
    res_us = []
    for split in ("1", "0"):
        os.environ["PIDM_CONV_SPLIT"] = split; L.pidm_reload_knobs()
        wp = torch.empty(L.pidm_conv_packed_weight_floats(d), device=dev)
        L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
        f = lambda: L.check(L.pidm_conv_forward(d, ptr(x0), ptr(x1), ptr(wp), ptr(bias), ptr(res), ptr(out), st))
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res_us.append(e0.elapsed_time(e1) / 20 * 1e3)
    gf = 2.0 * B * H * H * Cin * Cout / 1e9
    mb = 4.0 * B * H * H * (Cin + 2 * Cout) / 1e6
    print(f"{H:3d}x{H:<3d} {Cin:4d}->{Cout:4d}  split {res_us[0]:7.1f} us {gf/res_us[0]*1e3:6.1f} TF | fp32 {res_us[1]:7.1f} us {gf/res_us[1]*1e3:6.1f} TF | {mb:6.1f} MB min traffic = {mb/5e3*1e3:5.1f} us at 5 TB/s")
