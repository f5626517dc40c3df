"""Where does a short-K 3x3 convolution launch spend its time?  (measurement aid)
    python tools/conv_probe.py build     # here: variant libraries tools/_abl/libpidm_abl{0,1,2,3,4}.so (k_conv.hip with -DPIDM_ABLATE_FLAGS=n)
    python tools/conv_probe.py           # GPU box: times the 64x64 32->32 3x3 forward conv for B in {16,32,64,128,256} with
                                         # every variant (1 = no operand loads, 2 = no stores, 4 = no MFMAs, 3 = neither loads nor stores)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "physicsinformeddiffusionmodels_amd", "csrc")
ABL = os.path.join(ROOT, "tools", "_abl")
FLAGS = [0, 1, 2, 3, 4, 7]

if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(ABL, exist_ok=True)
    subprocess.run(["make", "-C", CSRC, "-j8", "all"], check=True, stdout=subprocess.DEVNULL)
    objs = [os.path.join(CSRC, "build", f) for f in os.listdir(os.path.join(CSRC, "build")) if f.endswith(".o") and not f.startswith("k_conv")]
    for f in FLAGS:
        o = os.path.join(ABL, f"k_conv_{f}.o")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", f"-DPIDM_ABLATE_FLAGS={f}",
                        "-x", "hip", "-c", os.path.join(CSRC, "k_conv.hip"), "-o", o], check=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ABL, f"libpidm_abl{f}.so"), o] + objs, check=True)
        os.remove(o)
    print("built", sorted(os.listdir(ABL)))
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, ROOT)
from physicsinformeddiffusionmodels_amd._lib import ConvDesc, PidmLib, ptr, stream_ptr  # noqa: E402

dev = torch.device("cuda:0")
st = stream_ptr(dev)
SHAPES = [(64, 32, 32, 3), (64, 64, 32, 3), (32, 64, 64, 3), (16, 128, 128, 3)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (H, Cin, Cout, K) in SHAPES:
    print(f"--- {H}x{H} {Cin}->{Cout} k{K}: us per launch (TFLOP/s) by ablation flags; MFMA-only floor = flops / 157.3 TF")
    for B in (16, 32, 64, 128, 256):
        if B * H * H * max(Cin, Cout) * 4 > 3e9:
            continue
        flops = 2.0 * B * H * H * Cout * Cin * K * K
        row = [f"B={B:4d} floor {flops / 157.3e6:6.1f}us |"]
        for f in FLAGS:
            L = PidmLib(os.path.join(ABL, f"libpidm_abl{f}.so"))
            d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cin, C1=0, ld0=Cin, ld1=0, Cout=Cout, KH=K, KW=K, stride=1, pad=K // 2, transposed=0, out_nchw=0, ldo=Cout)
            x = torch.randn(B, H, H, Cin, device=dev)
            w = torch.randn(Cout, Cin, K, K, device=dev) * 0.05
            bias = torch.randn(Cout, device=dev)
            wp = torch.zeros(L.pidm_conv_packed_weight_floats(d), device=dev)
            L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
            out = torch.empty(B, H, H, Cout, device=dev)
            t = timeit(lambda: L.pidm_conv_forward(d, ptr(x), None, ptr(wp), ptr(bias), None, ptr(out), st))
            row.append(f"f{f}: {t:6.1f} ({flops / t / 1e6:5.1f})")
        print("  ".join(row), flush=True)
