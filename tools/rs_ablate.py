"""Where does a row of conv3x3_rs_kernel (k_conv_rs.hip) spend its time?  Variant libraries with parts of the row removed (wrong results).
python tools/rs_ablate.py build 0 1 2 3 7 15 16   (here)   |   python tools/rs_ablate.py [batch]   (GPU box)"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "physicsinformeddiffusionmodels_amd", "csrc")
ABL = os.path.join(ROOT, "tools", "_abl")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(ABL, exist_ok=True)
    subprocess.run(["make", "-C", CSRC, "-j8", "all"], check=True, stdout=subprocess.DEVNULL)
    objs = [os.path.join(CSRC, "build", f) for f in os.listdir(os.path.join(CSRC, "build")) if f.endswith(".o") and not f.startswith("k_conv_rs")]
    procs = []
    for f in sys.argv[2:]:
        o = os.path.join(ABL, f"k_conv_rs_v{f}.o")
        procs.append((f, o, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", f"-DPIDM_RSF_ABLATE={f}",
                                              "-I" + os.path.join(ROOT, "include"), "-x", "hip", "-c", os.path.join(CSRC, "k_conv_rs.hip"), "-o", o])))
    for f, o, pr in procs:
        assert pr.wait() == 0
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ABL, f"librs_v{f}.so"), o] + objs + ["-ldl"], check=True)
        os.remove(o)
    print("built", sorted(os.listdir(ABL)))
    sys.exit(0)
import torch  # noqa: E402
sys.path.insert(0, ROOT)
from physicsinformeddiffusionmodels_amd._lib import ConvDesc, PidmLib, ptr, stream_ptr  # noqa: E402
dev = torch.device("cuda:0"); st = stream_ptr(dev)
SHAPES = [(64, 32, 32), (64, 64, 32), (64, 32, 64), (32, 64, 64)]
libs = sorted((f for f in os.listdir(ABL) if f.startswith("librs_v")), key=lambda f: int(f[7:-3]))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
print(f"batch {B}, us per forward launch; flags: 1 = no epilogue pieces, 2 = no split, 4 = no activation loads, 8 = no fragment reads, 16 = no MFMAs")
for (H, Cin, Cout) in SHAPES:
    row = [f"{H}x{H} {Cin:3d}->{Cout:3d}:"]
    for f in libs:
        L = PidmLib(os.path.join(ABL, f))
        d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cin, C1=0, ld0=Cin, ld1=0, Cout=Cout, KH=3, KW=3, stride=1, pad=1, transposed=0, out_nchw=0, ldo=Cout)
        x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
        wp = torch.zeros(L.pidm_conv_packed_weight_floats(d), device=dev); L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
        out = torch.empty(B, H, H, Cout, device=dev)
        fn = lambda: L.check(L.pidm_conv_forward(d, ptr(x), None, ptr(wp), ptr(bias), None, ptr(out), st))
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        row.append(f"[{f[7:-3]:>2}] {e0.elapsed_time(e1) / 50 * 1e3:6.1f}")
    print("  ".join(row))
