"""Sample the GPU clocks (rocm-smi) while one kernel class runs in a loop: is the fp32-MFMA conv running at the 2.4 GHz the
157.3 TFLOP/s peak assumes?  python tools/clock_probe.py"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicsinformeddiffusionmodels_amd._lib import ConvDesc, get_lib, ptr, stream_ptr  # noqa: E402
L = get_lib(); dev = torch.device("cuda:0"); st = stream_ptr(dev)
B, H, C = 64, 32, 64
d = ConvDesc(B=B, Hi=H, Wi=H, C0=C, C1=0, ld0=C, ld1=0, Cout=C, KH=3, KW=3, stride=1, pad=1, transposed=0, out_nchw=0, ldo=C)
x = torch.randn(B, H, H, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05; bias = torch.zeros(C, device=dev)
wp = torch.empty(L.pidm_conv_packed_weight_floats(d), device=dev); L.check(L.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
out = torch.empty(B, H, H, C, device=dev)
samples = []
stop = False
def probe():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            keep = [l.strip() for l in o.splitlines() if ("sclk" in l or "mclk" in l or "Power" in l) and "GPU[0]" in l]
            samples.append(" | ".join(k.split(":", 1)[-1].strip() for k in keep))
        except Exception as e:  # noqa: BLE001
            samples.append(repr(e))
        time.sleep(0.25)
def run(fn, secs, label):
    global stop, samples
    samples = []; stop = False
    th = threading.Thread(target=probe); th.start()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(200): fn()
        torch.cuda.synchronize(); n += 200
    el = time.perf_counter() - t0
    stop = True; th.join()
    print(f"== {label}: {el / n * 1e6:.1f} us per launch")
    for s in samples[2:10]: print("   ", s)
flops = 2.0 * B * H * H * C * C * 9
run(lambda: L.pidm_conv_forward(d, ptr(x), None, ptr(wp), ptr(bias), None, ptr(out), st), 3.0, "conv 3x3 64->64 @32x32 (fp32 MFMA)")
a = torch.randn(1 << 26, device=dev); b2 = torch.empty_like(a)
run(lambda: b2.copy_(a), 2.0, "256 MB copy (HBM)")
