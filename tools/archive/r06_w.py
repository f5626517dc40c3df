"""round 6: the streaming Darcy loss kernel at batch 4096 against the band kernel, every bit"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from physicsinformeddiffusionmodels_amd._lib import get_lib, ptr, stream_ptr, reload_knobs
from oracle import pidm_oracle as O
L = get_lib(); dev = torch.device("cuda:0"); P = 64; B = 4096
fs = O.darcy_source_field(P).reshape(-1).contiguous().to(dev)
tab = O.diffusion_tables(100)
tw, tv = tab["p2_loss_weight"].to(dev), tab["posterior_variance_clipped"].to(dev)
g = torch.Generator().manual_seed(3)
x0 = torch.randn(B, 2, P, P, generator=g).to(dev); pred = x0 + 0.3 * torch.randn(B, 2, P, P, generator=g).to(dev)
t = torch.randint(0, 100, (B,), generator=g).to(dev)
def run(env, reps=1):
    for k in list(os.environ):
        if k.startswith("PIDM_DARCY"): os.environ.pop(k)
    os.environ.update(env); reload_knobs()
    res = torch.empty(B, P * P, 3, device=dev); grad = torch.empty_like(pred); sc = torch.empty(4, device=dev)
    ws = torch.empty(L.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
    call = lambda: L.check(L.pidm_darcy_loss_fwd_bwd_t(ptr(x0), ptr(pred), ptr(fs), ptr(t), ptr(tw), ptr(tv), 1.0, 1e-3, float(P - 1), -float(P - 1),
                                                      ptr(res), ptr(grad), ptr(sc), ptr(ws), B, P, stream_ptr(dev)))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): call()
    e1.record(); torch.cuda.synchronize()
    return res, grad, sc, e0.elapsed_time(e1) * 1e3 / 30
ref = run({"PIDM_DARCY_FULL": "0"})
print(f"band: {ref[3]:.1f} us")
for name, env in (("full", {"PIDM_DARCY_STREAM": "0"}), ("stream", {}), ("stream 512 wgs", {"PIDM_DARCY_STREAM_WGS": "512"})):
    for rep in range(3):
        got = run(env)
        print(f"{name}: {got[3]:.1f} us  residual equal {torch.equal(got[0], ref[0])}  gradient equal {torch.equal(got[1], ref[1])}  "
              f"scalars {(got[2] - ref[2]).abs().max().item() / ref[2].abs().max().item():.1e}", flush=True)
