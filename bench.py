"""bench.py - headline benchmark: training samples/sec of the UNet + PDE-residual step, Darcy 64x64 (default workload),
plus the two other BASELINE.json single-GPU workloads behind --workload {darcy,mechanics,sampling}.

A step is the reference's training-loop body (main.py:157-166): model_estimation_loss (q-sample, UNet forward,
Darcy residual, PIDM loss) -> zero_grad -> backward -> [gradient all-reduce when N>1] -> clip_grad_norm_(1.0) ->
Adam.step, on synthetic Darcy-shaped fields already resident in HBM, fp32, default torch init under seed 0.
Workload = BASELINE.json configs[1] (batch 64 per GPU; configs[2] = 8 x 64 under weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

--workload mechanics = BASELINE configs[3] per-GPU share (Unet3D dim=128, 10->3 channels, K.u residual, batch 32 per GPU:
main.py:102-109,126,139); --workload sampling = configs[4] (sample.py:145-150: DDPM ancestral chain, 1000-step schedule,
batch 1024; a step = ONE p_sample step of the whole batch: UNet forward + residual + ancestral update, `value` in
sample-steps/s).  Same JSON schema, roofline and cpu_baseline for all three.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     - the dominant kernel class (implicit-GEMM conv fwd/dgrad + wgrad on the matrix cores):
                 algorithmic FLOPs / HIP-event time of those launches, measured live on the launch stream
  fp32_mfma_only - the same step with every contraction on the fp32 MFMA instead of the split-bf16 form
  eager_scalars  - the same step with the tracked loss terms returned as python floats every step (host sync per step)
  dropin_main_py - the loop body of the reference's main.py:157-183,316 UNCHANGED: python-float loss terms, torch's
                   clip_grad_norm_ + torch.optim.Adam, ema.update / ema.ema / ema.restore every iteration (what a user who only
                   swaps the import path gets; darcy / mechanics, N = 1)
  north_star_b256 - the same headline step at per-GPU batch 256, the configuration north_star quotes its target on (darcy, N = 1)
  residual_only  - the fused Darcy residual + loss kernel alone (SURVEY 8(d) secondary metric) at batch 64 and 4096, GB/s
  exchange       - N > 1 (or PIDM_BENCH_FORCE_EXCHANGE=1): torch.distributed backend, world size, ms per gradient exchange
  mechanics_b32 / sampling_b1024 - BASELINE configs[3] (per-GPU share) and configs[4] measured by the same script in child processes
                   (`--workload mechanics|sampling`): value, ms_per_step, step_flop_fraction (darcy headline run, N = 1)

`--gpus N` with N > 1 and no torch.distributed environment (WORLD_SIZE unset) re-launches itself under torch.distributed.run with
one rank per GPU (127.0.0.1 rendezvous); when WORLD_SIZE is set it must equal --gpus.
  cpu_baseline - the CPU oracle (oracle/pidm_oracle.py, a torch-CPU restatement pinned against the reference)
                 timed on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# multi-process GPU work on this stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle: invalid argument otherwise); the
# variable must be in place before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

FLOPS_PER_SAMPLE_FWD_BWD = 11.916e9   # SURVEY 8(d), FlopCounterMode on the reference, Darcy dim=32 64x64
BYTES_PER_SAMPLE = 148.5e6            # SURVEY 8(d) compulsory-traffic contract at per-GPU batch 256
FLOPS_PER_SAMPLE_FWD = 3.98e9         # Darcy dim=32 forward only (sampling)
FLOPS_PER_SAMPLE_MECH = 141.39e9      # mechanics dim=128, 10->3 channels, fwd+bwd (FlopCounterMode on the reference)
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
LOG_FREQ = 20                         # main.py:153


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("darcy", "mechanics", "sampling"), default="darcy")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 64 darcy = configs[1], 32 mechanics = configs[3] "
                                                         "per GPU, 1024 sampling = configs[4])")
    ap.add_argument("--ema", action="store_true", help="darcy/mechanics: include the EMA update of main.py:178-179 in the step "
                                                       "(folded into the Adam kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager-scalars", action="store_true", help="model_estimation_loss returns python floats every step (the "
                    "reference's types: one host sync per step) instead of floats that synchronise when read")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra run with the 3x3 convolutions on the fp32 MFMA")
    ap.add_argument("--torch-optimizer", action="store_true",
                    help="use torch clip_grad_norm_ + torch.optim.Adam instead of the fused flat clip+Adam kernel (same math)")
    ap.add_argument("--calib-copy", action="store_true",
                    help="also run one 1 GiB device copy (known byte count for the PMC traffic passes, tools/pmc_traffic.sh)")
    return ap.parse_args(argv)


def _oracle_params(m):
    trainable = {k for k, v in m.named_parameters() if v.requires_grad}
    return {k: v.detach().clone().requires_grad_(k in trainable) for k, v in m.state_dict().items()}


def cpu_baseline(workload, batch, steps, threads):
    """Time the CPU oracle (the restatement of the reference pinned by tests/golden) on this box's host cores: the same
    workload at the same per-GPU batch where that fits the time budget (Darcy: B=64), a bounded sample of it otherwise."""
    from oracle import pidm_oracle as O
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D
    from physicsinformeddiffusionmodels_amd.data_utils import synthetic_darcy_batch, synthetic_mechanics_batch
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    tables = O.diffusion_tables(100)
    g = torch.Generator().manual_seed(2)
    t = torch.randint(0, 100, (batch,), generator=g)
    if workload == "sampling":
        m = Unet3D(dim=32, channels=2)
        p = {k: v.detach().clone() for k, v in m.state_dict().items()}
        cfg = O.UnetCfg(dim=32, channels=2)
        tab = O.diffusion_tables(1000)
        x = torch.randn(batch, 2, 64, 64, generator=g)

        def one(it):
            nonlocal x
            with torch.no_grad():
                i = 999 - it
                x0p = O.unet_forward(p, x, torch.full((batch,), i, dtype=torch.long), cfg)
                O.darcy_residual(x0p)
                x = O.p_sample_update(tab, x0p, x, i, torch.randn(batch, 2, 64, 64, generator=g))
        what = "p_sample steps (UNet dim=32 forward, Darcy residual, ancestral update)"
        unit = "sample-steps/s"
    else:
        if workload == "darcy":
            m = Unet3D(dim=32, channels=2)
            cfg = O.UnetCfg(dim=32, channels=2)
            x0 = synthetic_darcy_batch(batch, 64, seed=1)
            eps = torch.randn(batch, 2, 64, 64, generator=g)
            what = "Darcy 64x64 training steps (UNet dim=32 fwd+bwd, residual, loss, clip, Adam)"
        else:
            m = Unet3D(dim=128, channels=10, out_dim=3, sigmoid_last_channel=True)
            cfg = O.UnetCfg(dim=128, channels=10, out_dim=3, sigmoid_last_channel=True)
            inp = synthetic_mechanics_batch(batch, seed=1)
            eps = torch.randn(batch, 3, 65, 65, generator=g)
            kloc, ed = O.q4_plane_stress_stiffness(1.0, 0.3, 1.0), O.synthetic_mesh_element_dofs(64)
            what = "mechanics 64x64 training steps (UNet dim=128 fwd+bwd, matrix-free K.u residual, loss, clip, Adam)"
        p = _oracle_params(m)
        state = {"used": None, "opt": None}
        unit = "samples/s"

        def one(it):
            if workload == "darcy":
                loss = O.darcy_training_loss(p, cfg, tables, x0, t, eps, 1.0, 1e-3)[0]
            else:
                loss = O.mechanics_training_loss(p, cfg, tables, inp, t, eps, kloc, ed, 1.0, 1e-3, 0.1, 0.01)[0]
            for v in p.values():
                v.grad = None
            loss.backward()
            if state["used"] is None:
                state["used"] = [v for v in p.values() if v.grad is not None]
                state["opt"] = torch.optim.Adam(state["used"], lr=1e-4)
            torch.nn.utils.clip_grad_norm_(state["used"], 1.0)
            state["opt"].step()
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        one(it)
        if it > 0:
            times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    torch.set_num_threads(prev_threads)
    return {"value": round(batch / dt, 3), "unit": unit, "cores": threads, "kind": "port",
            "sample": f"{steps} timed (1 warm-up) oracle {what} at batch {batch} on {threads} threads "
                      f"(host has {os.cpu_count()} logical cores), {dt:.2f} s/step"}


def rs_clock_probe(lib, dev, B):
    """The clock the matrix-pipe kernels actually run at.  The 3x3 forward convolution of the 64x64 level (32 -> 32 channels, this
    batch) through the C ABI with PIDM_RS_TRACE=1: workgroup 0 / wave 0 of conv3x3_rs_kernel stamps the shader-clock counter and the
    100 MHz real-time counter around its row loop (k_conv_rs.hip).  Outside the timed region; knobs restored afterwards."""
    import torch
    from physicsinformeddiffusionmodels_amd._lib import ConvDesc, ptr, stream_ptr
    try:
        st = stream_ptr(dev)
        H, Cc = 64, 32
        d = ConvDesc(B=B, Hi=H, Wi=H, C0=Cc, C1=0, ld0=Cc, ld1=0, Cout=Cc, KH=3, KW=3, stride=1, pad=1, transposed=0, out_nchw=0, ldo=Cc)
        x = torch.randn(B, H, H, Cc, device=dev)
        w = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05
        wp = torch.zeros(lib.pidm_conv_packed_weight_floats(d), device=dev)
        lib.check(lib.pidm_conv_pack_weights(d, ptr(w), ptr(wp), 0, st))
        out = torch.empty(B, H, H, Cc, device=dev)
        for _ in range(20):                               # the clock settles under load
            lib.check(lib.pidm_conv_forward(d, ptr(x), None, ptr(wp), None, None, ptr(out), st))
        os.environ["PIDM_RS_TRACE"] = "1"
        lib.pidm_reload_knobs()
        samples = []
        t = (C.c_ulonglong * 4)()
        try:
            with open(os.devnull, "w") as devnull:
                saved = os.dup(2)
                os.dup2(devnull.fileno(), 2)              # (the traced launch also prints its line)
                try:
                    for _ in range(8):
                        lib.check(lib.pidm_conv_forward(d, ptr(x), None, ptr(wp), None, None, ptr(out), st))
                        torch.cuda.synchronize()
                        if lib.pidm_debug_conv_rs_trace(t) == 0 and t[3] > t[1]:
                            samples.append((t[2] - t[0], t[3] - t[1]))
                finally:
                    os.dup2(saved, 2)
                    os.close(saved)
        finally:
            del os.environ["PIDM_RS_TRACE"]
            lib.pidm_reload_knobs()
        if not samples:
            return None
        cyc = sum(s_[0] for s_ in samples) / len(samples)
        ticks = sum(s_[1] for s_ in samples) / len(samples)
        # rows per strip: the launcher halves the image height until 1024 waves exist (k_conv_rs.hip: launch_conv_rs)
        R = H
        while R > 4 and B * (H // 32) * (H // R) < 1024:
            R //= 2
        mfma_rows = R                                     # input rows 0,1 and R, R+1 carry 1/3 + 2/3 of a row's MFMAs each
        return {"kernel": "conv3x3_rs_kernel, 64x64 32->32 forward at this batch, row loop of workgroup 0 / wave 0",
                "shader_clock_ghz": round(cyc / (ticks * 10.0), 3), "nominal_clock_ghz": 2.4,
                "row_loop_us": round(ticks * 0.01, 2), "rows_per_strip": R,
                "mfma_pipe_busy_in_cycles": round(mfma_rows * 108 * 32 / cyc, 3),
                "what": "108 MFMAs of 32 cycles per full input row; busy = MFMA cycles / shader cycles of the row loop.  The two MFMA peaks "
                        "of this object assume the nominal clock: at the measured one the matrix pipe is this busy in cycles"}
    except Exception as e:  # noqa: BLE001 - a probe must not take the bench line down
        return {"error": repr(e)}


def pmc_traffic(workload, batch):
    """(HBM bytes per launch of the conv class, where the figure comes from).  The PMC counters need rocprofv3 around the
    process, so they cannot be read live: the figure is the committed pass of tools/pmc_traffic.sh under profiles/ for this
    workload and batch (FETCH_SIZE / WRITE_SIZE in separate --pmc runs, gfx950 corrections applied); (None, reason) when no pass
    exists."""
    import json as _json
    tag = f"pmc_traffic_b{batch}.json" if workload == "darcy" else f"pmc_traffic_{workload}_b{batch}.json"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", tag)
    try:
        with open(path) as f:
            d = _json.load(f)
        return round(float(d["conv"]["hbm_bytes_per_launch"]), 0), f"committed rocprofv3 --pmc pass profiles/{tag} ({d.get('round', 'round 2')} build), not this run"
    except (OSError, KeyError, ValueError):
        return None, f"no committed PMC pass for workload {workload} at batch {batch} (profiles/{tag} absent)"


def pmc_step_traffic(workload, batch):
    """HBM bytes of the WHOLE step (every kernel class) from the same committed PMC pass, or None."""
    import json as _json
    tag = f"pmc_traffic_b{batch}.json" if workload == "darcy" else f"pmc_traffic_{workload}_b{batch}.json"
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", tag)) as f:
            d = _json.load(f)
        return float(sum(v["hbm_bytes_per_step"] for v in d.values() if isinstance(v, dict) and "hbm_bytes_per_step" in v))
    except (OSError, KeyError, ValueError):
        return None


def residual_only_rates(lib, residuals, diffusion, dev):
    """SURVEY 8(d) secondary metric: the fused Darcy residual + PIDM loss + d loss / d x0_pred kernel alone (csrc/k_darcy.hip), HIP
    events on the launch stream, algorithmic bytes = 32 KB prediction read + 48 KB residual write + 32 KB gradient write per 64x64
    sample (the contract figure; the 32 KB target read of the data term is not counted)."""
    from physicsinformeddiffusionmodels_amd._lib import ptr, stream_ptr
    out = {}
    P = 64
    for B in (64, 4096):
        g = torch.Generator(device="cpu").manual_seed(5)
        x0 = torch.randn(B, 2, P, P, generator=g).to(dev)
        pred = x0 + 0.1
        t = torch.randint(0, diffusion.n_steps, (B,), generator=g).to(dev)
        res = torch.empty(B, P * P, 3, device=dev)
        grad = torch.empty_like(pred)
        sc = torch.empty(4, device=dev)
        ws = torch.empty(lib.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
        dd = diffusion.diff_dict

        def call():
            lib.check(lib.pidm_darcy_loss_fwd_bwd_t(ptr(x0), ptr(pred), ptr(residuals._f_s_flat), ptr(t), ptr(dd['p2_loss_weight']),
                                                    ptr(dd['posterior_variance_clipped']), 1.0, 1e-3, residuals.inv_h0, residuals.inv_h1,
                                                    ptr(res), ptr(grad), ptr(sc), ptr(ws), B, P, stream_ptr(dev)), "darcy loss")
        # steady state: the memory-side clocks take tens of milliseconds of streaming to settle (batch 4096: 169 us per launch after
        # 3 warm-up launches, 152 after 50, 142 after 300 - profiles/r06_darcy_large_batch.txt); ~40 ms of warm-up, 50 timed launches
        for _ in range(300 if B >= 1024 else 100):
            call()
        n = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        nbytes = B * (32 + 48 + 32) * 1024
        out[f"b{B}"] = {"us_per_launch_pair": round(us, 2), "GB/s": round(nbytes / us / 1e3, 1),
                        "frac_of_hbm_peak": round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4)}
    out["what"] = ("batch 64: darcy_quad_kernel<loss> (row bands), batch 4096: darcy_stream_kernel (one persistent workgroup per CU, whole samples "
                   "streamed through LDS; profiles/r06_darcy_large_batch.txt), each + darcy_loss_finalize (residual, loss terms, d loss/d x0_pred; the scalars by a second 6 us launch: "
                   "totalling them in the last-arriving workgroup of the first was measured slower twice, profiles/r05_darcy_one_launch_ab.txt), "
                   "algorithmic 112 KiB per sample, back-to-back launch pairs timed with events on the launch stream after 100 / 300 warm-up launches")
    return out


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) without a torch.distributed environment: launch the N ranks ourselves, exactly as the
    driver would (one process per GPU, 127.0.0.1 rendezvous on a free port), and pass their output through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # (the script that was started: bench.py itself, or the test driver that calls bench.main with an injected library)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd))


def child_leg(workload, steps, warmup):
    """Another BASELINE.json single-GPU configuration measured by this same script in a child process (its own model, workspace and
    hipGraphs; the parent idles meanwhile): {value, unit, ms_per_step, step_flop_fraction, ...} or {error}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-alt", "--no-roofline"]
    env = dict(os.environ, PIDM_BENCH_CHILD="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:  # noqa: BLE001 - the headline line must not die with a leg
        return {"error": f"{type(e).__name__}: {e}"}
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
            "per_gpu_batch": d["config"]["per_gpu_batch"], "step_flop_fraction": d.get("step_flop_fraction"),
            "workload": d["config"]["workload"], "how": "child process of this run: " + " ".join(cmd[1:])}


def main(argv=None, test_env=None):
    """test_env (tests/bench_on_emulator.py only - never a measurement): {"lib": a PidmLib, "image": P, "dim": d, "batch": B} runs
    this function's control flow - rank set-up, sharding, gradient exchange, max-over-ranks timing, JSON - on CPU tensors over gloo
    with the library the test hands in; the JSON line says so (`selftest`)."""
    args = parse(argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the torch.distributed environment has WORLD_SIZE={world}: "
                         f"launch with --nproc-per-node {args.gpus} (or let bench.py spawn the ranks: unset WORLD_SIZE)")
    selftest = test_env is not None
    if selftest:
        lib = test_env["lib"]
        dev = torch.device("cpu")
        args.no_cpu_baseline = args.no_alt = args.no_roofline = True
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    # PIDM_BENCH_SHARE_GPU=1 (debug only): all ranks share cuda:0 and talk over gloo - exercises the N>1 code path on
    # a single-GPU box; real runs use one GPU per rank over RCCL (backend "nccl").
    share = os.environ.get("PIDM_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    if not selftest:
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but this node has {torch.cuda.device_count()} GPU(s)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    force_dp = os.environ.get("PIDM_BENCH_FORCE_EXCHANGE") == "1"    # debug: one rank, but the RCCL exchange path runs
    if world > 1 or force_dp:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dp and world == 1:
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if share or selftest:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    def dev_sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    from physicsinformeddiffusionmodels_amd._lib import get_lib
    from physicsinformeddiffusionmodels_amd.data_utils import synthetic_darcy_batch, synthetic_mechanics_batch
    from physicsinformeddiffusionmodels_amd.denoising_utils import EMA, DenoisingDiffusion
    from physicsinformeddiffusionmodels_amd.parallel import GradientExchange
    from physicsinformeddiffusionmodels_amd.residuals_darcy import ResidualsDarcy
    from physicsinformeddiffusionmodels_amd.residuals_mechanics_K import ResidualsMechanics
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D

    if not selftest:
        lib = get_lib()
    klib = lib if selftest else None     # what the host classes bind (None = the product library)
    if os.environ.get("PIDM_BENCH_STREAM") == "1":      # experiment: everything on a non-default stream (the null stream has
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))   # implicit-synchronisation semantics of its own)
    if os.environ.get("PIDM_BENCH_EAGER") == "1":
        args.eager_scalars = True
    wl = args.workload
    B = args.batch or {"darcy": 64, "mechanics": 32, "sampling": 1024}[wl]
    P_img, dim_darcy = (test_env["image"], test_env["dim"]) if selftest else (64, 32)
    if selftest:
        B = args.batch or test_env["batch"]
    torch.manual_seed(0)                      # identical initial weights on every rank
    train = wl != "sampling"
    if wl == "mechanics":
        model = Unet3D(dim=128, channels=10, out_dim=3, sigmoid_last_channel=True).to(dev)
        diffusion = DenoisingDiffusion(100, dev)
        residuals = ResidualsMechanics(model=model, pixels_per_dim=64, pixels_at_boundary=True, no_BC_folder="/nonexistent/",
                                       device=dev, topopt_eval=False)
        batch = synthetic_mechanics_batch(B, seed=100 + rank, device=dev)
        loss_kw = dict(c_data=1., c_residual=1e-3, c_ineq=0.1, lambda_opt=0.01)
        flops_per_unit, n_lr = FLOPS_PER_SAMPLE_MECH, 1.e-4
    else:
        model = Unet3D(dim=dim_darcy, channels=2).to(dev)
        model._pidm_lib = klib
        diffusion = DenoisingDiffusion(100 if train else 1000, dev, lib=klib)
        residuals = ResidualsDarcy(model=model, fd_acc=2, pixels_per_dim=P_img, pixels_at_boundary=True, reverse_d1=True,
                                   device=dev, bcs='none', domain_length=1., lib=klib)
        batch = synthetic_darcy_batch(B, P_img, seed=100 + rank, device=dev)   # resident in HBM; each rank its own shard
        loss_kw = dict(c_data=1., c_residual=1e-3, c_ineq=0., lambda_opt=0.)
        flops_per_unit, n_lr = (FLOPS_PER_SAMPLE_FWD_BWD if train else FLOPS_PER_SAMPLE_FWD), 1.e-4
    ema = None
    optimizer = exchange = None
    if train:
        if args.ema:
            ema = EMA(0.99)
            ema.register(model)
        if args.torch_optimizer:
            optimizer = torch.optim.Adam(model.parameters(), lr=n_lr)
        else:
            from physicsinformeddiffusionmodels_amd.optim import FusedClipAdam
            optimizer = FusedClipAdam(model, lr=n_lr, max_norm=1., image_size=P_img, ema=ema, ema_start=-1, lib=klib)   # clip_grad_norm_(1.) + Adam
        exchange = (GradientExchange(model, world, image_size=P_img, diffusion=diffusion, force=force_dp, lib=klib)
                    if (world > 1 or force_dp) else None)
    torch.manual_seed(1234 + rank)
    chain = {"x": torch.randn(B, 2, 64, 64, device=dev), "i": 999} if not train else None
    if not train:
        # the steps of this workload are p_sample_loop's iterations (denoising_utils.py: the loop runs inside frozen_weights(model):
        # constant parameters, the weights are packed / split once, not once per step)
        from physicsinformeddiffusionmodels_amd._engine import frozen_weights
        _frozen = frozen_weights(model)
        _frozen.__enter__()

    diffusion.deferred_scalars = not args.eager_scalars
    counter = {"it": 0, "last": None}

    def step(batch_=None, opt_=None, torch_opt=None):
        """One step on `batch_` (default: the headline batch) with optimizer `opt_` (default: the headline one)."""
        if not train:
            # one ancestral step of the whole batch (src/denoising_utils.py:388-455); the chain restarts at t = 999 when it ends
            (nx, _), _ = diffusion.p_sample(chain["x"], None, chain["i"], save_output=False, surpress_noise=True, residual_func=residuals)
            chain["x"] = nx
            chain["i"] = chain["i"] - 1 if chain["i"] > 0 else 999
            return nx
        b_ = batch if batch_ is None else batch_
        o_ = optimizer if opt_ is None else opt_
        use_torch = args.torch_optimizer if torch_opt is None else torch_opt
        loss, *tracked = diffusion.model_estimation_loss(b_, residual_func=residuals, **loss_kw)
        counter["it"] += 1
        if counter["it"] % LOG_FREQ == 0:      # main.py:167-175: the tracked loss terms are read every log_freq = 20 iterations
            counter["last"] = [float(v) for v in tracked]
        o_.zero_grad()
        loss.backward()
        if exchange is not None:
            exchange.allreduce()
        if use_torch:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.)
        o_.step()
        if ema is not None:
            ema.update(model)
        return loss

    def fence():
        if dist is not None:
            dist.barrier()
        dev_sync()

    def timed(fn, n, warm=3):
        """ms per call of n calls of fn between two fences (after `warm` untimed ones), max over ranks"""
        for _ in range(warm):
            fn()
        fence()
        t_ = time.perf_counter()
        for _ in range(n):
            fn()
        fence()
        el_ = time.perf_counter() - t_
        if dist is not None:
            tt_ = torch.tensor([el_], device=dev, dtype=torch.float64)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            el_ = float(tt_.item())
        return el_ / n * 1e3

    if args.calib_copy:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        dst.copy_(src)
        del src, dst
    def launch_counts():
        a = (C.c_longlong * 4)()
        lib.pidm_debug_launch_counts(a)
        return list(a)

    for _ in range(args.warmup):
        step()
    fence()
    lc0 = launch_counts()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enqueued = time.perf_counter() - t0          # the host is done enqueuing; the GPU still works through the queue
    fence()
    elapsed = time.perf_counter() - t0
    lc1 = launch_counts()
    launches = {
        "kernels_enqueued_one_by_one_per_step": round((lc1[0] - lc0[0]) / args.steps, 1),
        "graph_launches_per_step": round((lc1[1] - lc0[1]) / args.steps, 2),
        "kernels_inside_graphs_per_step": round((lc1[2] - lc0[2]) / args.steps, 1),
        "host_enqueue_ms_per_step": round(t_enqueued / args.steps * 1e3, 3),
        "what": "library-side accounting over the timed region (pidm_debug_launch_counts): the UNet forward and - for models below 512 channels "
                "(Darcy; not the mechanics model, whose backward stays launch by launch with its weight gradients on a side stream: "
                "PIDM_GRAPH_BWD) - backward are replayed hipGraphs (PIDM_GRAPH=0 turns that off); kernels enqueued one by one = q-sample, loss, "
                "optimizer kernels and a launch-by-launch backward (torch's own "
                "launches - RNG, the loss-scalar copy - are not counted); host_enqueue = wall time until the host has enqueued a step INSIDE this "
                "loop, where the launch queue is full and the host is held to the GPU's pace - the unloaded cost (queue drained: 1.2 ms per step "
                "with graphs, 2.7-3.4 without) is in profiles/r03_graph_ab.txt",
    }
    per_rank = None
    if dist is not None:
        # every rank's own time for the K steps (barrier + sync on both sides): the job's time is the slowest rank's; the list makes a
        # straggler visible in the line the driver keeps
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        times = [float(t_.item()) for t_ in every]
        elapsed = max(times)
        per_rank = {"ms_per_step": [round(t_ / args.steps * 1e3, 3) for t_ in times],
                    "value": [round(B * args.steps / t_, 2) for t_ in times],
                    "slowest_rank": times.index(max(times)), "ms_per_step_min": round(min(times) / args.steps * 1e3, 3),
                    "ms_per_step_max": round(max(times) / args.steps * 1e3, 3),
                    "what": "each rank's wall time for the timed steps between the two fences and its own shard's samples/s; "
                            "`value` = global batch x steps / the slowest rank's time"}
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed

    # the same step with the 3x3 contractions on the fp32 MFMA (v_mfma_f32_32x32x2_f32) instead of the bf16 pipe with 3-piece split
    # operands: reported next to `value` so that the split form can be judged (both are fp32-faithful)
    alt = None
    if not args.no_alt:
        os.environ["PIDM_CONV_SPLIT"] = "0"
        os.environ["PIDM_WGRAD_SPLIT"] = "0"
        os.environ["PIDM_LAP_SPLIT"] = "0"
        lib.pidm_reload_knobs()                # the library snapshots its knobs once per process
        n_alt = min(args.steps, 20)
        ms_alt = timed(step, n_alt)
        alt = {"value": round(B * world / ms_alt * 1e3, 2), "ms_per_step": round(ms_alt, 3), "steps": n_alt,
               "what": "PIDM_CONV_SPLIT=0 PIDM_WGRAD_SPLIT=0 PIDM_LAP_SPLIT=0: every contraction on the fp32 MFMA"}
        del os.environ["PIDM_CONV_SPLIT"], os.environ["PIDM_WGRAD_SPLIT"], os.environ["PIDM_LAP_SPLIT"]
        lib.pidm_reload_knobs()

    # and with the tracked loss terms returned as python floats every step (the reference's types: one host sync per step)
    eager = None
    if train and not args.no_alt and not args.eager_scalars:
        diffusion.deferred_scalars = False
        n_e = min(args.steps, 20)
        ms_e = timed(step, n_e)
        eager = {"value": round(B * world / ms_e * 1e3, 2), "ms_per_step": round(ms_e, 3), "steps": n_e,
                 "what": "model_estimation_loss returns python floats every step (DenoisingDiffusion.deferred_scalars = False)"}
        diffusion.deferred_scalars = True

    # gradient exchange: what ran and how long one exchange takes (events on the stream the collectives run on)
    exch = None
    if exchange is not None:
        exchange.measure = True
        for _ in range(min(args.steps, 5)):
            step()
        ems = exchange.exchange_ms()
        exchange.measure = False
        exch = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ms_per_exchange": None if ems is None else round(ems, 3),
                "overlapped_with_backward": bool(exchange.last_overlapped), "ranges": [len(r) for r in exchange.ranges],
                "payload_MB": round(sum(hi - lo for rs in exchange.ranges for lo, hi in rs) * 4 / 1e6, 2),
                "forced_single_rank": bool(force_dp and world == 1),
                "collective": exchange.collective, "collective_note": exchange.collective_note}

    roofline = None
    if not args.no_roofline:
        # same steps again with HIP events around every launch of the dominant kernel class (recorded on the
        # launch stream inside the library); kept out of the timed region so the events do not perturb `value`
        nprof = min(args.steps, 5)
        from physicsinformeddiffusionmodels_amd._lib import stream_ptr
        lib.check(lib.pidm_prof_kernels_begin(stream_ptr(dev)), "pidm_prof_kernels_begin")
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        kbuf = C.create_string_buffer(1 << 16)
        if lib.pidm_prof_kernels_collect(kbuf, len(kbuf)) < 0:
            raise SystemExit("pidm_prof_kernels_collect failed")
        # one line per kernel name: name, launches, total ms, declared work (conv FLOPs), class (0 / 1 = forward+dgrad / wgrad on the fp32
        # MFMA, 2 / 3 = the same in split form on the bf16 pipe, -1 = not a convolution)
        kernels = []
        for ln in kbuf.value.decode().splitlines():
            nm, n_, ms_, wk_, cl_ = ln.split("\t")
            kernels.append({"name": nm, "n": int(n_), "ms": float(ms_), "work": float(wk_), "cls": int(cl_)})
        ms, cnt, work = [0.0] * 4, [0] * 4, [0.0] * 4
        for k_ in kernels:
            if 0 <= k_["cls"] < 4:
                ms[k_["cls"]] += k_["ms"]
                cnt[k_["cls"]] += k_["n"]
                work[k_["cls"]] += k_["work"]
        all_ms = sum(k_["ms"] for k_ in kernels)
        conv_ms, conv_fl, conv_n = sum(ms), sum(work), sum(cnt)
        fd_ms, fd_fl = ms[0] + ms[2], work[0] + work[2]
        wg_ms, wg_fl = ms[1] + ms[3], work[1] + work[3]
        sp_ms, sp_fl, sp_n = ms[2] + ms[3], work[2] + work[3], cnt[2] + cnt[3]
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        step_flops = B * flops_per_unit
        traffic, traffic_source = pmc_traffic(wl, B)
        roofline = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
            # the same achieved rate against BOTH matrix peaks: fp32 FLOPs / 157.3 (= frac) and the bf16 terms they are executed as
            # (6 per fp32 product for the split-form launches, 1 for the rest) / 2500
            "frac_fp32_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
            "frac_bf16_peak": round(((6.0 * sp_fl + (conv_fl - sp_fl)) / (conv_ms * 1e-3) / 1e12) / PEAK_BF16_MFMA_TFLOPS, 4) if conv_ms > 0 else None,
            "peak_bf16": PEAK_BF16_MFMA_TFLOPS,
            "kernel": "implicit-GEMM convolutions: fwd, dgrad, wgrad (3x3, 4x4/s2, 7x7 and the compute-bound 1x1: conv3x3_rs_kernel (rows of 32 / 64 "
                      "pixels) / conv3x3_split*_kernel / conv1x1_split_kernel / conv7x7_split_kernel forward + input gradient, "
                      "conv_wgrad_rs*_kernel weight gradient (up to batch 128 the ~60 weight-gradient problems of a pass as three grouped "
                      "launches: conv_wgrad_rs_multi / _rs4_multi / _1x1_multi_kernel) - 6 bf16 "
                      "MFMAs per fp32 product on 3-piece split operands; memory-bound 1x1 layers, their weight gradients and the linears: "
                      "fp32 MFMA)",
            "peak_note": "achieved = algorithmic fp32 FLOPs / HIP-event time; peak = the fp32 MFMA's dense peak, the rate an fp32 "
                         "contraction is priced at; frac_bf16_pipe prices the split-form launches alone against the pipe they run on.  Both "
                         "peaks assume the nominal 2.4 GHz; stamps inside conv3x3_rs_kernel (PIDM_RS_TRACE, profiles/r04_m_conv_rs_clock.txt) "
                         "show a shader clock of 1.5-1.9 GHz while these kernels run and 76-83 % of the matrix pipe busy in cycles",
            # the split-form launches against THEIR pipe: 6 bf16 MFMA terms per fp32 product / their HIP-event time / 2500 TFLOP/s dense
            "frac_bf16_pipe": round(6.0 * sp_fl / (sp_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4) if sp_ms > 0 else None,
            "split_form": {"launches_per_step": sp_n // nprof, "ms_per_step": round(sp_ms / nprof, 3),
                           "fp32_equiv_tflops": round(sp_fl / max(sp_ms, 1e-9) / 1e9, 2),
                           "bf16_tflops": round(6.0 * sp_fl / max(sp_ms, 1e-9) / 1e9, 2), "bf16_peak": PEAK_BF16_MFMA_TFLOPS},
            "fp32_mfma_form": {"launches_per_step": (cnt[0] + cnt[1]) // nprof, "ms_per_step": round((ms[0] + ms[1]) / nprof, 3),
                               "tflops": round((work[0] + work[1]) / max(ms[0] + ms[1], 1e-9) / 1e9, 2)},
            "timing": "one HIP event per library launch (pidm_prof_kernels_begin / _collect) in extra steps after the timed region: a "
                      "launch's time = the interval since the previous launch's event, launch by launch on one stream - graph replay and "
                      "the weight-gradient side-stream overlap are OFF while the hooks are on (a kernel that shares the chip has no "
                      "duration of its own); `value` is measured with both on.  An interval contains the event itself and the launch gap: "
                      "~3 us per launch more than the kernel-only durations of rocprofv3 (profiles/r05_*kernel_stats*.csv) - "
                      "all_kernels_ms_per_step exceeds ms_per_step by that",
            "all_kernels_ms_per_step": round(all_ms / nprof, 3),
            "launches_per_step": conv_n // nprof, "avg_launch_us": round(conv_ms * 1e3 / max(conv_n, 1), 2),
            "kernel_ms_per_step": round(conv_ms / nprof, 3),
            "fwd_dgrad": {"ms_per_step": round(fd_ms / nprof, 3), "tflops": round(fd_fl / max(fd_ms, 1e-9) / 1e9, 2)},
            "wgrad": {"ms_per_step": round(wg_ms / nprof, 3), "tflops": round(wg_fl / max(wg_ms, 1e-9) / 1e9, 2)},
            "step_flop_fraction": round(step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
        }
        if wl == "darcy":
            roofline["step_hbm_fraction"] = round(B * BYTES_PER_SAMPLE / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
        if wl == "darcy":
            roofline["clock_probe"] = rs_clock_probe(lib, dev, B)
        # ---- the step priced against the pipe it runs on ----
        # split form = 6 bf16 MFMA terms per fp32 product, so 2500 / 6 TFLOP/s of fp32-equivalent work is the most this engine's
        # convolutions can do at the nominal clock; the contract FLOPs of the whole step and the split-form launches against that, and
        # both again at the shader clock the probe measured (peaks scale with the clock)
        ceil_sp = PEAK_BF16_MFMA_TFLOPS / 6.0
        step_tf = step_flops / (ms_per_step * 1e-3) / 1e12
        sp_tf = sp_fl / max(sp_ms, 1e-9) / 1e9
        roofline["ceiling_split_form_tflops"] = round(ceil_sp, 1)
        roofline["step_frac_of_split_ceiling"] = round(step_tf / ceil_sp, 4)
        roofline["conv_split_frac_of_split_ceiling"] = round(sp_tf / ceil_sp, 4)
        clk = (roofline.get("clock_probe") or {}).get("shader_clock_ghz")
        if clk:
            scale = clk / 2.4
            roofline["frac_at_measured_clock"] = {
                "shader_clock_ghz": clk,
                "conv_class_vs_fp32_peak": round(achieved / (PEAK_FP32_MFMA_TFLOPS * scale), 4),
                "step_vs_fp32_peak": round(step_tf / (PEAK_FP32_MFMA_TFLOPS * scale), 4),
                "step_vs_split_ceiling": round(step_tf / (ceil_sp * scale), 4),
                "conv_split_vs_split_ceiling": round(sp_tf / (ceil_sp * scale), 4),
                "what": "the same fractions with both peaks scaled by measured / nominal clock (the chip clocks to its power budget; "
                        "the probe reads the clock inside conv3x3_rs_kernel at this batch)"}
        top = []
        for k_ in kernels[:5]:
            e = {"name": k_["name"], "launches_per_step": round(k_["n"] / nprof, 1), "us_per_step": round(k_["ms"] / nprof * 1e3, 1),
                 "avg_us": round(k_["ms"] / max(k_["n"], 1) * 1e3, 2), "share_of_kernel_time": round(k_["ms"] / max(all_ms, 1e-9), 4)}
            if k_["work"] > 0:
                tf = k_["work"] / (k_["ms"] * 1e-3) / 1e12
                e["fp32_equiv_tflops"] = round(tf, 1)
                e["frac_fp32_peak"] = round(tf / PEAK_FP32_MFMA_TFLOPS, 3)
                if k_["cls"] in (2, 3):
                    e["frac_bf16_pipe"] = round(6.0 * tf / PEAK_BF16_MFMA_TFLOPS, 3)
            else:
                e["fp32_equiv_tflops"] = None     # no FLOP contract declared for this kernel (attention / normalisation / reductions)
            top.append(e)
        roofline["top_kernels"] = top
        roofline["kernel_classes_ms_per_step"] = {
            "conv": round(conv_ms / nprof, 3),
            "attention": round(sum(k_["ms"] for k_ in kernels if k_["name"].startswith(("lap_", "la_", "mid_attn"))) / nprof, 3),
            "groupnorm": round(sum(k_["ms"] for k_ in kernels if k_["name"].startswith("gn_")) / nprof, 3),
            "layernorm": round(sum(k_["ms"] for k_ in kernels if k_["name"].startswith("layernorm")) / nprof, 3),
            "reductions_and_pack": round(sum(k_["ms"] for k_ in kernels if k_["name"].startswith(("reduce_multi", "pack_", "wgrad_reduce", "colsum"))) / nprof, 3),
        }
        st_bytes = pmc_step_traffic(wl, B)
        if st_bytes is not None:
            # every kernel class of the step from the same PMC pass; for darcy next to the SURVEY 8(d) contract bytes
            roofline["step_traffic_bytes"] = round(st_bytes, 0)
            roofline["step_traffic_tbs"] = round(st_bytes / (ms_per_step * 1e-3) / 1e12, 3)
            if wl == "darcy":
                roofline["step_traffic_over_contract"] = round(st_bytes / (B * BYTES_PER_SAMPLE), 3)

    # ---- the reference's loop body, unchanged (main.py:157-183,316): what a user who only swaps the import path gets ----
    dropin = None
    if train and world == 1 and not args.no_alt:
        from physicsinformeddiffusionmodels_amd.denoising_utils import EMA as _EMA
        d_opt = torch.optim.Adam(model.parameters(), lr=n_lr)                     # main.py:134
        d_ema = _EMA(0.99)                                                        # main.py:53,131
        d_ema.register(model)
        diffusion.deferred_scalars = False                                        # python floats, as the reference returns them
        it_d = {"i": 0}

        def dropin_step():
            model.train()
            loss, data_loss, residual_loss, ineq_loss, opt_loss = diffusion.model_estimation_loss(batch, residual_func=residuals, **loss_kw)
            d_opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.)
            d_opt.step()
            if it_d["i"] % LOG_FREQ == 0:
                _ = f"training loss: {loss.item():.3e}"                           # main.py:168
            d_ema.update(model)                                                   # main.py:178-179 (iteration > ema_start: steady state)
            model.eval()
            d_ema.ema(residuals.model)                                            # main.py:183
            d_ema.restore(residuals.model)                                        # main.py:316
            it_d["i"] += 1
        n_d = min(args.steps, 20)
        ms_d = timed(dropin_step, n_d)
        dropin = {"value": round(B / ms_d * 1e3, 2), "ms_per_step": round(ms_d, 3), "steps": n_d,
                  "what": "main.py:157-183,316 loop body unchanged: python-float loss terms (host sync per step), torch clip_grad_norm_ + "
                          "torch.optim.Adam, ema.update + ema.ema + ema.restore every iteration"}
        diffusion.deferred_scalars = not args.eager_scalars
        model.train()
        del d_opt, d_ema

    # ---- north-star configuration: per-GPU batch 256 (BASELINE.json north_star), same step as `value` ----
    b256 = None
    if wl == "darcy" and world == 1 and B != 256 and not args.no_alt:
        batch256 = synthetic_darcy_batch(256, 64, seed=300 + rank, device=dev)
        n_b = min(args.steps, 10)
        ms_b = timed(lambda: step(batch256), n_b)
        b256 = {"value": round(256 / ms_b * 1e3, 2), "ms_per_step": round(ms_b, 3), "steps": n_b, "per_gpu_batch": 256,
                "step_flop_fraction": round(256 * FLOPS_PER_SAMPLE_FWD_BWD / (ms_b * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                "step_hbm_fraction": round(256 * BYTES_PER_SAMPLE / (ms_b * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                "what": "the headline step (loss + backward + fused clip+Adam, deferred loss scalars) at per-GPU batch 256; fractions = "
                        "SURVEY 8(d) contract FLOPs / bytes per sample over the measured step time against 157.3 TFLOP/s and 8 TB/s"}
        st256 = pmc_step_traffic("darcy", 256)
        if st256 is not None:
            b256["step_traffic_bytes"] = round(st256, 0)
            b256["step_traffic_over_contract"] = round(st256 / (256 * BYTES_PER_SAMPLE), 3)
            b256["step_traffic_tbs"] = round(st256 / (ms_b * 1e-3) / 1e12, 3)
        del batch256

    resonly = None
    if wl == "darcy" and world == 1 and rank == 0 and not args.no_alt:
        resonly = residual_only_rates(lib, residuals, diffusion, dev)

    # ---- BASELINE configs[3] (per-GPU share) and configs[4] in front of the driver: the same script, child processes ----
    legs = {}
    if wl == "darcy" and world == 1 and rank == 0 and not args.no_alt and os.environ.get("PIDM_BENCH_CHILD") != "1":
        legs["mechanics_b32"] = child_leg("mechanics", min(args.steps, 10), 3)
        legs["sampling_b1024"] = child_leg("sampling", min(args.steps, 10), 3)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = max(1, min(32, os.cpu_count() or 1))
        cpu_b, cpu_steps = {"darcy": (64, 2), "mechanics": (2, 1), "sampling": (32, 3)}[wl]
        cpu = cpu_baseline(wl, cpu_b, cpu_steps, threads)

    if rank == 0:
        if wl == "darcy":
            metric, unit = "training samples/sec (UNet+PDE-residual step), 64x64 Darcy", "samples/s"
            workload = ("Darcy 64x64 2-ch (K,p), PIDM loss on (c_residual=1e-3), Unet3D dim=32, 100 diffusion steps, "
                        "loss+backward+clip+Adam (main.py:157-166)")
        elif wl == "mechanics":
            metric, unit = "training samples/sec (UNet+PDE-residual step), 64x64 topology optimisation (mechanics)", "samples/s"
            workload = ("topology optimisation 64x64 elements (65x65 nodes), 10-ch input / 3-ch output, Unet3D dim=128 with sigmoid "
                        "density head, matrix-free K.u residual + volume + compliance terms, loss+backward+clip+Adam "
                        "(main.py:102-109,126,157-166)")
        else:
            metric, unit = "sample-steps/sec (DDPM ancestral sampling: UNet forward + residual + update per step), 64x64 Darcy", "sample-steps/s"
            workload = ("sample.py DDPM sampling, Darcy 64x64, 1000-step schedule, Unet3D dim=32; a step = one p_sample step of the "
                        "whole batch (a full chain = 1000 steps; sample.py:145-150)")
        cfg = {"workload": workload, "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
               "arithmetic": "fp32 tensors and accumulation; 3x3 / 4x4-stride-2 / 7x7 convolution contractions (fwd, dgrad, wgrad), the "
                             "compute-bound 1x1 convolutions (fwd, dgrad) and the pixel sums / projections of the projected attention "
                             "as 6 bf16 MFMA terms on "
                             "round-to-nearest 3-piece splits of both operands (24 mantissa bits; error <= the fp32 MFMA's own, "
                             "profiles/r02_bf16_split_probe.txt), everything else fp32 MFMA / VALU"}
        if train:
            cfg["optimizer"] = "torch clip_grad_norm_+Adam" if args.torch_optimizer else "fused flat clip+Adam (k_optim.hip)"
            cfg["ema_in_step"] = bool(args.ema)
            cfg["loss_scalars"] = ("python floats every step (host sync per step)" if args.eager_scalars else
                                   "computed and copied to the host every step, read every 20 steps as main.py:167-175 does "
                                   "(DenoisingDiffusion.deferred_scalars)")
        else:
            cfg["seconds_per_1000_step_chain"] = round(ms_per_step, 2)
        out = {
            "metric": metric, "value": round(value, 2), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": cfg, "roofline": roofline, "cpu_baseline": cpu,
            "fp32_mfma_only": alt, "eager_scalars": eager, "dropin_main_py": dropin, "north_star_b256": b256,
            "residual_only": resonly, "exchange": exch, "per_rank": per_rank, "launches": launches,
            # contract FLOPs of the step (SURVEY 8(d) / FlopCounterMode on the reference) over the measured step time against the
            # fp32 MFMA peak - also present when the roofline legs are skipped
            "step_flop_fraction": round(B * flops_per_unit / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
        }
        out.update(legs)
        if selftest:
            out["selftest"] = ("NOT A MEASUREMENT: bench.main(test_env=...) ran this script's control flow on CPU tensors over gloo with "
                               f"the library the test handed in ({lib.backend}), {P_img}x{P_img} fields, Unet3D dim={dim_darcy}")
            out["data"] = "synthetic (selftest)"
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its banner through C stdio (block-buffered when stdout is a pipe): flush it first so that the JSON line is
        # the last line of the output
        try:
            C.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
