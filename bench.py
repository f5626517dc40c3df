"""bench.py - headline benchmark: training samples/sec of the UNet + PDE-residual step, Darcy 64x64.

A step is the reference's training-loop body (main.py:157-166): model_estimation_loss (q-sample, UNet forward,
Darcy residual, PIDM loss) -> zero_grad -> backward -> [gradient all-reduce when N>1] -> clip_grad_norm_(1.0) ->
Adam.step, on synthetic Darcy-shaped fields already resident in HBM, fp32, default torch init under seed 0.
Workload = BASELINE.json configs[1] (batch 64 per GPU; configs[2] = 8 x 64 under weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     - the dominant kernel class (implicit-GEMM conv fwd/dgrad + wgrad on the fp32 matrix cores):
                 algorithmic FLOPs / HIP-event time of those launches, measured live on the launch stream
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

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

FLOPS_PER_SAMPLE_FWD_BWD = 11.916e9   # SURVEY 8(d), FlopCounterMode on the reference, Darcy dim=32 64x64
BYTES_PER_SAMPLE = 148.5e6            # SURVEY 8(d) compulsory-traffic contract at per-GPU batch 256
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (configs[1] = 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--torch-optimizer", action="store_true",
                    help="use torch clip_grad_norm_ + torch.optim.Adam instead of the fused flat clip+Adam kernel (same math)")
    ap.add_argument("--calib-copy", action="store_true",
                    help="also run one 1 GiB device copy (known byte count for the PMC traffic passes, tools/pmc_traffic.sh)")
    return ap.parse_args()


def cpu_baseline(batch=8, steps=2):
    """Time the oracle's full training step (fwd + autograd bwd + clip + Adam) on the host cores."""
    from oracle import pidm_oracle as O
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D
    from physicsinformeddiffusionmodels_amd.data_utils import synthetic_darcy_batch
    torch.manual_seed(0)
    m = Unet3D(dim=32, channels=2)
    trainable = {k for k, v in m.named_parameters() if v.requires_grad}
    p = {k: v.detach().clone().requires_grad_(k in trainable) for k, v in m.state_dict().items()}
    used = None
    cfg = O.UnetCfg(dim=32, channels=2)
    tables = O.diffusion_tables(100)
    x0 = synthetic_darcy_batch(batch, 64, seed=1)
    g = torch.Generator().manual_seed(2)
    t = torch.randint(0, 100, (batch,), generator=g)
    eps = torch.randn(batch, 2, 64, 64, generator=g)
    opt = None
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        loss, _, _, _ = O.darcy_training_loss(p, cfg, tables, x0, t, eps, 1.0, 1e-3)
        for v in p.values():
            v.grad = None
        loss.backward()
        if used is None:
            used = [v for v in p.values() if v.grad is not None]
            opt = torch.optim.Adam(used, lr=1e-4)
        torch.nn.utils.clip_grad_norm_(used, 1.0)
        opt.step()
        if it > 0:
            times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return {"value": round(batch / dt, 3), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed steps (1 warm-up) of the oracle's Darcy 64x64 training step (UNet dim=32 fwd+bwd, "
                      f"residual, loss, clip, Adam) at batch {batch}, {dt:.2f} s/step"}


def pmc_traffic(batch):
    """HBM bytes per launch of the conv class from the committed PMC passes (tools/pmc_traffic.sh -> profiles/): the
    counters need rocprofv3 around the process, so they cannot be read live; null when no pass exists for this batch."""
    import json as _json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"pmc_traffic_b{batch}.json")
    try:
        with open(path) as f:
            return round(float(_json.load(f)["conv"]["hbm_bytes_per_launch"]), 0)
    except (OSError, KeyError, ValueError):
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    # PIDM_BENCH_SHARE_GPU=1 (debug only): all ranks share cuda:0 and talk over gloo - exercises the N>1 code path on
    # a single-GPU box; real runs use one GPU per rank over RCCL (backend "nccl").
    share = os.environ.get("PIDM_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from physicsinformeddiffusionmodels_amd._lib import get_lib
    from physicsinformeddiffusionmodels_amd.data_utils import synthetic_darcy_batch
    from physicsinformeddiffusionmodels_amd.denoising_utils import DenoisingDiffusion
    from physicsinformeddiffusionmodels_amd.parallel import allreduce_gradients
    from physicsinformeddiffusionmodels_amd.residuals_darcy import ResidualsDarcy
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D

    lib = get_lib()
    B = args.batch
    torch.manual_seed(0)                      # identical initial weights on every rank
    model = Unet3D(dim=32, channels=2).to(dev)
    diffusion = DenoisingDiffusion(100, dev)
    residuals = ResidualsDarcy(model=model, fd_acc=2, pixels_per_dim=64, pixels_at_boundary=True, reverse_d1=True,
                               device=dev, bcs='none', domain_length=1.)
    if args.torch_optimizer:
        optimizer = torch.optim.Adam(model.parameters(), lr=1.e-4)
    else:
        from physicsinformeddiffusionmodels_amd.optim import FusedClipAdam
        optimizer = FusedClipAdam(model, lr=1.e-4, max_norm=1., image_size=64)   # clip_grad_norm_(1.) + Adam, 2 launches
    batch = synthetic_darcy_batch(B, 64, seed=100 + rank, device=dev)   # resident in HBM; each rank its own shard
    torch.manual_seed(1234 + rank)

    def step():
        loss, data_loss, residual_loss, _, _ = diffusion.model_estimation_loss(
            batch, residual_func=residuals, c_data=1., c_residual=1e-3, c_ineq=0., lambda_opt=0.)
        optimizer.zero_grad()
        loss.backward()
        if world > 1:
            allreduce_gradients(model, world)
        if args.torch_optimizer:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.)
        optimizer.step()
        return loss

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.calib_copy:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        dst.copy_(src)
        del src, dst
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed

    roofline = None
    if not args.no_roofline:
        # same steps again with HIP events around every launch of the dominant kernel class (recorded on the
        # launch stream inside the library); kept out of the timed region so the events do not perturb `value`
        nprof = min(args.steps, 5)
        lib.pidm_prof_enable(1)
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        lib.pidm_prof_enable(0)
        ms = (C.c_double * 4)()
        cnt = (C.c_longlong * 4)()
        work = (C.c_double * 4)()
        lib.pidm_prof_collect(ms, cnt, work)
        conv_ms, conv_fl, conv_n = ms[0] + ms[1], work[0] + work[1], cnt[0] + cnt[1]
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        step_flops = B * FLOPS_PER_SAMPLE_FWD_BWD
        roofline = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": pmc_traffic(B),
            "kernel": "conv_igemm_kernel + conv_wgrad_kernel (fp32 MFMA implicit GEMM: fwd, dgrad, wgrad)",
            "timing": "HIP events per launch in extra steps after the timed region; the library keeps the weight-gradient "
                      "side-stream overlap OFF while these hooks are on (a kernel that shares the chip has no duration of its own); "
                      "`value` is measured with the overlap on",
            "launches_per_step": conv_n // nprof, "avg_launch_us": round(conv_ms * 1e3 / max(conv_n, 1), 2),
            "kernel_ms_per_step": round(conv_ms / nprof, 3),
            "fwd_dgrad": {"ms_per_step": round(ms[0] / nprof, 3), "tflops": round(work[0] / max(ms[0], 1e-9) / 1e9, 2)},
            "wgrad": {"ms_per_step": round(ms[1] / nprof, 3), "tflops": round(work[1] / max(ms[1], 1e-9) / 1e9, 2)},
            "step_flop_fraction": round(step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            "step_hbm_fraction": round(B * BYTES_PER_SAMPLE / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
        }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        out = {
            "metric": "training samples/sec (UNet+PDE-residual step), 64x64 Darcy", "value": round(value, 2),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Darcy 64x64 2-ch (K,p), PIDM loss on (c_residual=1e-3), Unet3D dim=32, 100 diffusion "
                                   "steps, loss+backward+clip+Adam (main.py:157-166)", "per_gpu_batch": B,
                       "optimizer": "torch clip_grad_norm_+Adam" if args.torch_optimizer else "fused flat clip+Adam (k_optim.hip)",
                       "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
