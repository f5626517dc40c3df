"""main_dp.py - data-parallel training driver for the gfx950 engine (one process per MI355X, RCCL over xGMI).

The reference trains in a single process (main.py:150-316); BASELINE configs[2]/[3] shard the batch over the 8 GPUs of a
node.  This driver keeps main.py's step (loss -> zero_grad -> backward -> clip -> Adam -> EMA update / swap-in / restore,
checkpoints with the EMA weights) and adds what data parallelism needs: per-rank batch shards fed by a pinned-memory
prefetcher, the overlapped three-bucket gradient all-reduce (parallel.GradientExchange), the fused clip + Adam + EMA
kernel on flat buffers, rank-0 logging and checkpoints.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 main_dp.py \
        --gov-eqs darcy --global-batch 512 --iterations 300000
    python main_dp.py --gov-eqs mechanics --global-batch 32 --iterations 20 --synthetic          # single GPU

Reads model.yaml (same keys as main.py:20-45) when present; datasets from ./data/... as main.py:66-113 expects, or
--synthetic fields of the same shape.  Evaluation sampling / plotting of main.py:200-313 is not part of this driver
(use sample.py with the written checkpoint)."""
from __future__ import annotations

import argparse
import os
import sys
import time
from pathlib import Path

# multi-process GPU work on this stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle: invalid argument otherwise); the
# variable must be in place before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

DEFAULTS = dict(x0_estimation='mean', ddim_steps=0, residual_grad_guidance=False, gov_eqs='darcy', fd_acc=2, c_data=1.,
                c_residual=1.e-3, c_ineq=0., lambda_opt=0., diff_steps=100)


def parse():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--config', default='model.yaml')
    ap.add_argument('--gov-eqs', choices=('darcy', 'mechanics'), default=None)
    ap.add_argument('--global-batch', type=int, default=None, help='default: 64 per rank (darcy), 32 per rank (mechanics)')
    ap.add_argument('--iterations', type=int, default=None)
    ap.add_argument('--lr', type=float, default=1.e-4)
    ap.add_argument('--ema-start', type=int, default=1000)
    ap.add_argument('--save-freq', type=int, default=20000)
    ap.add_argument('--log-freq', type=int, default=20)
    ap.add_argument('--name', default='run_dp')
    ap.add_argument('--synthetic', action='store_true', help='synthetic fields of the dataset shape (no ./data needed)')
    ap.add_argument('--seed', type=int, default=42)
    return ap.parse_args()


def main():
    args = parse()
    import torch.distributed as dist
    from torch.utils.data import DataLoader, TensorDataset
    from physicsinformeddiffusionmodels_amd.data_utils import (Dataset, Dataset_Paths, DevicePrefetcher, cycle,
                                                               synthetic_darcy_batch, synthetic_mechanics_batch)
    from physicsinformeddiffusionmodels_amd.denoising_utils import EMA, DenoisingDiffusion, fix_seeds, save_model
    from physicsinformeddiffusionmodels_amd.optim import FusedClipAdam
    from physicsinformeddiffusionmodels_amd.parallel import GradientExchange
    from physicsinformeddiffusionmodels_amd.residuals_darcy import ResidualsDarcy
    from physicsinformeddiffusionmodels_amd.residuals_mechanics_K import ResidualsMechanics
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('main_dp.py needs MI355X devices (the engine has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', device_id=device)      # "nccl" is RCCL on ROCm

    config = dict(DEFAULTS)
    if Path(args.config).exists():
        config.update(yaml.safe_load(Path(args.config).read_text()))
    if args.gov_eqs:
        config['gov_eqs'] = args.gov_eqs
    gov_eqs = config['gov_eqs']
    use_ddim_x0 = config['x0_estimation'] == 'sample'
    per_rank = {'darcy': 64, 'mechanics': 32}[gov_eqs]
    global_batch = args.global_batch or per_rank * world
    if global_batch % world:
        raise SystemExit('--global-batch must be divisible by the number of ranks')
    iterations = args.iterations if args.iterations is not None else {'darcy': 300000, 'mechanics': 600000}[gov_eqs]

    fix_seeds(args.seed)                      # identical initial weights on every rank
    diffusion = DenoisingDiffusion(config['diff_steps'], device, config['residual_grad_guidance'])
    if gov_eqs == 'darcy':
        model = Unet3D(dim=32, channels=2).to(device)
        residuals = ResidualsDarcy(model=model, fd_acc=config['fd_acc'], pixels_per_dim=64, pixels_at_boundary=True, reverse_d1=True,
                                   device=device, bcs='none', domain_length=1., residual_grad_guidance=config['residual_grad_guidance'],
                                   use_ddim_x0=use_ddim_x0, ddim_steps=config['ddim_steps'])
        ds = (TensorDataset(synthetic_darcy_batch(max(4 * global_batch, 1024), 64, seed=1)) if args.synthetic else
              Dataset(('./data/darcy/train/p_data.csv', './data/darcy/train/K_data.csv')))
    else:
        model = Unet3D(dim=128, channels=10, out_dim=3, sigmoid_last_channel=True).to(device)
        residuals = ResidualsMechanics(model=model, pixels_per_dim=64, pixels_at_boundary=True, device=device, bcs='none',
                                       no_BC_folder='./data/mechanics/solidspy_k_no_BC/', topopt_eval=False,
                                       use_ddim_x0=use_ddim_x0, ddim_steps=config['ddim_steps'])
        ds = (TensorDataset(synthetic_mechanics_batch(max(4 * global_batch, 256), seed=1)) if args.synthetic else
              Dataset_Paths('./data/mechanics/train/fields/'))
    # every rank walks the same loader in the same order and keeps its contiguous slice of each global batch
    loader = DataLoader(ds, batch_size=global_batch, shuffle=False, drop_last=True, num_workers=0 if args.synthetic else 4)
    batches = (b[0] if isinstance(b, (list, tuple)) else b for b in cycle(loader))
    feed = DevicePrefetcher(batches, device, depth=2, rank=rank, world=world)

    ema = EMA(0.99)
    ema.register(model)
    optimizer = FusedClipAdam(model, lr=args.lr, max_norm=1., image_size=64, ema=ema, ema_start=args.ema_start)
    exchange = GradientExchange(model, world, diffusion=diffusion)
    diffusion.deferred_scalars = True       # the tracked loss terms synchronise when they are printed (every log_freq iterations)
    torch.manual_seed(args.seed + 1000 * (rank + 1))        # per-rank (t, eps) streams
    out_dir = f'./trained_models/{args.name}'
    loss_kw = dict(c_data=config['c_data'], c_residual=config['c_residual'], c_ineq=config['c_ineq'], lambda_opt=config['lambda_opt'])
    if rank == 0:
        n_par = sum(p.numel() for p in model.parameters() if p.requires_grad)
        print(f'{gov_eqs}: {n_par} trainable parameters, global batch {global_batch} over {world} rank(s)', flush=True)

    t0 = time.perf_counter()
    for iteration in range(iterations + 1):
        cur_batch = next(feed)
        loss, data_loss, residual_loss, ineq_loss, opt_loss = diffusion.model_estimation_loss(cur_batch, residual_func=residuals, **loss_kw)
        optimizer.zero_grad()
        loss.backward()
        exchange.allreduce()
        optimizer.step()                      # clip_grad_norm_(1.) + Adam (+ EMA once iteration > ema_start), one pass
        if iteration > args.ema_start:
            ema.update(model)                 # acknowledged: the update already ran inside the optimizer kernel
        if rank == 0 and iteration % args.log_freq == 0:
            dt = time.perf_counter() - t0
            print(f'it {iteration}: loss {loss.item():.3e} data {data_loss:.3e} |r| {residual_loss:.3e} '
                  f'[{(iteration + 1) * global_batch / max(dt, 1e-9):.0f} samples/s]', flush=True)
        if rank == 0 and iteration > 0 and iteration % args.save_freq == 0:
            ema.ema(residuals.model)          # checkpoints hold the averaged weights (main.py:183,314-316): a pointer flip
            save_model(config, model, iteration, out_dir)
            ema.restore(residuals.model)
    if rank == 0 and iterations > 0:
        ema.ema(residuals.model)
        save_model(config, model, iterations, out_dir)
        ema.restore(residuals.model)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
