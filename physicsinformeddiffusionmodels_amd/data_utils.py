"""Dataset readers with the reference's tensor layouts (reference: src/data_utils.py:26-119), plus the
synthetic generators bench.py / the tests use (the real datasets are an external download, README.md:31-44).
IO only - nothing here is on the accelerated path."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import pandas as pd
import torch
from torch.utils import data

from .unet_model import cycle, generalized_b_xy_c_to_image, generalized_image_to_b_xy_c  # noqa: F401


class Dataset(data.Dataset):
    """CSV files (one per channel, rows = samples, columns = P*P pixels) -> [N, C, P, P] tensor in RAM."""

    def __init__(self, data_directories, use_double=False, return_img=True, gaussian_prior=False):
        super().__init__()
        self.data_paths = list(data_directories)
        arrs = [pd.read_csv(p, header=None).to_numpy() for p in self.data_paths]
        stacked = np.stack(arrs, axis=-1)                         # [N, P*P, C]
        dtype = torch.float64 if use_double else torch.float32
        self.data = torch.tensor(stacked, dtype=dtype)
        self.num_datapoints = len(self.data)
        if return_img:
            assert self.data.dim() == 3, 'Data must be of shape (num_datapoints, pixels_x*pixels_y, channels)'
            self.data = generalized_b_xy_c_to_image(self.data).contiguous()
        if gaussian_prior:
            self.data = torch.randn_like(self.data)

    def normalize(self, arr, min_val, max_val):
        return (arr - min_val) / (max_val - min_val)

    def unnorm(self, arr, min_val, max_val):
        return arr * (max_val - min_val) + min_val

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        if index >= self.num_datapoints:
            raise IndexError('index out of range')
        return self.data[index]


class Dataset_Paths(data.Dataset):
    """One `.npy` [65,65,10] per sample -> [10,65,65] (vf, strain energy, von Mises, u_x, u_y, E, BC_x, BC_y, load_x, load_y)."""

    def __init__(self, data_directories, use_double=False, return_img=True, gaussian_prior=False, exts=['npy']):
        super().__init__()
        self.paths = [p for ext in exts for p in Path(f'{data_directories}').glob(f'**/*.{ext}')]
        self.paths = sorted(self.paths, key=lambda x: int(x.name.split('.')[0]))
        self.num_datapoints = len(self.paths)
        self.dtype = torch.float64 if use_double else torch.float32
        self.return_img = return_img
        self.gaussian_prior = gaussian_prior

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        if index >= self.num_datapoints:
            raise IndexError('index out of range')
        arr = np.load(self.paths[index], allow_pickle=True, encoding='latin1')
        return torch.tensor(arr.transpose(2, 0, 1), dtype=self.dtype)


class DevicePrefetcher:
    """Iterator over a (cycled) loader that keeps `depth` batches in flight to the GPU: each batch is staged in a reusable
    PINNED host buffer and copied on a side stream while the previous step computes, so `next(dl).to(device)` of
    main.py:158 (a pageable, synchronous copy per iteration) leaves the critical path.  Yields device tensors that are safe
    to use on the current stream; the tensor handed out is only recycled `depth` batches later.  With `rank` / `world` the
    loader's batches are sharded contiguously (data parallel: every rank iterates the same loader, keeps its slice).
    On a CPU device it degrades to a plain pass-through (the tests)."""

    def __init__(self, loader, device, depth=2, rank=0, world=1):
        self.it = iter(loader)
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self.rank, self.world = rank, world
        self.cuda = self.device.type == 'cuda'
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.queue = []
        self.pinned = {}
        self.slot = 0
        for _ in range(self.depth):
            self._enqueue()

    def _shard(self, batch):
        if self.world == 1:
            return batch
        n = batch.shape[0] // self.world
        if n == 0:
            raise ValueError('batch smaller than the number of ranks')
        return batch[self.rank * n:(self.rank + 1) * n]

    def _enqueue(self):
        try:
            batch = next(self.it)
        except StopIteration:
            return
        batch = self._shard(batch)
        if not self.cuda:
            self.queue.append((batch.to(self.device), None))
            return
        key = (self.slot % (self.depth + 1), tuple(batch.shape), batch.dtype)
        self.slot += 1
        host, busy = self.pinned.get(key, (None, None))
        if host is None:
            host = torch.empty(batch.shape, dtype=batch.dtype, pin_memory=True)
        elif busy is not None:
            busy.synchronize()           # the copy that last read this staging buffer (depth + 1 batches ago) has drained
        host.copy_(batch)
        with torch.cuda.stream(self.stream):
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pinned[key] = (host, ev)
        self.queue.append((dev, ev))

    def __iter__(self):
        return self

    def __next__(self):
        if not self.queue:
            raise StopIteration
        dev, ev = self.queue.pop(0)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            dev.record_stream(torch.cuda.current_stream(self.device))
        self._enqueue()
        return dev


def synthetic_mechanics_batch(batch, seed=0, device='cpu'):
    """Synthetic [B,10,65,65] in the dataset's channel order (main.py:102-109): vf, strain energy, von Mises | u_x, u_y, E
    (64x64, zero padded) | bc_x, bc_y, load_x, load_y - left edge clamped, one point load on the right edge."""
    g = torch.Generator().manual_seed(seed)
    inp = torch.zeros(batch, 10, 65, 65)
    inp[:, 0] = (0.2 + 0.3 * torch.rand(batch, generator=g)).view(batch, 1, 1)
    inp[:, 1:3] = torch.randn(batch, 2, 65, 65, generator=g)
    inp[:, 3:5] = 0.1 * torch.randn(batch, 2, 65, 65, generator=g)
    inp[:, 5, :64, :64] = torch.rand(batch, 64, 64, generator=g)
    inp[:, 6:8, :, 0] = 1.0
    inp[:, 9, 32, 64] = -1.0
    return inp.to(device)


def synthetic_darcy_batch(batch, pixels=64, seed=0, device='cpu'):
    """Synthetic [B,2,P,P] Darcy-shaped fields: ch0 p = 0.1*randn, ch1 K = box-smoothed exp(0.5*randn)."""
    g = torch.Generator().manual_seed(seed)
    p = 0.1 * torch.randn(batch, 1, pixels, pixels, generator=g)
    k = torch.exp(0.5 * torch.randn(batch, 1, pixels, pixels, generator=g))
    k = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(k, (2, 2, 2, 2), mode='replicate'), 5, stride=1)
    return torch.cat([p, k], dim=1).to(device)
