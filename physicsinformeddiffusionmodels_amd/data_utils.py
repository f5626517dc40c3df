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


def synthetic_darcy_batch(batch, pixels=64, seed=0, device='cpu'):
    """Synthetic [B,2,P,P] Darcy-shaped fields: ch0 p = 0.1*randn, ch1 K = box-smoothed exp(0.5*randn)."""
    g = torch.Generator().manual_seed(seed)
    p = 0.1 * torch.randn(batch, 1, pixels, pixels, generator=g)
    k = torch.exp(0.5 * torch.randn(batch, 1, pixels, pixels, generator=g))
    k = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(k, (2, 2, 2, 2), mode='replicate'), 5, stride=1)
    return torch.cat([p, k], dim=1).to(device)
