"""Stand-in for rotary_embedding_torch (not installed here).

Only constructed by the reference (src/unet_model.py:439); the temporal attention that would call
rotate_queries_or_keys is never executed on the image path.
"""
import torch
from torch import nn


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        raise NotImplementedError("temporal attention is never called on the image path")
