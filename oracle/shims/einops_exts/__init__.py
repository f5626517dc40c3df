"""Stand-in for einops_exts (not installed here). No arithmetic."""
from einops import rearrange


def rearrange_many(tensors, pattern, **kwargs):
    return tuple(rearrange(t, pattern, **kwargs) for t in tensors)
