"""Host-side mirror of the reference denoiser `Unet3D` (reference: src/unet_model.py:406-623).

The class keeps the reference's constructor signature, its 317-entry `state_dict` (same names, same
shapes - Conv3d weights stay `[Cout,Cin,1,k,k]`) and its `forward(x, time, ...)` contract, so
`main.py` / `sample.py` and pretrained checkpoints stay drop-in.  It is a *parameter container*:
`forward` does not run PyTorch layers, it hands raw device pointers to the hand-written gfx950 engine
(`csrc/`, C-ABI in `include/pidm.h`) through `_engine.UnetFunction`.  There is no CPU / eager
fallback: if the HIP library is missing or the tensors are not on an MI355X, `forward` raises.
"""
from __future__ import annotations

import math

import torch
from torch import nn


def prob_mask_like(shape, prob, device):
    """Classifier-free dropout mask, True = drop (src/unet_model.py:63-69; same RNG consumption)."""
    if prob == 1:
        return torch.ones(shape, device=device, dtype=torch.bool)
    elif prob == 0:
        return torch.zeros(shape, device=device, dtype=torch.bool)
    return torch.zeros(shape, device=device).float().uniform_(0, 1) < prob


def exists(x):
    return x is not None


def noop(*args, **kwargs):
    pass


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def cycle(dl):
    while True:
        for data in dl:
            yield data


def generalized_image_to_b_xy_c(tensor):
    """[B, c..., X, Y] -> [B, X*Y, c...]  (reference: src/unet_model.py:12-18)."""
    nd = tensor.dim()
    perm = [0, nd - 2, nd - 1] + list(range(1, nd - 2))
    t = tensor.permute(*perm)
    return t.reshape(t.shape[0], t.shape[1] * t.shape[2], *t.shape[3:])


def generalized_b_xy_c_to_image(tensor, pixels_x=None, pixels_y=None):
    """[B, X*Y, c...] -> [B, c..., X, Y]  (reference: src/unet_model.py:20-28)."""
    if pixels_x is None or pixels_y is None:
        pixels_x = pixels_y = int(math.sqrt(tensor.shape[1]))
    t = tensor.reshape(tensor.shape[0], pixels_x, pixels_y, *tensor.shape[2:])
    nd = t.dim()
    perm = [0] + list(range(3, nd)) + [1, 2]
    return t.permute(*perm)


# --------------------------------------------------------------------------------------------------
# parameter containers (names chosen so that state_dict() keys equal the reference's)
# --------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    """A module that only owns parameters/sub-modules; compute happens in the HIP engine."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: compute runs in the gfx950 engine via Unet3D.forward")


class _RotaryFreqs(_Holder):
    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)


class _RelPosBias(_Holder):
    def __init__(self, heads, num_buckets=32):
        super().__init__()
        self.relative_attention_bias = nn.Embedding(num_buckets, heads)


class _ChanLayerNorm(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(1, dim, 1, 1, 1))


class _Wrap(_Holder):
    """Residual / EinopsToAndFrom: a single child called `fn`."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class _PreNorm(_Holder):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = _ChanLayerNorm(dim)


class _Block(_Holder):
    def __init__(self, dim, dim_out, groups):
        super().__init__()
        self.proj = nn.Conv3d(dim, dim_out, (1, 3, 3), padding=(0, 1, 1))
        self.norm = nn.GroupNorm(groups, dim_out)
        self.act = nn.SiLU()


class _ResnetBlock(_Holder):
    def __init__(self, dim, dim_out, time_emb_dim=None, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, dim_out * 2)) if exists(time_emb_dim) else None
        self.block1 = _Block(dim, dim_out, groups)
        self.block2 = _Block(dim_out, dim_out, groups)
        self.res_conv = nn.Conv3d(dim, dim_out, 1) if dim != dim_out else nn.Identity()


class _SpatialLinearAttention(_Holder):
    def __init__(self, dim, heads, dim_head=32, cond_dim=64):
        super().__init__()
        hidden = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_q = nn.Conv2d(dim, hidden, 1, bias=False)          # unused by forward (kept for state_dict)
        self.to_k = nn.Linear(cond_dim, hidden, bias=False)        # unused
        self.to_v = nn.Linear(cond_dim, hidden, bias=False)        # unused
        self.to_out = nn.Conv2d(hidden, dim, 1)


class _Attention(_Holder):
    def __init__(self, dim, heads, dim_head=32, rotary_emb=None, cond_dim=64):
        super().__init__()
        hidden = dim_head * heads
        self.rotary_emb = rotary_emb
        self.to_qkv = nn.Linear(dim, hidden * 3, bias=False)
        self.to_q = nn.Linear(dim, hidden, bias=False)             # unused
        self.to_k = nn.Linear(cond_dim, hidden, bias=False)        # unused
        self.to_v = nn.Linear(cond_dim, hidden, bias=False)        # unused
        nn.Conv2d(hidden, dim, 1)  # the reference builds and discards this (unet_model.py:337-339): same RNG draw
        self.to_out = nn.Linear(hidden, dim, bias=False)


class _SignalEmbeddingCNN(_Holder):
    def __init__(self, init_channel, ups):
        super().__init__()
        chans = [init_channel, *ups]
        mods = []
        for a, b in zip(chans[:-1], chans[1:]):
            mods.append(nn.Conv1d(a, b, kernel_size=4, stride=2, padding=1))
            mods.append(nn.SiLU())
        self.emb_model = nn.Sequential(*mods)


class _SinusoidalPosEmb(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim


class Unet3D(nn.Module):
    """Drop-in for reference `src.unet_model.Unet3D` (src/unet_model.py:406-623); compute on gfx950."""

    def __init__(
        self,
        dim,
        out_dim=None,
        dim_mults=(1, 2, 4, 8),
        channels=2,
        self_condition=False,
        attn_heads=8,
        attn_dim_head=32,
        init_dim=None,
        init_kernel_size=7,
        use_sparse_linear_attn=True,
        resnet_groups=8,
        cond_bias=False,
        cond_attention='none',
        cond_attention_tokens=6,
        cond_to_time='add',
        padding_mode='zeros',
        sigmoid_last_channel=False,
    ):
        super().__init__()
        if padding_mode != 'zeros':
            raise ValueError('Unknown padding mode: {} (the gfx950 engine implements zero padding)'.format(padding_mode))
        if not use_sparse_linear_attn:
            raise NotImplementedError('use_sparse_linear_attn=False is not on the accelerated path')
        self.dim = dim
        self.channels = channels
        self.input_channels = channels * (2 if self_condition else 1)
        self.self_condition = self_condition
        self.dim_mults = tuple(dim_mults)
        self.attn_heads = attn_heads
        self.attn_dim_head = attn_dim_head
        self.resnet_groups = resnet_groups
        self.init_kernel_size = init_kernel_size
        time_dim = dim * 4
        self.cond_bias = cond_bias
        self.cond_dim = time_dim
        self.cond_to_time = cond_to_time
        self.padding_mode = padding_mode

        rotary = _RotaryFreqs(min(32, attn_dim_head))

        def temporal_attn(d):
            return _Wrap(_Attention(d, attn_heads, attn_dim_head, rotary_emb=rotary, cond_dim=self.cond_dim))

        self.time_rel_pos_bias = _RelPosBias(attn_heads)
        init_dim = default(init_dim, dim)
        assert init_kernel_size % 2 == 1
        pad = init_kernel_size // 2
        self.init_conv = nn.Conv3d(self.input_channels, init_dim, (1, init_kernel_size, init_kernel_size),
                                   padding=(0, pad, pad))
        self.init_temporal_attn = _Wrap(_PreNorm(init_dim, temporal_attn(init_dim)))

        dims = [init_dim, *[dim * m for m in dim_mults]]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.time_mlp = nn.Sequential(_SinusoidalPosEmb(dim), nn.Linear(dim, time_dim), nn.GELU(),
                                      nn.Linear(time_dim, time_dim))
        self.sign_emb_CNN = _SignalEmbeddingCNN(1, (16, 32, 64, 128, self.cond_dim))

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        n_res = len(in_out)
        tdim = time_dim + int(self.cond_dim or 0) if cond_to_time == 'concat' else self.cond_dim

        def lin_attn(d):
            return _Wrap(_PreNorm(d, _SpatialLinearAttention(d, attn_heads, cond_dim=self.cond_dim)))

        for ind, (din, dout) in enumerate(in_out):
            last = ind >= n_res - 1
            self.downs.append(nn.ModuleList([
                _ResnetBlock(din, dout, tdim, resnet_groups),
                _ResnetBlock(dout, dout, tdim, resnet_groups),
                lin_attn(dout),
                nn.Conv3d(dout, dout, (1, 4, 4), (1, 2, 2), (0, 1, 1)) if not last else nn.Identity(),
            ]))
        mid = dims[-1]
        self.mid_block1 = _ResnetBlock(mid, mid, tdim, resnet_groups)
        self.mid_spatial_attn = _Wrap(_PreNorm(mid, _Wrap(_Attention(mid, attn_heads, cond_dim=self.cond_dim))))
        self.mid_temporal_attn = _Wrap(_PreNorm(mid, temporal_attn(mid)))
        self.mid_block2 = _ResnetBlock(mid, mid, tdim, resnet_groups)
        for ind, (din, dout) in enumerate(reversed(in_out)):
            last = ind >= n_res - 1
            self.ups.append(nn.ModuleList([
                _ResnetBlock(dout * 2, din, tdim, resnet_groups),
                _ResnetBlock(din, din, tdim, resnet_groups),
                lin_attn(din),
                nn.ConvTranspose3d(din, din, (1, 4, 4), (1, 2, 2), (0, 1, 1)) if not last else nn.Identity(),
            ]))
        out_dim = default(out_dim, channels)
        self.out_dim = out_dim
        self.final_conv = nn.Sequential(_ResnetBlock(dim * 2, dim, None, resnet_groups), nn.Conv3d(dim, out_dim, 1))
        self.emb_conv = nn.Sequential(nn.Conv2d(channels, init_dim, 1), nn.GELU(),
                                      nn.Conv2d(init_dim, init_dim, 3, padding=1))
        self.combine_conv = nn.Conv2d(init_dim * 2, init_dim, 1)
        self.sigmoid_last_channel = sigmoid_last_channel
        self._pidm_lib = None  # tests may bind the host-emulated build of csrc here; None = libpidm_hip.so

    # ---- engine plumbing ------------------------------------------------------------------------
    # per-model engine state: native handles (ctypes pointers cannot be pickled / deep-copied), workspaces, flat buffers
    _PIDM_PRIVATE = ('_engines', '_pidm_flat_params', '_pidm_frozen', '_pidm_tape_slot')

    def __deepcopy__(self, memo):
        """copy.deepcopy(model) - what EMA.ema_copy and user code do - copies the nn.Module state only: the copy builds its
        own engine (handle, workspace, gradient buffer) on its first forward; `_pidm_lib` (which library to bind) is shared."""
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in self._PIDM_PRIVATE:
                continue
            new.__dict__[k] = v if k == '_pidm_lib' else copy.deepcopy(v, memo)
        return new

    def used_parameter_names(self):
        """Names (state_dict order) of the tensors `forward` reads - 259 for the default config
        (SURVEY Appendix A); all others keep `.grad is None`, as in the reference."""
        from ._engine import used_parameter_names
        return used_parameter_names(self)

    def forward(self, x, time, x_self_cond=None, cond=None, null_cond_prob=0.):
        """`cond` [B, P*P, C] = the residual-gradient conditioning field of the guidance baseline
        (src/unet_model.py:571-587): per-sample classifier-free dropout with probability `null_cond_prob` (same RNG call
        as the reference's prob_mask_like), then emb_conv / combine_conv inside the engine."""
        from ._engine import unet_apply
        if cond is not None:
            if cond.dim() != 3:
                raise ValueError('Input must be [BxP*PxC].')
            mask = prob_mask_like((cond.shape[0],), null_cond_prob, device=cond.device)
            cond = torch.where(mask.view(-1, 1, 1), torch.zeros_like(cond), cond)
        return unet_apply(self, x, time, lib=self._pidm_lib, cond=cond, x_self_cond=x_self_cond)

    def forward_with_guidance_scale(self, *args, **kwargs):
        """null + (cond - null) * scale from two forward passes (src/unet_model.py:530-540)."""
        guidance_scale = kwargs.pop('guidance_scale', 3.)
        logits = self.forward(*args, null_cond_prob=0., **kwargs)
        if guidance_scale == 1:
            return logits
        null_logits = self.forward(*args, null_cond_prob=1., **kwargs)
        return null_logits + (logits - null_logits) * guidance_scale
