"""Host glue between the `Unet3D` parameter container and the gfx950 UNet engine (C ABI, include/pidm.h).

PyTorch is plumbing here: it owns the parameter / gradient / workspace memory and the RNG; every FLOP of the
forward and backward pass runs in hand-written HIP kernels behind `pidm_unet_forward` / `pidm_unet_backward`.
There is no CPU or eager fallback: tensors must live on an MI355X and `libpidm_hip.so` must be built.
(The unit tests bind the host-emulated build of the same C sources by passing `lib=` explicitly.)
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math

import torch

from ._lib import PidmError, PidmLib, UnetCfg, get_lib, ptr, stream_ptr, vp


class UnetEngine:
    """One native engine handle per (model, image size).  Not thread-safe (one per process per device)."""

    def __init__(self, model, image_size: int, lib: PidmLib | None = None):
        self.lib = lib or get_lib()
        cfg = UnetCfg()
        cfg.dim = model.dim
        cfg.channels = model.channels
        cfg.out_dim = model.out_dim
        cfg.n_levels = len(model.dim_mults)
        for i, m in enumerate(model.dim_mults):
            cfg.dim_mults[i] = int(m)
        cfg.heads = model.attn_heads
        cfg.dim_head = model.attn_dim_head
        cfg.groups = model.resnet_groups
        cfg.init_kernel = model.init_kernel_size
        cfg.image_size = image_size
        cfg.sigmoid_last_channel = int(bool(model.sigmoid_last_channel))
        cfg.self_condition = int(bool(model.self_condition))
        self.image_size = image_size
        self.handle = vp()
        self.lib.check(self.lib.pidm_unet_create(C.byref(cfg), C.byref(self.handle)), "pidm_unet_create")
        n = self.lib.pidm_unet_num_params(self.handle)
        self.names = [self.lib.pidm_unet_param_name(self.handle, i).decode() for i in range(n)]
        self.numels = [self.lib.pidm_unet_param_numel(self.handle, i) for i in range(n)]
        named = dict(model.named_parameters())
        missing = [k for k in self.names if k not in named]
        if missing:
            raise PidmError(f"engine parameter(s) not found in the model: {missing[:4]}")
        for k, ne in zip(self.names, self.numels):
            if named[k].numel() != ne:
                raise PidmError(f"parameter {k}: numel {named[k].numel()} != engine's {ne}")
        self.params = [named[k] for k in self.names]
        self.n_cond = self.lib.pidm_unet_num_cond_params(self.handle)   # trailing entries: emb_conv / combine_conv
        self.cond_enabled = False
        self.flat_grad = None
        self.grad_views = None
        # staging buffer of an EARLY backward (denoising_utils._DarcyStepFn): the pass runs at forward time into this buffer and
        # `loss.backward()` later scales / copies it into `flat_grad` (which `p.grad` alias) - never visible to the caller
        self.early_grad = None
        self.early_views = None
        self.early_generation = -1   # tape generation whose gradients `early_grad` holds
        self._bound_key = None
        self.workspace = None
        self.backward_calls = 0      # backward passes since the last gradient exchange (parallel.GradientExchange)
        self.tape_generation = 0
        self.tape_busy = False   # a training-mode forward whose backward has not run yet owns the tape
        self._packed_for = None  # (frozen-scope token, bound key, workspace) of the last inference forward that re-packed the weights

    def __del__(self):
        try:
            if self.handle:
                self.lib.pidm_unet_destroy(self.handle)
        except Exception:
            pass

    # ---- memory owned by torch, borrowed by the engine -------------------------------------------------
    def _flat_buffer(self, dev):
        total = sum(self.numels)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        views, off = [], 0
        for p, ne in zip(self.params, self.numels):
            views.append(flat[off:off + ne].view(p.shape))
            off += ne
        return flat, views

    def _ensure_bound(self, need_grad: bool, early: bool = False):
        """Binds parameter (and gradient) pointers.  early=True: the backward pass writes the private staging buffer."""
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise PidmError("engine parameters must be contiguous fp32 tensors on one device")
        if need_grad and (self.flat_grad is None or self.flat_grad.device != dev):
            # one flat gradient buffer in the engine's canonical order: p.grad become views of it (a single
            # all-reduce payload for data parallel training, contiguous FiLM-linear gradients for the engine)
            self.flat_grad, self.grad_views = self._flat_buffer(dev)
            self._bound_key = None
        if early and (self.early_grad is None or self.early_grad.device != dev):
            self.early_grad, self.early_views = self._flat_buffer(dev)
        # once a gradient buffer exists it stays bound: an inference-mode forward between a training forward and its
        # backward (EMA evaluation, sampling inside a step) must not unbind it
        with_grad = need_grad or self.flat_grad is not None
        target = (self.early_views if early else self.grad_views) if with_grad else None
        key = (tuple(p.data_ptr() for p in self.params), with_grad and target[0].data_ptr())
        if key != self._bound_key:
            n = len(self.params)
            pp = (vp * n)(*[vp(p.data_ptr()) for p in self.params])
            if with_grad:
                gp = (vp * n)(*[vp(g.data_ptr()) for g in target])
                self.lib.check(self.lib.pidm_unet_bind(self.handle, pp, gp), "pidm_unet_bind")
            else:
                self.lib.check(self.lib.pidm_unet_bind(self.handle, pp, None), "pidm_unet_bind")
            self._bound_key = key

    def _ensure_workspace(self, B: int, training: bool, device):
        # asked on every call (a cached lookup in the library): the plan depends on knobs that tests / A-B runs flip on a
        # live handle (PIDM_NO_LAP, PIDM_LAP_MIN_N), not only on the batch size
        nbytes = self.lib.pidm_unet_workspace_bytes(self.handle, B, int(training))
        if nbytes == 0:
            raise PidmError("pidm_unet_workspace_bytes: " + self.lib.lib.pidm_last_error().decode())
        if self.workspace is None or self.workspace.numel() < nbytes + 256 or self.workspace.device != torch.device(device):
            self.workspace = None
            self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        return self.workspace

    # ---- forward / backward ------------------------------------------------------------------------------
    def forward(self, x_nhwc: torch.Tensor, t: torch.Tensor, training: bool, repack: bool = True,
                cond: torch.Tensor | None = None, early: bool = False, frozen=None) -> torch.Tensor:
        """`frozen`: the token of an enclosing `frozen_weights(model)` scope (None outside one).  Inside such a scope the
        caller guarantees constant parameters, so only the scope's first inference forward of this engine re-packs (and
        re-splits) the weights; everywhere else every forward does (the parameters may have been written by anything)."""
        B = x_nhwc.shape[0]
        dev = x_nhwc.device
        self._ensure_bound(training, early=early)
        if cond is not None:
            if not self.cond_enabled:          # size the workspace for the conditioning branch from now on
                self.lib.check(self.lib.pidm_unet_enable_cond(self.handle, 1), "pidm_unet_enable_cond")
                self.cond_enabled = True
            self.lib.check(self.lib.pidm_unet_set_condition(self.handle, ptr(cond)), "pidm_unet_set_condition")
        ws = self._ensure_workspace(B, training, dev)
        P = self.image_size
        out = torch.empty(B, self.lib_out_dim, P, P, dtype=torch.float32, device=dev)
        base = ws.data_ptr()
        al = (-base) % 256
        if frozen is not None and not training and repack:
            key = (frozen, self._bound_key, base)
            repack = key != self._packed_for     # same scope, same parameter storage, same workspace: the packed weights are current
            self._packed_for = key
        else:
            self._packed_for = None
        self.lib.check(self.lib.pidm_unet_forward(self.handle, ptr(x_nhwc), ptr(t), ptr(out), B, int(training), int(repack),
                                                  vp(base + al), ws.numel() - al, stream_ptr(dev)), "pidm_unet_forward")
        return out

    def backward(self, grad_out: torch.Tensor, want_grad_x: bool, channels: int):
        B = grad_out.shape[0]
        dev = grad_out.device
        ws = self.workspace
        gx = torch.empty(B, self.image_size * self.image_size, channels, dtype=torch.float32, device=dev) if want_grad_x else None
        base = ws.data_ptr()
        al = (-base) % 256
        self.lib.check(self.lib.pidm_unet_backward(self.handle, ptr(grad_out), ptr(gx), B, vp(base + al), ws.numel() - al,
                                                   stream_ptr(dev)), "pidm_unet_backward")
        return gx


class _TapeLease:
    """Lives in the autograd ctx of a training-mode forward.  When the graph is dropped WITHOUT a backward (main.py:187 runs
    the validation loss with grad enabled and never differentiates it) the ctx dies, and with it this object: the engine's
    tape is then free again instead of staying "busy" for good."""

    __slots__ = ("engine", "generation")

    def __init__(self, engine, generation):
        self.engine, self.generation = engine, generation

    def release(self):
        eng = self.engine
        if eng is not None and eng.tape_generation == self.generation:
            eng.tape_busy = False
        self.engine = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class _UnetFunction(torch.autograd.Function):
    """autograd node whose forward/backward are single C-ABI calls.  Parameter gradients are written by the
    engine into the flat gradient buffer and attached as `p.grad` views (accumulated if a grad already exists),
    so `optimizer.zero_grad(); loss.backward(); clip_grad_norm_; optimizer.step()` behaves as in main.py:163-166.
    Parameters that forward never reads keep `grad is None`, exactly like the reference."""

    @staticmethod
    def forward(ctx, engine: UnetEngine, x_nhwc, t, anchor, training: bool, cond=None, primary=None, frozen=None):
        ctx.engine = engine
        ctx.primary = primary if primary is not None else engine
        ctx.used_cond = cond is not None
        ctx.x_requires_grad = x_nhwc.requires_grad
        ctx.channels = x_nhwc.shape[-1]
        ctx.training = training
        if training:
            engine.tape_generation += 1
            engine.tape_busy = True
        ctx.generation = engine.tape_generation
        ctx.lease = _TapeLease(engine, engine.tape_generation) if training else None
        out = engine.forward(x_nhwc, t, training=training, cond=cond, frozen=frozen)
        # the engine keeps RAW pointers to its inputs and output until backward: keep the tensors alive
        if cond is not None:
            ctx.save_for_backward(out, x_nhwc, cond)
        else:
            ctx.save_for_backward(out, x_nhwc)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.engine
        if not ctx.training:
            raise PidmError("backward through a forward that ran without gradient tracking")
        if ctx.generation != eng.tape_generation:
            raise PidmError("the engine's activation tape was overwritten by a later training-mode forward of the same "
                            "model: only the latest differentiable UNet call can be differentiated (the two calls of "
                            "x0_estimation='sample' get one tape each through DenoisingDiffusion.ddim_sample_x0)")
        # Every backward of this model lands in ONE flat buffer - the primary engine's (slot 0): `p.grad` are views of it, the
        # data-parallel all-reduce and the fused optimizer read it.  An engine of another slot (second activation tape of
        # x0_estimation='sample') writes its own buffer, which is then copied / added into the primary one, whatever order
        # autograd runs the two nodes in.
        primary = ctx.primary
        first = eng.params[0]
        if primary.flat_grad is None:
            primary._ensure_bound(True)
        aliased = first.grad is not None and first.grad.data_ptr() == primary.grad_views[0].data_ptr()
        # the engine WRITES its flat gradient buffer.  If p.grad already aliases that buffer (a second backward without
        # zero_grad, or zero_grad(set_to_none=False)), torch semantics are accumulation: keep the old contents and add them
        # back afterwards (one 4-byte-per-parameter copy, only on this path - main.py's zero_grad() sets grads to None)
        prev = primary.flat_grad.clone() if (aliased and eng is primary) else None
        gx = eng.backward(grad_out.contiguous(), ctx.x_requires_grad, ctx.channels)
        if prev is not None:
            eng.flat_grad.add_(prev)
        if eng is not primary:
            if aliased:
                primary.flat_grad.add_(eng.flat_grad)
            else:
                primary.flat_grad.copy_(eng.flat_grad)
        if ctx.lease is not None:
            ctx.lease.release()
        eng.tape_busy = False
        # the phase events of the overlapped exchange describe the buffer only if this was the step's one plain backward
        primary.backward_calls += 1 if (eng is primary and prev is None) else 2
        n_plain = len(eng.params) - eng.n_cond
        for i, (p, g) in enumerate(zip(eng.params, primary.grad_views)):
            if not p.requires_grad:
                continue
            if i >= n_plain and not ctx.used_cond:
                continue        # conditioning branch not used by this forward: grads stay None, as in the reference
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() != g.data_ptr():
                p.grad.add_(eng.grad_views[i] if eng is not primary else g)
            # else: p.grad already aliases the primary buffer, which now holds the accumulated gradient
        return None, gx, None, None, None, None, None, None


def get_engine(model, image_size: int, lib: PidmLib | None = None, slot: int = 0) -> UnetEngine:
    cache = model.__dict__.setdefault("_engines", {})
    key = (image_size, id(lib), slot)
    eng = cache.get(key)
    if eng is None:
        eng = UnetEngine(model, image_size, lib)
        eng.lib_out_dim = model.out_dim
        cache[key] = eng
    return eng


def used_parameter_names(model, image_size: int = 64, with_condition: bool = False):
    eng = get_engine(model, image_size)
    return list(eng.names) if with_condition else list(eng.names[:len(eng.names) - eng.n_cond])


@contextlib.contextmanager
def frozen_weights(model):
    """Scope in which the caller guarantees that `model`'s parameters do not change (the sampler loop, the no-grad
    evaluations of one x0 estimate): the engine re-packs / re-splits the weights for the first inference forward only
    (csrc/k_conv.hip: pack_multi_kernel, 113 us and 71 MB per call at dim 32).  Nesting is allowed; outside a scope every
    forward re-packs, because raw-pointer writers (fused optimizer, EMA kernels, `p.data` arithmetic) leave no trace."""
    prev = getattr(model, "_pidm_frozen", None)
    model._pidm_frozen = prev if prev is not None else object()
    try:
        yield
    finally:
        model._pidm_frozen = prev


def unet_apply(model, x, time, lib: PidmLib | None = None, cond=None, x_self_cond=None):
    """x: [B,P*P,C] (reference interchange layout), [B,C,P,P] or [B,C,1,P,P]; time: int64 [B].
    Returns [B,out_dim,P,P] (or [B,out_dim,1,P,P] for 5-D input), as reference Unet3D.forward does."""
    video = False
    if x.dim() == 3:
        B, N, Cc = x.shape
        P = int(math.isqrt(N))
        if P * P != N:
            raise ValueError('Input [B, P*P, C] needs a square number of pixels.')
        x_nhwc = x
    elif x.dim() == 4:
        B, Cc, P, P2 = x.shape
        x_nhwc = x.permute(0, 2, 3, 1).reshape(B, P * P2, Cc)
    elif x.dim() == 5:
        video = True
        B, Cc, F_, P, P2 = x.shape
        if F_ != 1:
            raise NotImplementedError('the gfx950 engine runs the image path (singleton frame axis) only')
        x_nhwc = x[:, :, 0].permute(0, 2, 3, 1).reshape(B, P * P2, Cc)
    else:
        raise ValueError('Input must be image [BxCxPxP] or image sequence [BxCxFxPxP].')
    if Cc != model.channels:
        raise ValueError(f'expected {model.channels} input channels, got {Cc}')
    if lib is None and not x.is_cuda:
        raise PidmError("Unet3D.forward needs tensors on an MI355X (cuda device): the gfx950 engine has no CPU fallback")
    if model.self_condition:
        # init_conv reads cat(x_self_cond, x) (src/unet_model.py:564-566); absent self-conditioning input = zeros
        if x_self_cond is None:
            sc = torch.zeros_like(x_nhwc)
        elif x_self_cond.dim() == 3:
            sc = x_self_cond
        elif x_self_cond.dim() == 4:
            sc = x_self_cond.permute(0, 2, 3, 1).reshape(B, P * P, Cc)
        else:
            sc = x_self_cond[:, :, 0].permute(0, 2, 3, 1).reshape(B, P * P, Cc)
        x_nhwc = torch.cat((sc.to(x_nhwc.dtype), x_nhwc), dim=-1)
    elif x_self_cond is not None:
        raise ValueError('x_self_cond given but the model was built with self_condition=False')
    x_nhwc = x_nhwc.contiguous().float()
    t = time.to(device=x.device, dtype=torch.int64).contiguous()
    if t.numel() != B:
        raise ValueError('time must have one entry per batch element')
    primary = eng = get_engine(model, P, lib)
    anchor = eng.params[0]
    training = torch.is_grad_enabled() and (anchor.requires_grad or x_nhwc.requires_grad)
    slot = getattr(model, "_pidm_tape_slot", None)
    if training and slot:
        # a second differentiable UNet call inside one step (x0_estimation: 'sample' evaluates the model at (x_t, t)
        # and at (x_t, 0), src/denoising_utils.py:741-753): it gets its own engine slot = own activation tape and own
        # gradient buffer.  The slot is NAMED by the caller (ddim_sample_x0: 0 then 1), so a tape left "busy" by a loss that
        # was never differentiated is simply overwritten - the number of engines per model is bounded by the slots in use.
        eng = get_engine(model, P, lib, int(slot))
    if not training and eng.tape_busy:
        # an inference-mode forward while a training forward of this model still waits for its backward (evaluation inside a
        # step): run it on a sibling engine so that the activation tape and its pointers stay intact
        eng = get_engine(model, P, lib, slot=-1)
    if cond is not None:
        if cond.shape != (B, P * P, model.channels):
            raise ValueError(f'cond must be [B, P*P, {model.channels}], got {tuple(cond.shape)}')
        cond = cond.detach().contiguous().float()
    out = _UnetFunction.apply(eng, x_nhwc, t, anchor, training, cond, primary, getattr(model, "_pidm_frozen", None))
    if video:
        out = out.unsqueeze(2)
    return out
