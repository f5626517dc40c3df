"""Diffusion process: host-side mirror of reference `src/denoising_utils.py` (DenoisingDiffusion, EMA,
save_model/load_model, extract and the helpers main.py / sample.py star-import).

`DenoisingDiffusion.model_estimation_loss` / `p_sample` / `p_sample_loop` keep the reference signatures and
return structures.  For the Darcy mean-estimation configuration (the `north_star` hot path) the step is three
native calls: fused q-sample (csrc/k_norm.hip), UNet forward (csrc/unet_engine.hip), fused residual + loss +
d loss/d x0_pred (csrc/k_darcy.hip); `loss.backward()` then runs the native UNet backward.  Other
configurations compose the same native pieces through autograd with the reference's op sequence.
"""
from __future__ import annotations

import contextlib
import os
from pathlib import Path  # noqa: F401  (re-exported: main.py uses `Path` from the star-import)

import numpy as np  # noqa: F401  (re-exported)
import torch
import torch.nn as nn
import torch.nn.functional as F
import yaml

from ._lib import get_lib, ptr, stream_ptr
from .residuals_darcy import ResidualsDarcy
from .unet_model import (default, exists, generalized_b_xy_c_to_image, generalized_image_to_b_xy_c,  # noqa: F401
                         noop)

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def fix_seeds(seed=42):
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    np.random.seed(seed)


def image_to_b_xy_c(tensor):
    """[B,C,X,Y] -> [B,X*Y,C] (a strided view, like the reference's permute+view; src/denoising_utils.py:36-42)."""
    assert tensor.dim() == 4, 'Input tensor must have shape [batch, channels, x, y].'
    b, c, px, py = tensor.shape
    return tensor.permute(0, 2, 3, 1).reshape(b, px * py, c)


def b_xy_c_to_image(tensor, pixels_x=None, pixels_y=None):
    assert tensor.dim() == 3, 'Input tensor must have shape [batch, x*y, channels].'
    b, n, c = tensor.shape
    if pixels_x is None and pixels_y is None:
        assert np.sqrt(n) % 1 == 0, 'Number of pixels must be a perfect square.'
        pixels_x = pixels_y = int(np.sqrt(n))
    else:
        assert pixels_x * pixels_y == n, 'Number of given pixels must match dim 1 of input tensor.'
    return tensor.reshape(b, pixels_x, pixels_y, c).permute(0, 3, 1, 2)


def resize_image(tensor, target_size):
    """[B, C..., X, Y] -> [B, C..., target, target], bilinear without antialiasing (src/denoising_utils.py:57-68: torchvision
    Resize = F.interpolate(align_corners=False)).  Forward-only tensors on an MI355X go through the native resize kernel
    (csrc/k_mech.hip); anything that needs a gradient, or lives on the host, takes torch's own interpolate."""
    assert len(tensor.shape) > 3, f"Expected image, got {tensor.shape}"
    shape = tensor.shape
    flat = tensor.reshape(shape[0], -1, shape[-2], shape[-1])
    if tensor.is_cuda and not tensor.requires_grad and tensor.dtype == torch.float32 and shape[-1] == shape[-2]:
        from .residuals_mechanics_K import resize_image as _native_resize
        out = _native_resize(flat.contiguous(), target_size)
    else:
        out = F.interpolate(flat, size=(target_size, target_size), mode='bilinear', align_corners=False, antialias=False)
    return out.view(shape[0], *shape[1:-2], target_size, target_size)


def right_pad_dims_to(x, t):
    padding_dims = x.ndim - t.ndim
    if padding_dims <= 0:
        return t
    return t.view(*t.shape, *((1,) * padding_dims))


def extract(input, t, x):
    """gather(table, t) reshaped to [B,1,...] (src/denoising_utils.py:302-306)."""
    out = torch.gather(input, 0, t.to(input.device))
    return out.reshape(t.shape[0], *([1] * (len(x.shape) - 1)))


class EMA(object):
    """Shadow-weight EMA with the reference's swap-in / swap-out protocol (src/denoising_utils.py:163-205): `update` after each
    optimizer step, `ema(model)` swaps the averaged weights in, `restore` swaps back - main.py does all three EVERY iteration
    (main.py:178-183,316).  Same dict-of-tensors state (`shadow`, what `state_dict()` returns and checkpoints hold) and the same
    arithmetic, (1-mu)*p + mu*shadow with every product and the sum rounded to fp32, so the shadow is bit-identical to the
    reference's.  What differs is how it is executed:

    * `update`: once the model's parameters live in the engine's flat buffer (`optim.flatten_parameters`, done by
      `FusedClipAdam`) the shadow is re-homed into one flat buffer too and the update is ONE kernel (`pidm_ema_update`); with
      `FusedClipAdam(..., ema=this)` it rides inside the Adam kernel and `update` only acknowledges it.  Parameters outside the
      flat buffer (the ones forward never reads) take three multi-tensor launches.
    * `ema` / `restore`: a POINTER FLIP instead of three full copies - `param.data` is re-pointed at the shadow tensor and back
      (the engine re-reads parameter pointers on every call).  Between `ema()` and `restore()` the parameters ARE the shadow:
      evaluate, sample and checkpoint, but do not run an optimizer step there (`FusedClipAdam.step` refuses).
      `ema(module, backup=False)` has no way back, so it copies, as the reference does."""

    def __init__(self, mu=0.999):
        self.mu = mu
        self.shadow = {}
        self.backup = {}
        self._flat = None            # (flat shadow buffer, flat parameter buffer, names) once re-homed
        self._fused_updates = 0      # updates already applied inside FusedClipAdam.step and not yet acknowledged

    @staticmethod
    def _named(module):
        return [(n, p) for n, p in module.named_parameters() if p.requires_grad]

    def register(self, module):
        for name, param in self._named(module):
            self.shadow[name] = param.data.clone()
        self._flat = None

    # ---- flat layout ------------------------------------------------------------------------------------------
    def _flat_layout(self, module):
        """(flat shadow, flat params, engine) when the module's engine parameters are flat and all trainable, else None.
        Re-homes the shadow tensors of those parameters into one buffer (same values) the first time, and again whenever
        the shadow dict was replaced (load_state_dict) or the parameters were re-flattened."""
        if self.backup:
            return None               # swapped in: parameter pointers are the shadow's right now
        pflat = module.__dict__.get("_pidm_flat_params")
        engines = module.__dict__.get("_engines")
        if pflat is None or not engines:
            return None
        eng = next(iter(engines.values()))
        off = pflat.data_ptr()
        for p, ne in zip(eng.params, eng.numels):
            if p.data_ptr() != off or not p.requires_grad:
                return None
            off += 4 * ne
        names = eng.names
        if any(n not in self.shadow for n in names):
            return None
        sflat = self._flat[0] if self._flat is not None and self._flat[1] is pflat else None
        ok = sflat is not None
        if ok:
            off = sflat.data_ptr()
            for n, ne in zip(names, eng.numels):
                if self.shadow[n].data_ptr() != off:
                    ok = False
                    break
                off += 4 * ne
        if not ok:
            sflat = torch.empty_like(pflat)
            off = 0
            for n, p, ne in zip(names, eng.params, eng.numels):
                view = sflat[off:off + ne].view(p.shape)
                view.copy_(self.shadow[n].to(sflat.device))
                self.shadow[n] = view
                off += ne
            self._flat = (sflat, pflat, tuple(names))
        return sflat, pflat, eng

    def update(self, module):
        if self.backup:
            # between ema() and restore() the parameters ARE the shadow tensors (pointer flip): averaging the shadow with itself
            # would silently drop this step's update
            raise RuntimeError("EMA.update() while the averaged weights are swapped in: call ema.restore(model) first")
        named = self._named(module)
        rest = named
        lay = self._flat_layout(module)
        if lay is not None:
            sflat, pflat, eng = lay
            if self._fused_updates > 0:
                self._fused_updates -= 1          # FusedClipAdam.step already applied this update inside the Adam kernel
            else:
                eng.lib.check(eng.lib.pidm_ema_update(ptr(sflat), ptr(pflat), pflat.numel(), float(self.mu),
                                                      stream_ptr(pflat.device)), 'pidm_ema_update')
            flat_names = set(self._flat[2])
            rest = [(n, p) for n, p in named if n not in flat_names]
        if rest:
            shadows = [self.shadow[n].data for n, _ in rest]
            scaled = torch._foreach_mul([p.data for _, p in rest], 1. - self.mu)     # (1-mu)*p, rounded
            torch._foreach_mul_(shadows, self.mu)                                     # mu*shadow, rounded
            torch._foreach_add_(shadows, scaled)                                      # their sum, rounded

    def ema(self, module, backup=True):
        named = self._named(module)
        for n, _ in named:
            assert n in self.shadow
        if not backup:
            torch._foreach_copy_([p.data for _, p in named], [self.shadow[n].data for n, _ in named])
            return
        dev = named[0][1].device if named else None
        for n, p in named:
            sh = self.shadow[n]
            if sh.device != dev:
                sh = self.shadow[n] = sh.to(dev)
            self.backup[n] = p.data
            p.data = sh

    def restore(self, module):
        assert hasattr(self, 'backup')
        for n, p in self._named(module):
            assert n in self.backup
            p.data = self.backup[n]
        self.backup = {}

    def ema_copy(self, module):
        """A second model holding the averaged weights (src/denoising_utils.py:195-199; the reference builds it from
        `module.config`, which its own Unet3D never sets - here the constructor arguments the model was built with are used when
        it has no `config`).  The copy owns its weights (`backup=False`: copied, not aliased to the shadow)."""
        if hasattr(module, 'config'):
            module_copy = type(module)(module.config).to(module.config.device)
        else:
            import copy
            module_copy = copy.deepcopy(module)                 # Unet3D.__deepcopy__ leaves the engine state behind
            for k in ('_engines', '_pidm_flat_params', '_pidm_frozen', '_pidm_tape_slot'):   # (other module types)
                module_copy.__dict__.pop(k, None)
        module_copy.load_state_dict(module.state_dict())
        self.ema(module_copy, backup=False)
        return module_copy

    def state_dict(self):
        return self.shadow

    def load_state_dict(self, state_dict):
        self.shadow = state_dict
        self._flat = None


def image_array_to_gif(image_array, output_file, frame_duration=0.05, normalization_mode='final_pred', given_min_max=None):
    """Animated GIF of a [frames, H, W] array (src/denoising_utils.py:244-271; sample.py calls it with create_gif=True by
    default, sample.py:25,212-214,312).  Grey-level range: 'final_pred' = range of the last frame, 'global' = of the whole
    array, 'given' = `given_min_max`, 'individual' = per frame, 'none' = frames are written as they are.  Host-side IO, not
    on the accelerated path: written with imageio when it is installed (what the reference uses), otherwise with Pillow;
    with neither, a warning is printed and nothing is written (the sampling results themselves are saved by the caller)."""
    frames = np.asarray(image_array)
    if normalization_mode == 'given' and given_min_max is None:
        raise ValueError("Please provide min and max values for 'given' normalization mode.")
    if normalization_mode not in ('final_pred', 'global', 'given', 'individual', 'none'):
        raise ValueError(f'unknown normalization_mode {normalization_mode!r}')
    if normalization_mode != 'none':
        if normalization_mode == 'individual':
            flat = frames.reshape(len(frames), -1)
            lo, hi = flat.min(axis=1), flat.max(axis=1)
            lo, hi = lo.reshape(-1, 1, 1), hi.reshape(-1, 1, 1)
        else:
            ref = {'final_pred': frames[-1], 'global': frames}.get(normalization_mode)
            lo, hi = given_min_max if ref is None else (ref.min(), ref.max())
        with np.errstate(divide='ignore', invalid='ignore'):
            frames = ((frames - lo) / (hi - lo) * 255).astype(np.uint8)     # same arithmetic (and wrap-around) as the reference
    try:
        import imageio
    except ImportError:
        imageio = None
    if imageio is not None:
        with imageio.get_writer(output_file, mode='I', duration=frame_duration) as writer:
            for frame in frames:
                writer.append_data(frame)
        return
    try:
        from PIL import Image
    except ImportError:
        print(f'image_array_to_gif: neither imageio nor Pillow is installed - {output_file} not written')
        return
    imgs = [Image.fromarray(np.ascontiguousarray(f if f.dtype == np.uint8 else np.clip(f, 0, 255).astype(np.uint8)))
            for f in frames]
    imgs[0].save(output_file, save_all=True, append_images=imgs[1:], duration=max(int(round(frame_duration * 1000)), 10), loop=0)


def save_model(config, model, train_iterations, output_save_dir):
    """{dir}/model/model.yaml + checkpoint_{it}.pt = {'model': state_dict} (src/denoising_utils.py:273-287)."""
    os.makedirs(Path(output_save_dir, 'model/'), exist_ok=True)
    with open(output_save_dir + '/model/model.yaml', 'w') as f:
        yaml.dump(dict(config), f, default_flow_style=False)
    with open(output_save_dir + '/model/checkpoint_' + str(train_iterations) + '.pt', 'wb') as f:
        torch.save(dict(model=model.state_dict()), f)
    print(f'\ncheckpoint saved to {output_save_dir}/.')


def load_model(path, model, strict=True):
    with open(path, 'rb') as f:
        loaded_obj = torch.load(f, map_location='cpu')
    try:
        model.load_state_dict(loaded_obj['model'], strict=strict)
    except RuntimeError:
        print('Failed loading state dict.')
    print('\nCheckpoint loaded from {}'.format(path))
    return model


def save_training_state(path, model, optimizer=None, ema=None, iteration=0, extra=None):
    """Extension: everything needed to resume a run bit-exactly - the reference checkpoints hold the
    model weights only (src/denoising_utils.py:273-287).  `model` keys are the reference's 317 `state_dict` entries, so the
    file doubles as a `load_model` checkpoint.  optimizer: torch.optim.Adam or optim.FusedClipAdam; ema: EMA."""
    state = {'model': model.state_dict(), 'iteration': int(iteration), 'rng_cpu': torch.get_rng_state()}
    if torch.cuda.is_available():
        state['rng_cuda'] = torch.cuda.get_rng_state_all()
    if optimizer is not None:
        state['optimizer'] = optimizer.state_dict()
    if ema is not None:
        state['ema'] = ema.state_dict()
    if extra is not None:
        state['extra'] = extra
    with open(path, 'wb') as f:
        torch.save(state, f)


def load_training_state(path, model, optimizer=None, ema=None, strict=True):
    """Restores what save_training_state wrote (in place); returns (iteration, extra)."""
    with open(path, 'rb') as f:
        state = torch.load(f, map_location='cpu', weights_only=False)
    model.load_state_dict(state['model'], strict=strict)      # copies into the existing (possibly flattened) Parameters
    if optimizer is not None and 'optimizer' in state:
        optimizer.load_state_dict(state['optimizer'])
    if ema is not None and 'ema' in state:
        dev = next(model.parameters()).device
        ema.shadow = {k: v.to(dev) for k, v in state['ema'].items()}
    torch.set_rng_state(state['rng_cpu'])
    if 'rng_cuda' in state and torch.cuda.is_available():
        torch.cuda.set_rng_state_all(state['rng_cuda'])
    return state['iteration'], state.get('extra')


class _ScalarFetch:
    """The loss scalars of one step on their way to the host: one non-blocking copy into pinned memory + an event."""

    def __init__(self, host, event):
        self.host, self.event, self._vals = host, event, None

    def values(self):
        if self._vals is None:
            self.event.synchronize()
            self._vals = self.host.tolist()
            self.host = None
        return self._vals


class DeferredFloat:
    """One of the floats `model_estimation_loss` returns (reference: `.item()` calls, src/denoising_utils.py:681,688,699,707)
    that synchronises with the GPU when it is first USED instead of when it is produced (`DenoisingDiffusion.deferred_scalars`).
    The reference's loop reads these values every `log_freq = 20` iterations only (main.py:167-175); returning them eagerly
    stalls the host once per step, and the GPU then idles ~0.7 ms while the host catches up with the next step's launches.
    Behaves like a float wherever one is formatted, converted, compared or used in arithmetic."""
    __slots__ = ("_src", "_idx")

    def __init__(self, src, idx):
        self._src, self._idx = src, idx

    def _v(self):
        v = self._src.values()
        for i in self._idx:
            v = v[i]
        return v

    def __float__(self): return float(self._v())
    def __format__(self, spec): return format(self._v(), spec)
    def __repr__(self): return repr(self._v())
    __str__ = __repr__
    def __bool__(self): return bool(self._v())
    def __hash__(self): return hash(self._v())
    def __eq__(self, o): return self._v() == float(o)
    def __lt__(self, o): return self._v() < float(o)
    def __le__(self, o): return self._v() <= float(o)
    def __gt__(self, o): return self._v() > float(o)
    def __ge__(self, o): return self._v() >= float(o)
    def __neg__(self): return -self._v()
    def __abs__(self): return abs(self._v())
    def __add__(self, o): return self._v() + float(o)
    __radd__ = __add__
    def __sub__(self, o): return self._v() - float(o)
    def __rsub__(self, o): return float(o) - self._v()
    def __mul__(self, o): return self._v() * float(o)
    __rmul__ = __mul__
    def __truediv__(self, o): return self._v() / float(o)
    def __rtruediv__(self, o): return float(o) / self._v()
    def item(self): return self._v()


class _DarcyPidmLossFn(torch.autograd.Function):
    """loss = c_data*mean_b(w_t*mse) + mean(c_r*0.5*r^2/var_t) in one kernel, together with d loss/d x0_pred
    (src/denoising_utils.py:666-692).  `t`, `p2w_table`, `var_table`: the step's time levels and the two schedule tables - the
    gathers and the reciprocal happen inside the kernel.  Returns (loss, scalars[4], residual)."""

    @staticmethod
    def forward(ctx, x0_pred, x0, f_s, t, p2w_table, var_table, c_data, c_residual, inv_h0, inv_h1, lib):
        pred = x0_pred.contiguous()
        B, C, P, _ = pred.shape
        dev = pred.device
        res = torch.empty(B, P * P, 3, dtype=torch.float32, device=dev)
        grad = torch.empty_like(pred)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
        lib.check(lib.pidm_darcy_loss_fwd_bwd_t(ptr(x0), ptr(pred), ptr(f_s), ptr(t), ptr(p2w_table), ptr(var_table),
                                                float(c_data), float(c_residual), inv_h0, inv_h1, ptr(res), ptr(grad), ptr(out),
                                                ptr(ws), B, P, stream_ptr(dev)), 'pidm_darcy_loss_fwd_bwd_t')
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(out, res)
        return out[0].clone(), out, res

    @staticmethod
    def backward(ctx, g_loss, _g_out, _g_res):
        (grad,) = ctx.saved_tensors
        return grad * g_loss, None, None, None, None, None, None, None, None, None, None


class _EarlyStepFn(torch.autograd.Function):
    """A whole single-UNet-call training step as ONE autograd node with an EARLY backward: UNet forward, the fused loss kernel
    (loss + d loss/d x0_pred) AND the UNet backward pass are enqueued inside `forward`; `loss.backward()` later only scales the
    staged parameter gradients by its upstream gradient and attaches them as `p.grad`.

    Why: the reference API returns the tracked loss terms as python floats (`.item()`, src/denoising_utils.py:681,688,699,707), i.e.
    one host synchronisation per step right after the loss kernel.  With the backward pass enqueued BEFORE that synchronisation (and
    the 16-byte scalar copy ahead of it in stream order) the GPU keeps working through it while the host wakes up, runs
    `optimizer.zero_grad()` and the autograd dispatch (0.5 ms of idle GPU per Darcy step otherwise, bench.py `eager_scalars` /
    `dropin_main_py`).  The gradients are the same numbers: backward is linear in the upstream gradient, and the staging buffer is
    private, so nothing the caller can observe changes before `backward()` - `p.grad` of an earlier step stay intact until then
    (main.py calls `zero_grad()` AFTER `model_estimation_loss`, and accumulation without `zero_grad()` still adds).  Used only when
    it pays and is safe: python-float scalars, `model.training`, gradients enabled, no data-parallel exchange attached, one UNet
    call per step (DenoisingDiffusion._early_backward_engine).

    loss_call(pred) runs the fused loss kernel and returns (grad wrt pred, device scalars with the loss in [0])."""

    @staticmethod
    def forward(ctx, anchor, eng, diffusion, x_bxyc, t, loss_call):
        eng.tape_generation += 1
        ctx.generation = eng.tape_generation
        pred = eng.forward(x_bxyc, t, training=True, early=True)
        grad, out = loss_call(pred)
        # the loss terms start their way to the host NOW, ahead of the backward pass in stream order: the host will wait for this
        # copy's event, not for the stream
        ctx.fetch = diffusion._start_scalar_fetch(out) if out.is_cuda else None
        loss_t = out[0].clone()           # (before the backward pass in stream order: `loss.item()` then does not wait for it)
        # the UNet backward for an upstream gradient of 1, into the engine's private staging buffer
        eng.backward(grad, False, x_bxyc.shape[-1])
        eng.early_generation = ctx.generation
        eng.tape_busy = False
        ctx.eng = eng
        diffusion._early_fetch = ctx.fetch
        ctx.mark_non_differentiable(out)
        return loss_t, out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        eng = ctx.eng
        if eng.early_generation != ctx.generation:
            from ._lib import PidmError
            raise PidmError("the staged gradients of this step were overwritten by a later training-mode forward of the same model "
                            "(only the latest step can be differentiated)")
        first = eng.params[0]
        aliased = first.grad is not None and first.grad.data_ptr() == eng.grad_views[0].data_ptr()
        g = g_loss.to(eng.early_grad.dtype)
        if aliased:
            eng.flat_grad.add_(eng.early_grad * g)        # a second backward without zero_grad: torch semantics are accumulation
        else:
            torch.mul(eng.early_grad, g, out=eng.flat_grad)
        eng.early_generation = -1
        eng.backward_calls += 1 if not aliased else 2
        n_plain = len(eng.params) - eng.n_cond
        for i, (p, gv) in enumerate(zip(eng.params, eng.grad_views)):
            if not p.requires_grad or i >= n_plain:
                continue                                   # conditioning branch: not used by this step, grads stay None
            if p.grad is None:
                p.grad = gv
            elif p.grad.data_ptr() != gv.data_ptr():
                p.grad.add_(eng.early_views[i] * g)
        return None, None, None, None, None, None


class DenoisingDiffusion(nn.Module):
    """Drop-in for reference DenoisingDiffusion (src/denoising_utils.py:308-788)."""

    def __init__(self, n_steps, device, residual_grad_guidance=False, lib=None):
        super().__init__()
        self.n_steps = n_steps
        self.device = device
        self.diff_dict = self.create_diff_dict()
        self.residual_grad_guidance = residual_grad_guidance
        self._lib = lib
        # data parallelism (parallel.GradientExchange sets these): number of ranks that average their gradients
        self.data_parallel_world = 1
        self.data_parallel_group = None
        # False (default): model_estimation_loss returns python floats like the reference (one host sync per step).
        # True: it returns DeferredFloat objects that synchronise when first used (a loop that logs every N iterations, like
        # main.py:167-175, then never stalls the host in between) - bench.py and main_dp.py run this way
        self.deferred_scalars = False
        self._scalar_ring, self._scalar_slot = [], 0
        self._early_fetch = None

    def _host_scalars(self, scalars, idx_list):
        """floats (or DeferredFloats) for the entries `idx_list` (index tuples) of the device tensor `scalars`"""
        if not (self.deferred_scalars and scalars.is_cuda):
            s = scalars.tolist()  # single D2H sync
            out = []
            for idx in idx_list:
                v = s
                for i in idx:
                    v = v[i]
                out.append(v)
            return out
        fetch = self._start_scalar_fetch(scalars)
        return [DeferredFloat(fetch, tuple(idx)) for idx in idx_list]

    def _start_scalar_fetch(self, scalars):
        """Enqueues the copy of a small device tensor into pinned host memory and records an event behind it; `.values()` of the
        returned object waits for THAT event only (not for work enqueued on the stream afterwards)."""
        if not self._scalar_ring:
            self._scalar_ring = [[torch.empty(16, dtype=torch.float32).pin_memory(), None] for _ in range(64)]
        slot = self._scalar_ring[self._scalar_slot]
        self._scalar_slot = (self._scalar_slot + 1) % len(self._scalar_ring)
        if slot[1] is not None:
            slot[1].values()          # 64 steps old: long complete; keeps its values before the buffer is reused
        host = slot[0][:scalars.numel()].view(scalars.shape)
        host.copy_(scalars.detach(), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = _ScalarFetch(host, ev)
        return slot[1]

    @property
    def lib(self):
        if self._lib is None:
            self._lib = get_lib()
        return self._lib

    # ---- schedule (fp32 op order of src/denoising_utils.py:315-370 preserved: the tables are bit-exact) ----
    def make_beta_schedule(self, schedule='linear', n_timesteps=1000, start=1e-5, end=1e-2):
        if schedule == 'linear':
            return torch.linspace(start, end, n_timesteps)
        if schedule == 'quad':
            return torch.linspace(start ** 0.5, end ** 0.5, n_timesteps) ** 2
        if schedule == 'sigmoid':
            return torch.sigmoid(torch.linspace(-6, 6, n_timesteps)) * (end - start) + start
        if schedule == 'cosine':
            s = 0.008
            x = torch.linspace(0, n_timesteps, n_timesteps + 1)
            ac = torch.cos(((x / n_timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
            ac = ac / ac[0]
            return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
        raise ValueError(schedule)

    def create_diff_dict(self):
        d = self._host_schedule(self.n_steps)
        self._host_tables = {k: v.clone() for k, v in d.items()}  # python-float access without device syncs
        return {k: v.to(self.device) for k, v in d.items()}

    @staticmethod
    def schedule_tables(n_steps, device):
        """The schedule tables alone (no DenoisingDiffusion object): what the toy study's create_diff_dict returns
        (src/denoising_toy_utils.py:43-83 - the same tables as :315-370 of src/denoising_utils.py)."""
        return {k: v.to(device) for k, v in DenoisingDiffusion._host_schedule(n_steps).items()}

    @staticmethod
    def _host_schedule(n_steps):
        b = DenoisingDiffusion.make_beta_schedule(None, schedule='cosine', n_timesteps=n_steps, start=1e-5, end=1e-2)
        d = {'betas': b}
        d['alphas'] = 1. - b
        d['sqrt_recip_alphas'] = torch.sqrt(1. / d['alphas'])
        ap = torch.cumprod(d['alphas'], 0)
        d['alphas_prod'] = ap
        d['alphas_prod_p'] = torch.cat([torch.ones(1), ap[:-1]], 0)
        d['alphas_bar_sqrt'] = torch.sqrt(ap)
        d['sqrt_recip_alphas_cumprod'] = torch.sqrt(1. / ap)
        d['sqrt_recipm1_alphas_cumprod'] = torch.sqrt(1. / ap - 1)
        d['one_minus_alphas_bar_log'] = torch.log(1 - ap)
        d['one_minus_alphas_bar_sqrt'] = torch.sqrt(1 - ap)
        app = F.pad(ap[:-1], (1, 0), value=1.)
        d['alphas_prod_prev'] = app
        d['posterior_mean_coef1'] = b * torch.sqrt(app) / (1. - ap)
        d['posterior_mean_coef2'] = (1. - app) * torch.sqrt(d['alphas']) / (1. - ap)
        d['noise_mean_coeff'] = torch.sqrt(1. / d['alphas']) * (1. - d['alphas']) / torch.sqrt(1. - ap)
        pv = b * (1. - app) / (1. - ap)
        d['posterior_variance'] = pv
        pvc = pv.clone()
        pvc[0] = pv[1]
        d['posterior_variance_clipped'] = pvc
        d['posterior_log_variance_clipped'] = torch.log(pvc)
        snr = ap / (1. - ap)
        d['p2_loss_weight'] = torch.minimum(snr, torch.ones_like(snr) * 5.0)
        return d

    def q_sample(self, x_0, t, alphas_bar_sqrt, one_minus_alphas_bar_sqrt, noise=None):
        if noise is None:
            noise = torch.randn_like(x_0)
        return extract(alphas_bar_sqrt, t, x_0) * x_0 + extract(one_minus_alphas_bar_sqrt, t, x_0) * noise

    def plot_diffusion(self, dataset, alphas_bar_sqrt, one_minus_alphas_bar_sqrt):
        """Scatter plots of q(x_t) for t = 0, 10, ..., 90 of a 2-D dataset (src/denoising_utils.py:380-386; toy visualisation,
        host-side; needs matplotlib)."""
        import matplotlib.pyplot as plt
        fig, axs = plt.subplots(1, 10, figsize=(18, 2))
        for i in range(10):
            q_i = self.q_sample(dataset, torch.tensor([i * 10]), alphas_bar_sqrt, one_minus_alphas_bar_sqrt)
            axs[i].scatter(q_i[:, 0], q_i[:, 1], s=10)
            axs[i].set_axis_off()
            axs[i].set_title('$q(\\mathbf{x}_{' + str(i * 10) + '})$', fontsize=10)
        plt.show()

    def normal_kl(self, mean1, logvar1, mean2, logvar2):
        """KL(N(mean1, e^logvar1) || N(mean2, e^logvar2)), elementwise (src/denoising_utils.py:547-552)."""
        return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * torch.exp(-logvar2))

    def gaussian_log_likelihood(self, x, means, variance):
        return -0.5 * ((x - means) ** 2) / variance

    def predict_start_from_noise(self, x_t, t, noise):
        return (extract(self.diff_dict['sqrt_recip_alphas_cumprod'], t, x_t) * x_t -
                extract(self.diff_dict['sqrt_recipm1_alphas_cumprod'], t, x_t) * noise)

    def predict_noise_from_start(self, x_t, t, x0):
        return (extract(self.diff_dict['sqrt_recip_alphas_cumprod'], t, x_t) * x_t - x0) / \
            extract(self.diff_dict['sqrt_recipm1_alphas_cumprod'], t, x_t)

    def loss_variational(self, output, x_0, x_t, t, base_2=False):
        """Variational-bound term with the model variance fixed to the clipped posterior variance (src/denoising_utils.py:576-614):
        KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) per sample for t > 0, -log p(x_0|x_1) at t == 0; mean over the batch.  Not used
        by main.py / sample.py (the PIDM loss is model_estimation_loss); plain tensor algebra on the schedule tables."""
        batch_size = x_0.shape[0]
        true_mean = (extract(self.diff_dict['posterior_mean_coef1'], t, x_t) * x_0 +
                     extract(self.diff_dict['posterior_mean_coef2'], t, x_t) * x_t)
        true_var = extract(self.diff_dict['posterior_variance_clipped'], t, x_t)
        model_var, model_mean = true_var, output
        kl = self.normal_kl(true_mean, torch.log(true_var), model_mean, torch.log(model_var))
        kl = torch.mean(kl.view(batch_size, -1), dim=1)
        if base_2:
            kl = kl / np.log(2.)
        log_likelihood = self.gaussian_log_likelihood(x_0, means=model_mean, variance=model_var)
        log_likelihood = torch.mean(log_likelihood.view(batch_size, -1), dim=1)
        if base_2:
            log_likelihood = log_likelihood / np.log(2.)
        assert not log_likelihood.isnan().any(), 'Log likelihood is nan.'
        assert not log_likelihood.isinf().any(), 'Log likelihood is inf.'
        loss = torch.where(t == 0, -1. * log_likelihood, kl)
        return loss.mean(-1)

    def predict_noise_from_mean(self, x_t, t, mean_t):
        return (extract(self.diff_dict['sqrt_recip_alphas'], t, mean_t) * x_t - mean_t) / \
            extract(self.diff_dict['noise_mean_coeff'], t, mean_t)

    # ---- training loss (src/denoising_utils.py:616-710) ---------------------------------------------------------
    def _darcy_fast_path_ok(self, residual_func, c_ineq, lambda_opt, x):
        return (isinstance(residual_func, ResidualsDarcy) and not residual_func.residual_grad_guidance and c_ineq <= 0.
                and lambda_opt <= 0. and x.dtype == torch.float32 and (x.is_cuda or self._lib is not None))

    def _mech_fast_path_ok(self, residual_func, x):
        from .residuals_mechanics_K import ResidualsMechanics
        return (isinstance(residual_func, ResidualsMechanics) and x.dtype == torch.float32
                and (x.is_cuda or self._lib is not None))

    def model_estimation_loss(self, input, residual_func=None, c_data=1., c_residual=0., c_ineq=0., lambda_opt=0.):
        batch_size = len(input)
        t = torch.randint(0, self.n_steps, size=(batch_size,), device=input.device)   # RNG draw #1 (:625)
        if residual_func.gov_eqs == 'darcy':
            x_0 = input
            conditioning = bcs = None
        elif residual_func.gov_eqs == 'mechanics':
            conditioning, x_0, bcs = torch.tensor_split(input, (3, 6), dim=1)
        else:
            raise ValueError('Unknown governing equations.')
        e = torch.randn_like(x_0)                                                       # RNG draw #2 (:636)

        if self._darcy_fast_path_ok(residual_func, c_ineq, lambda_opt, x_0):
            return self._darcy_step(x_0, e, t, residual_func, c_data, c_residual)
        if residual_func.gov_eqs == 'mechanics' and self._mech_fast_path_ok(residual_func, x_0):
            return self._mech_step(x_0, conditioning, bcs, e, t, residual_func, c_data, c_residual, c_ineq, lambda_opt)

        a = extract(self.diff_dict['alphas_bar_sqrt'], t, x_0)
        am1 = extract(self.diff_dict['one_minus_alphas_bar_sqrt'], t, x_0)
        x = x_0 * a + e * am1
        if residual_func.gov_eqs == 'mechanics':
            x = torch.cat((x, conditioning), dim=1)
        model_input = (image_to_b_xy_c(x), t)
        return_inequality = c_ineq > 0.
        return_optimizer = lambda_opt > 0. or residual_func.gov_eqs == 'mechanics'
        if residual_func.gov_eqs == 'darcy':
            residual_input = (model_input,)
        else:
            vf = conditioning[:, 0, 0, 0]
            residual_input = (model_input, bcs, vf, x_0)
        out_dict = residual_func.compute_residual(residual_input, reduce='per-batch', return_model_out=True,
                                                  return_optimizer=return_optimizer, return_inequality=return_inequality,
                                                  ddim_func=self.ddim_sample_x0)
        residual, output = out_dict['residual'], out_dict['model_out']
        if output.dim() == 3:
            output = b_xy_c_to_image(output)
        loss = ((x_0 - output) ** 2).reshape(batch_size, -1).mean(dim=1)
        loss = (loss * extract(self.diff_dict['p2_loss_weight'], t, loss)).mean()
        data_loss = c_data * loss
        data_loss_track = data_loss.item()
        loss = data_loss
        var = extract(self.diff_dict['posterior_variance_clipped'], t, residual)
        residual_loss_track = residual.abs().mean().item()
        loss = loss + (c_residual * 0.5 * residual ** 2 / var).mean()
        ineq_loss_track = 0.
        if return_inequality:
            # NOTE the reference broadcasts [B] against [B,1] -> [B,B] here (src/denoising_utils.py:697); reproduced
            ineq = out_dict['inequality']
            ineq_loss_track = ineq.mean().item()
            loss = loss + (c_ineq * 0.5 * ineq ** 2 / var).mean()
        opt_loss_track = 0.
        if return_optimizer:
            opt_loss_track = out_dict['optimizer'].mean().item()
            loss = loss + (lambda_opt * out_dict['optimizer']).mean()
        return loss, data_loss_track, residual_loss_track, ineq_loss_track, opt_loss_track

    def _darcy_step(self, x_0, e, t, residual_func, c_data, c_residual):
        """Native hot path: q-sample -> UNet -> fused residual+loss(+grad).  One host sync for the 2 floats the
        reference API returns (it forces two .item() calls, src/denoising_utils.py:681,688)."""
        lib = residual_func.lib
        B, C, P, _ = x_0.shape
        dev = x_0.device
        dd = self._kernel_tables(dev)
        x_0 = x_0.contiguous()
        t = t.to(device=dev, dtype=torch.int64).contiguous()      # drawn by model_estimation_loss in [0, n_steps)
        if residual_func._f_s_flat.device != dev:
            residual_func._f_s_flat = residual_func._f_s_flat.to(dev)
        xt = torch.empty(B, P * P, C, dtype=torch.float32, device=dev)
        # the extract() gathers of the schedule tables by t happen inside the kernels (no indexing / reciprocal launches)
        lib.check(lib.pidm_qsample_nhwc_t(ptr(x_0), ptr(e.contiguous()), ptr(t), ptr(dd['alphas_bar_sqrt']),
                                          ptr(dd['one_minus_alphas_bar_sqrt']), ptr(xt), B, C, P * P, stream_ptr(dev)),
                  'pidm_qsample_nhwc_t')
        eng = self._early_backward_engine(residual_func, xt.shape[-1], P) if not residual_func.use_ddim_x0 else None
        if eng is not None:
            # python-float loss terms (one host sync per step): the backward pass is enqueued before that sync
            f_s = residual_func._f_s_flat

            def loss_call(pred):
                res = torch.empty(B, P * P, 3, dtype=torch.float32, device=dev)
                grad = torch.empty_like(pred)
                out = torch.empty(4, dtype=torch.float32, device=dev)
                ws = torch.empty(lib.pidm_darcy_loss_ws(B, P), dtype=torch.uint8, device=dev)
                lib.check(lib.pidm_darcy_loss_fwd_bwd_t(ptr(x_0), ptr(pred), ptr(f_s), ptr(t), ptr(dd['p2_loss_weight']),
                                                        ptr(dd['posterior_variance_clipped']), float(c_data), float(c_residual),
                                                        residual_func.inv_h0, residual_func.inv_h1, ptr(res), ptr(grad), ptr(out), ptr(ws),
                                                        B, P, stream_ptr(dev)), 'pidm_darcy_loss_fwd_bwd_t')
                return grad, out
            loss, scalars = _EarlyStepFn.apply(eng.params[0], eng, self, xt, t, loss_call)
            fetch, self._early_fetch = self._early_fetch, None
            if fetch is not None:
                v = fetch.values()                    # waits for the loss kernel + the 16-byte copy; the backward pass keeps running
                return loss, v[1], v[2], 0., 0.
            d, r = self._host_scalars(scalars, [(1,), (2,)])
            return loss, d, r, 0., 0.
        if residual_func._f_s_flat.device != dev:
            residual_func._f_s_flat = residual_func._f_s_flat.to(dev)
        args = (residual_func._f_s_flat, t, dd['p2_loss_weight'], dd['posterior_variance_clipped'])
        geo = (residual_func.inv_h0, residual_func.inv_h1, lib)
        if residual_func.use_ddim_x0:
            # x0_estimation 'sample' (src/residuals_darcy.py:127-128): the data term sees model(x_t, t), the residual term
            # model(x_t, 0) - the same fused kernel once per tensor, with the other term's weight set to zero
            x0_pred, model_out = self.ddim_sample_x0(xt, t, residual_func.model, xt.shape, residual_func.ddim_steps, 0.)
            l_data, s_data, _ = _DarcyPidmLossFn.apply(model_out, x_0, *args, c_data, 0., *geo)
            l_res, s_res, _res = _DarcyPidmLossFn.apply(x0_pred, x_0, *args, 0., c_residual, *geo)
            d, r = self._host_scalars(torch.stack((s_data, s_res)), [(0, 1), (1, 2)])
            return l_data + l_res, d, r, 0., 0.
        x0_pred = residual_func.model(xt, t)
        loss, scalars, _res = _DarcyPidmLossFn.apply(x0_pred, x_0, *args, c_data, c_residual, *geo)
        d, r = self._host_scalars(scalars, [(1,), (2,)])
        return loss, d, r, 0., 0.

    def _kernel_tables(self, dev):
        """The schedule tables the kernels index by `t` through raw pointers (q-sample, fused Darcy loss): float32, contiguous
        and on the batch's device - torch indexing used to raise on a device mismatch, a raw pointer would read foreign memory.
        Tables built on another device are moved once (cached per device), as `_f_s_flat` is."""
        cache = self.__dict__.setdefault('_pidm_tables', {})
        got = cache.get(dev)
        if got is None or got[0] is not self.diff_dict:
            tabs = {}
            for k in ('alphas_bar_sqrt', 'one_minus_alphas_bar_sqrt', 'p2_loss_weight', 'posterior_variance_clipped'):
                v = self.diff_dict[k]
                if v.device != dev or v.dtype != torch.float32 or not v.is_contiguous():
                    v = v.to(device=dev, dtype=torch.float32).contiguous()
                if v.dim() != 1 or v.numel() != self.n_steps:
                    raise ValueError(f'schedule table {k!r} has shape {tuple(v.shape)}, expected ({self.n_steps},)')
                tabs[k] = v
            cache[dev] = got = (self.diff_dict, tabs)
        return got[1]

    def _early_backward_engine(self, residual_func, channels, image_size):
        """The engine to run the step's backward pass at forward time on (see _EarlyStepFn), or None when that is not applicable."""
        from ._engine import get_engine
        from .unet_model import Unet3D
        model = residual_func.model
        first = next(model.parameters())
        if (self.deferred_scalars or not torch.is_grad_enabled() or os.environ.get('PIDM_EARLY_BACKWARD') == '0'
                or type(model) is not Unet3D or not model.training or model.self_condition
                or not (first.is_cuda or self._lib is not None) or model._forward_hooks or model._forward_pre_hooks
                or channels != model.channels or getattr(model, '_pidm_tape_slot', None)):
            return None
        eng = get_engine(model, image_size, model._pidm_lib)
        if getattr(eng, '_exchange_owner', None) is not None or eng.tape_busy or not all(p.requires_grad for p in eng.params):
            return None
        return eng

    def _mech_step(self, x_0, conditioning, bcs, e, t, residual_func, c_data, c_residual, c_ineq, lambda_opt):
        """Mechanics configuration (main.py:102-109,139): the loss algebra of src/denoising_utils.py:666-708 on top of the
        matrix-free residual runs as ONE fused kernel that also returns d loss / d x0_pred (csrc/k_mech.hip mech_loss_kernel);
        one host sync for the four floats the reference API returns."""
        from .residuals_mechanics_K import _MechLossFn, resize_image
        lib = residual_func.lib
        dd = self.diff_dict
        a = extract(dd['alphas_bar_sqrt'], t, x_0)
        am1 = extract(dd['one_minus_alphas_bar_sqrt'], t, x_0)
        x = torch.cat((x_0 * a + e * am1, conditioning), dim=1)
        P = residual_func.pixels_per_dim
        net_in = torch.cat((resize_image(x, P, lib), resize_image(bcs, P, lib)), dim=1)          # 10 channels
        vf = conditioning[:, 0, 0, 0].contiguous()
        p2w = dd['p2_loss_weight'][t].contiguous()
        inv_var = (1.0 / dd['posterior_variance_clipped'][t]).contiguous()
        if residual_func.stiffs.kloc_dev.device != x_0.device:
            residual_func.stiffs.to(x_0.device)
        ivs = None
        if c_ineq > 0. and self.data_parallel_world > 1:
            # the inequality term couples the samples of a batch through its [B,B] broadcast (:697).  Its only cross-sample
            # factor that involves other ranks' data is sum_i 1/var_i, which depends on t alone: one scalar all-reduce (issued
            # before the UNet forward, it never waits) makes the rank-averaged loss and gradients exactly the global-batch ones
            import torch.distributed as dist
            ivs = inv_var.sum().reshape(1)
            dist.all_reduce(ivs, op=dist.ReduceOp.SUM, group=self.data_parallel_group)
            ivs = ivs / float(self.data_parallel_world)
        fixed = (x_0.contiguous(), bcs.contiguous(), vf, p2w, inv_var, ivs)
        if residual_func.use_ddim_x0:
            x0_pred, model_out = self.ddim_sample_x0(net_in, t, residual_func.model, x.shape, residual_func.ddim_steps, 0.,
                                                     gov_eqs='mechanics')
            l_data, s_data = _MechLossFn.apply(model_out, *fixed, c_data, 0., 0., 0., residual_func.stiffs, lib)
            l_res, s_res = _MechLossFn.apply(x0_pred, *fixed, 0., c_residual, c_ineq, lambda_opt, residual_func.stiffs, lib)
            d, r, q, o = self._host_scalars(torch.stack((s_data, s_res)), [(0, 1), (1, 2), (1, 3), (1, 4)])
            return l_data + l_res, d, r, q, o
        eng = self._early_backward_engine(residual_func, net_in.shape[1], P) if not net_in.requires_grad else None
        if eng is not None:
            # python-float loss terms (one host sync per step): the backward pass is enqueued before that sync (_EarlyStepFn)
            stiffs = residual_func.stiffs
            tgt, bcs_c, vf_c = fixed[0].float(), fixed[1].float(), vf.float()
            nel = P
            x_bxyc = net_in.permute(0, 2, 3, 1).reshape(net_in.shape[0], P * P, net_in.shape[1]).contiguous().float()
            tt = t.to(dtype=torch.int64).contiguous()

            def loss_call(pred):
                Bm = pred.shape[0]
                grad = torch.empty_like(pred)
                out = torch.empty(8, dtype=torch.float32, device=pred.device)
                ws = torch.empty(lib.pidm_mech_loss_ws(Bm), dtype=torch.uint8, device=pred.device)
                lib.check(lib.pidm_mech_loss_fwd_bwd(ptr(pred), ptr(tgt), ptr(bcs_c), ptr(vf_c), ptr(p2w), ptr(inv_var), ptr(ivs),
                                                     float(c_data), float(c_residual), float(c_ineq), float(lambda_opt),
                                                     ptr(stiffs.kloc_dev), stiffs.kloc_stride, ptr(stiffs.elem_dofs32),
                                                     ptr(stiffs.dof_elems32), nel, ptr(grad), ptr(out), ptr(ws), Bm,
                                                     stream_ptr(pred.device)), 'pidm_mech_loss_fwd_bwd')
                return grad, out
            loss, scalars = _EarlyStepFn.apply(eng.params[0], eng, self, x_bxyc, tt, loss_call)
            fetch, self._early_fetch = self._early_fetch, None
            if fetch is not None:
                v = fetch.values()
                return loss, v[1], v[2], v[3], v[4]
            d, r, q, o = self._host_scalars(scalars, [(1,), (2,), (3,), (4,)])
            return loss, d, r, q, o
        x0_pred = residual_func.model(net_in, t)
        loss, scalars = _MechLossFn.apply(x0_pred, *fixed, c_data, c_residual, c_ineq, lambda_opt, residual_func.stiffs, lib)
        d, r, q, o = self._host_scalars(scalars, [(1,), (2,), (3,), (4,)])
        return loss, d, r, q, o

    # ---- sampling (src/denoising_utils.py:388-545) ---------------------------------------------------------------
    def p_sample(self, x, conditioning_input, t, save_output=False, surpress_noise=False, use_dynamic_threshold=False,
                 residual_func=None, eval_residuals=False, return_optimizer=False, return_inequality=False,
                 residual_correction=False, correction_mode='none'):
        assert correction_mode in ['x0', 'xt'] or not residual_correction, 'Correction mode unknown or not given.'
        x_init = x.detach()
        if conditioning_input is not None:
            conditioning, bcs, solution = conditioning_input
            x = torch.cat((x, conditioning), dim=1)
        batch_size = len(x)
        t_int = int(t)
        tt = torch.full((batch_size,), t_int, device=x.device, dtype=torch.long)
        model_input = (image_to_b_xy_c(x), tt)
        if residual_func.gov_eqs == 'darcy':
            residual_input = (model_input,)
            sample = True
        else:
            vf = conditioning[:, 0, 0, 0]
            residual_input = (model_input, bcs, vf, solution)
            sample = t_int == 0
        # the reference builds (and discards) an autograd graph here (:492-493); nothing downstream needs it
        with torch.no_grad():
            out_dict = residual_func.compute_residual(residual_input, reduce='per-batch', return_model_out=True,
                                                      return_optimizer=return_optimizer, return_inequality=return_inequality,
                                                      sample=sample, ddim_func=self.ddim_sample_x0)
            model_out, residual = out_dict['model_out'], out_dict['residual']
            if model_out.dim() == 3:
                model_out = generalized_b_xy_c_to_image(model_out)
            if residual_correction and correction_mode == 'x0':      # CoCoGen (:434-436)
                model_out, residual = residual_func.residual_correction(generalized_image_to_b_xy_c(model_out).contiguous())
                model_out = generalized_b_xy_c_to_image(model_out)
            model_intermediate = model_out.clone() if save_output else None
            ht = self._host_tables
            c1, c2 = float(ht['posterior_mean_coef1'][t_int]), float(ht['posterior_mean_coef2'][t_int])
            sigma = float(ht['betas'][t_int].sqrt())
            z = torch.randn_like(x_init)
            if surpress_noise and t_int == 0:
                sigma = 0.0
            lib = self.lib
            x0p = model_out.contiguous()
            xi = x_init.contiguous()
            out = torch.empty_like(xi)
            lib.check(lib.pidm_psample_update(ptr(x0p), ptr(xi), ptr(z), c1, c2, sigma, ptr(out), xi.numel(),
                                              stream_ptr(xi.device)), 'pidm_psample_update')
            if residual_correction and correction_mode == 'xt':      # CoCoGen (:457-459)
                out, residual = residual_func.residual_correction(generalized_image_to_b_xy_c(out).contiguous())
                out = generalized_b_xy_c_to_image(out).contiguous()
            if use_dynamic_threshold:
                # off in main.py / sample.py (src/denoising_utils.py:461-473): per-sample 0.9-quantile of |x|, at least 1;
                # a sort per step, kept on torch (not on the accelerated path)
                s_thr = torch.quantile(out.reshape(batch_size, -1).abs().float(), 0.9, dim=-1).clamp_(min=1.0)
                s_thr = right_pad_dims_to(out, s_thr)
                out = torch.maximum(torch.minimum(out, s_thr), -s_thr) / s_thr
        if t_int == 0 and eval_residuals:
            aux_out = {'residual': residual}
            if return_optimizer:
                aux_out['optimized_quant'] = out_dict['optimizer']
            if return_inequality:
                aux_out['inequality_quant'] = out_dict['inequality']
            if residual_func.gov_eqs == 'mechanics' and residual_func.topopt_eval:
                for k in ('rel_CE_error_full_batch', 'vf_error_full_batch', 'fm_error_full_batch'):
                    aux_out[k] = out_dict[k]
            return (out, model_intermediate), aux_out
        return (out, model_intermediate), None

    def p_sample_loop(self, conditioning_input, shape, save_output=False, surpress_noise=True, use_dynamic_threshold=False,
                      residual_func=None, eval_residuals=False, return_optimizer=False, return_inequality=False,
                      M_correction=0, N_correction=0, correction_mode='none', keep_history=True):
        """`keep_history=False` (extension) skips the two blocking D2H copies per step the reference performs
        (src/denoising_utils.py:531-532) and returns only the final state in the lists."""
        cur_x = torch.randn(shape, device=self.diff_dict['alphas'].device)
        x_seq = [cur_x.detach().cpu()] if keep_history else []
        interm_imgs = [torch.zeros(shape)] if (save_output and keep_history) else []
        output = None
        # nothing inside this loop writes the parameters: the engine packs / splits the weights once, not once per step
        from ._engine import frozen_weights
        net = getattr(residual_func, "model", None)
        with (frozen_weights(net) if isinstance(net, torch.nn.Module) else contextlib.nullcontext()):
            for i in reversed(range(self.n_steps)):
                residual_correction = False
                if i < N_correction:            # CoCoGen correction inside the last N steps (:520-523)
                    residual_correction = True
                    eval_residuals = True
                output = self.p_sample(cur_x.detach(), conditioning_input, i, save_output, surpress_noise, use_dynamic_threshold,
                                       residual_func=residual_func, eval_residuals=eval_residuals,
                                       return_optimizer=return_optimizer, return_inequality=return_inequality,
                                       residual_correction=residual_correction, correction_mode=correction_mode)
                cur_x, interm_img = output[0]
                if keep_history:
                    x_seq.append(cur_x.detach().cpu())
                    interm_imgs.append(interm_img.detach().cpu())
        for i in range(M_correction):       # CoCoGen post-correction (:535-540)
            cur_x, residual = residual_func.residual_correction(generalized_image_to_b_xy_c(cur_x).contiguous())
            cur_x = generalized_b_xy_c_to_image(cur_x).contiguous()
            if keep_history:
                x_seq.append(cur_x.detach().cpu())
            if eval_residuals and i == M_correction - 1:
                output[1]['residual'] = residual
        if not keep_history:
            x_seq.append(cur_x.detach())
            if save_output:
                interm_imgs.append(interm_img.detach())
        if eval_residuals:
            return (x_seq, interm_imgs), output[1]
        return x_seq, interm_imgs

    def ddim_sample_x0(self, xt, t, model, shape, reduced_n_steps, ddim_sampling_eta, gov_eqs=None, self_cond=None):
        """Sample estimation (src/denoising_utils.py:712-788).  The reference walks ddim_steps + 2 time levels from t down to 0
        but never updates `model_input`: every model call sees the same x_t, the value it returns is
        the LAST call's output (at time 0) and `model_out` is the FIRST call's (at time t); the intermediate calls only
        feed a `cur_x` that the final assignment overwrites.  So for any ddim_steps the result is two UNet evaluations;
        the ddim_steps + 1 noise tensors the reference draws along the way are drawn here too (and discarded), so a seeded
        run consumes the RNG identically."""
        if reduced_n_steps < 0:
            raise ValueError('ddim_steps must be >= 0')
        batch = shape[0]
        if len(t) == 1:
            t = torch.ones(batch, device=xt.device, dtype=torch.long) * t
        # two live activation tapes: the call at (x_t, t) records on engine slot 0, the call at (x_t, 0) on slot 1
        # (_engine.unet_apply); named slots, so tapes of losses that are never differentiated (validation) do not pile up
        from ._engine import frozen_weights
        try:
            with frozen_weights(model):         # (matters for the no-grad case: the second evaluation re-uses the packed weights)
                model._pidm_tape_slot = 0
                model_out = model(xt, t)
                model._pidm_tape_slot = 1
                x0_pred = model(xt, torch.zeros_like(t))
        finally:
            model._pidm_tape_slot = None
        # RNG parity with :775: the same randn_like call on a tensor of the same shape AND strides as the reference's cur_x
        # (a permuted view of x_t - the CPU generator consumes differently for non-contiguous outputs); sigma is 0 for
        # eta = 0, so the values never matter
        xt_img = generalized_b_xy_c_to_image(xt) if xt.dim() == 3 else xt
        cur_x = xt_img[:, :3] if gov_eqs == 'mechanics' else xt_img
        for _ in range(reduced_n_steps + 1):
            torch.randn_like(cur_x)
        return x0_pred, model_out
