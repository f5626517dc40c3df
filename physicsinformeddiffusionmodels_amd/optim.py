"""Fused optimizer step for the gfx950 engine: global-norm clip + Adam in two HIP launches over flat buffers.

Replaces `torch.nn.utils.clip_grad_norm_(model.parameters(), 1.)` + `torch.optim.Adam.step()` (reference main.py:165-166,
SURVEY 8(f) rank 1).  The engine already produces every gradient in ONE flat buffer in its canonical parameter order
(`_engine.UnetEngine.flat_grad`); `flatten_parameters` re-homes the parameters themselves into one flat buffer in the same
order (the `nn.Parameter`s stay what they were - `state_dict`, checkpoints, EMA are untouched, they are views now), and both
Adam moments are flat too, so the update is a single stream over 7 x 4 bytes per parameter.

Semantics = torch.optim.Adam(lr, betas, eps, weight_decay=0, amsgrad=False) preceded by clip_grad_norm_(max_norm):
parameters whose gradient is None (the ones the forward never reads) are not updated, as in torch.
No CPU fallback: the buffers must live on an MI355X and `libpidm_hip.so` must be built.
"""
from __future__ import annotations

import torch

from ._engine import get_engine
from ._lib import PidmError, PidmLib, ptr, stream_ptr, vp


def flatten_parameters(model, image_size: int = 64, lib: PidmLib | None = None) -> torch.Tensor:
    """Move the engine-used parameters of `model` into one flat fp32 buffer (engine order) and return it.  Idempotent.
    Call after `model.to(device)`; a later `.to()` / `.float()` that re-allocates parameters undoes it (detected)."""
    eng = get_engine(model, image_size, lib)
    flat = model.__dict__.get("_pidm_flat_params")
    if flat is not None and _is_flat(eng, flat):
        return flat
    dev = eng.params[0].device
    total = sum(eng.numels)
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    off = 0
    with torch.no_grad():
        for p, ne in zip(eng.params, eng.numels):
            view = flat[off:off + ne].view(p.shape)
            view.copy_(p.data)
            p.data = view
            off += ne
    model.__dict__["_pidm_flat_params"] = flat
    return flat


def _is_flat(eng, flat) -> bool:
    off = flat.data_ptr()
    for p, ne in zip(eng.params, eng.numels):
        if p.data_ptr() != off:
            return False
        off += 4 * ne
    return True


class FusedClipAdam:
    """clip_grad_norm_(max_norm) + Adam.step() for a `Unet3D` driven by the gfx950 engine.

        opt = FusedClipAdam(model, lr=1e-4, max_norm=1.0)
        loss.backward(); [allreduce_gradients(model, world)]; opt.step(); opt.zero_grad()

    `step()` returns the pre-clip global gradient norm as a device scalar (what clip_grad_norm_ returns)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=None, image_size: int = 64,
                 lib: PidmLib | None = None, ema=None, ema_start: int = -1):
        """ema (optional `denoising_utils.EMA`, already `register`ed): its update (main.py:178-179) is folded into the Adam
        kernel for every step whose 0-based index is > ema_start (main.py:52 uses ema_start = 1000); the `ema.update(model)`
        call main.py makes right after then only acknowledges it, so the loop body stays as it is."""
        self.model = model
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.max_norm = None if max_norm is None else float(max_norm)
        self.image_size = image_size
        self.eng = get_engine(model, image_size, lib)
        self.lib = self.eng.lib
        if any(not p.requires_grad for p in self.eng.params):
            raise PidmError("FusedClipAdam updates the whole flat parameter buffer: frozen parameters (requires_grad=False) "
                            "are not supported - use torch.optim.Adam for partially frozen models")
        self.flat = flatten_parameters(model, image_size, lib)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0
        dev = self.flat.device
        self._ws = torch.empty(self.lib.pidm_clip_adam_ws_bytes(), dtype=torch.uint8, device=dev)
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.ema, self.ema_start = ema, int(ema_start)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.model.parameters():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _flat_grad(self):
        eng = self.eng
        if eng.flat_grad is None:
            raise PidmError("FusedClipAdam.step(): no gradient yet (run a training forward + backward first)")
        n_plain = len(eng.params) - eng.n_cond
        for i, (p, g) in enumerate(zip(eng.params, eng.grad_views)):
            if i >= n_plain and p.grad is None:
                continue    # conditioning branch unused this step: the engine zero-filled its gradient slots
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                raise PidmError("FusedClipAdam.step(): p.grad does not alias the engine's flat gradient buffer "
                                "(gradients were re-assigned or only partially computed)")
        return eng.flat_grad

    @torch.no_grad()
    def step(self):
        if not _is_flat(self.eng, self.flat):
            raise PidmError("FusedClipAdam: the model's parameters do not live in the flat buffer (re-allocated after "
                            "flatten_parameters(), or the EMA weights are swapped in: call ema.restore(model) first)")
        g = self._flat_grad()
        dev = self.flat.device
        max_norm = -1.0 if self.max_norm is None else self.max_norm
        lay = None
        if self.ema is not None and self.step_count > self.ema_start:
            lay = self.ema._flat_layout(self.model)
        if lay is not None:
            if self.ema._fused_updates != 0:      # checked before step_count moves: a caller that fixes its loop retries the SAME step
                # the previous fused update was never acknowledged: the loop's `ema.update(model)` condition and this
                # optimizer's ema_start disagree (or the call is missing) - applying another one would double-count silently
                raise PidmError("FusedClipAdam.step(): the EMA update folded into the previous step was not acknowledged by "
                                "ema.update(model) - call it after every optimizer step with index > ema_start "
                                f"(ema_start={self.ema_start}), as main.py:178-179 does")
            self.step_count += 1
            self.lib.check(self.lib.pidm_clip_adam_ema_step(
                ptr(self.flat), ptr(g), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(lay[0]), self.flat.numel(), self.lr,
                self.betas[0], self.betas[1], self.eps, self.step_count, max_norm, float(self.ema.mu), ptr(self._norm), ptr(self._ws),
                stream_ptr(dev)), "pidm_clip_adam_ema_step")
            self.ema._fused_updates += 1
        else:
            self.step_count += 1
            self.lib.check(self.lib.pidm_clip_adam_step(
                ptr(self.flat), ptr(g), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.flat.numel(), self.lr, self.betas[0],
                self.betas[1], self.eps, self.step_count, max_norm, ptr(self._norm), ptr(self._ws), stream_ptr(dev)),
                "pidm_clip_adam_step")
        return self._norm[0]

    # checkpointing: flat tensors in the engine's canonical parameter order (names alongside for inspection)
    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr,
                "betas": self.betas, "eps": self.eps, "max_norm": self.max_norm, "param_names": list(self.eng.names)}

    def load_state_dict(self, sd):
        if list(sd["param_names"]) != list(self.eng.names):
            raise PidmError("FusedClipAdam.load_state_dict: parameter order mismatch")
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"].to(self.exp_avg.device))
        self.exp_avg_sq.copy_(sd["exp_avg_sq"].to(self.exp_avg_sq.device))
        self.lr, self.betas, self.eps, self.max_norm = float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"]), sd["max_norm"]
