"""Fused clip+Adam (csrc/k_optim.hip, optim.FusedClipAdam) against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam -
the two calls of the reference training loop it replaces (main.py:165-166)."""
import copy

import numpy as np
import pytest
import torch

from oracle import pidm_oracle as O
from physicsinformeddiffusionmodels_amd._lib import ptr, stream_ptr
from physicsinformeddiffusionmodels_amd.optim import FusedClipAdam, flatten_parameters
from tests.test_training_step import patched_rng, setup


@pytest.mark.parametrize("n,max_norm", [(10007, 1.0), (4096, 1e9), (777, -1.0)])
def test_clip_adam_kernel_vs_torch(backend, n, max_norm):
    L, dev = backend
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    ref = torch.optim.Adam([ref_p], lr=3e-3, betas=(0.9, 0.999), eps=1e-8)
    p = p0.clone().to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    ws = torch.empty(L.pidm_clip_adam_ws_bytes(), dtype=torch.uint8, device=dev)
    norm = torch.zeros(1, device=dev)
    for step in range(1, 5):
        grad = torch.randn(n, generator=g) * (10.0 if step % 2 else 0.01)
        ref_p.grad = grad.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_([ref_p], max_norm) if max_norm > 0 else grad.norm()
        ref.step()
        gd = grad.to(dev)
        L.check(L.pidm_clip_adam_step(ptr(p), ptr(gd), ptr(m), ptr(v), n, 3e-3, 0.9, 0.999, 1e-8, step, max_norm, ptr(norm),
                                      ptr(ws), stream_ptr(dev)))
        assert abs(norm.item() - ref_norm.item()) <= 2e-6 * ref_norm.item()
        st = ref.state[ref_p]
        # fp32 round-off of one update (different but equally valid operation order): 2e-6 relative + 2e-7 of the scale
        for mine, theirs in ((m, st["exp_avg"]), (v, st["exp_avg_sq"]), (p, ref_p.detach())):
            np.testing.assert_allclose(mine.cpu().numpy(), theirs.numpy(), rtol=2e-6, atol=2e-7 * theirs.abs().max().item())


def test_clip_adam_rejects_bad_arguments(backend):
    L, dev = backend
    x = torch.zeros(64, device=dev)
    ws = torch.empty(L.pidm_clip_adam_ws_bytes(), dtype=torch.uint8, device=dev)
    assert L.pidm_clip_adam_step(ptr(x), ptr(x), ptr(x), ptr(x), 64, 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, None, ptr(ws), stream_ptr(dev)) != 0
    assert L.pidm_clip_adam_step(ptr(x), None, ptr(x), ptr(x), 64, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None, ptr(ws), stream_ptr(dev)) != 0


def test_fused_optimizer_training_steps_match_torch(backend):
    """Two identical models, three training steps each: torch clip+Adam vs FusedClipAdam on flattened parameters."""
    L, dev = backend
    lib = L if dev.type == "cpu" else None
    dim, P, B = 8, 16, 2
    ma, diff, resa, _ = setup(backend, dim, P, 100)
    mb, _, resb, _ = setup(backend, dim, P, 100)
    sd0 = copy.deepcopy(ma.state_dict())
    opt_a = torch.optim.Adam(ma.parameters(), lr=1e-3)
    flat = flatten_parameters(mb, P, lib)
    assert flatten_parameters(mb, P, lib) is flat          # idempotent
    for k, v_ in mb.state_dict().items():                  # flattening does not change any value or key
        assert torch.equal(v_, sd0[k]), k
    opt_b = FusedClipAdam(mb, lr=1e-3, max_norm=1.0, image_size=P, lib=lib)
    g = torch.Generator().manual_seed(0)
    for it in range(2 if dev.type == "cpu" else 3):
        x0 = torch.randn(B, 2, P, P, generator=g)
        x0[:, 1] = torch.exp(0.3 * x0[:, 1])
        x0 = x0.to(dev)
        eps = torch.randn(B, 2, P, P, generator=g).to(dev)
        t = torch.randint(0, 100, (B,), generator=g).to(dev)
        norms = []
        for m, res, opt in ((ma, resa, opt_a), (mb, resb, opt_b)):
            with patched_rng(randint=lambda *a, **k: t.clone(), randn_like=lambda *a, **k: eps.clone()):
                loss, *_ = diff.model_estimation_loss(x0, residual_func=res, c_data=1., c_residual=1e-3)
            opt.zero_grad()
            loss.backward()
            if opt is opt_a:
                norms.append(torch.nn.utils.clip_grad_norm_(m.parameters(), 1.).item())
                opt.step()
            else:
                norms.append(opt.step().item())
        assert abs(norms[0] - norms[1]) <= 1e-5 * norms[0]
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    changed = 0
    for k in pa:
        a, b = pa[k].detach().cpu().numpy(), pb[k].detach().cpu().numpy()
        scale = max(np.abs(a).max(), 1e-6)
        changed += int(not np.array_equal(a, sd0[k].cpu().numpy()))
        if k.endswith(".proj.bias"):
            # conv bias in front of a GroupNorm: its exact gradient is 0, what arrives is round-off noise that Adam
            # normalises to +-lr steps in BOTH implementations - only boundedness is comparable
            assert np.abs(a - b).max() <= 2 * 3 * 1e-3, k
            continue
        assert np.abs(a - b).max() <= 2e-5 * scale + 2e-7, k
    assert changed == len(opt_b.eng.names) - opt_b.eng.n_cond   # exactly the parameters the forward used moved, the rest untouched
    # optimizer state round trip
    sd = copy.deepcopy(opt_b.state_dict())
    opt_c = FusedClipAdam(mb, lr=1e-3, max_norm=1.0, image_size=P, lib=lib)
    opt_c.load_state_dict(sd)
    assert opt_c.step_count == opt_b.step_count and torch.equal(opt_c.exp_avg, opt_b.exp_avg)


def test_full_state_resume_is_bit_exact(backend, tmp_path):
    """save_training_state / load_training_state (extension): 2 steps + save + 2 steps == load + 2 steps."""
    from physicsinformeddiffusionmodels_amd.denoising_utils import EMA, load_training_state, save_training_state
    L, dev = backend
    lib = L if dev.type == "cpu" else None
    dim, P, B = 8, 16, 2

    def make():
        m, diff, res, _ = setup(backend, dim, P, 100)
        opt = FusedClipAdam(m, lr=1e-3, max_norm=1.0, image_size=P, lib=lib)
        ema = EMA(0.99)
        ema.register(m)
        return m, diff, res, opt, ema

    def run(m, diff, res, opt, ema, n, x0):
        for _ in range(n):
            loss, *_ = diff.model_estimation_loss(x0, residual_func=res, c_data=1., c_residual=1e-3)   # draws t, eps from the RNG
            opt.zero_grad()
            loss.backward()
            opt.step()
            ema.update(m)

    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(B, 2, P, P, generator=g)
    x0[:, 1] = torch.exp(0.3 * x0[:, 1])
    x0 = x0.to(dev)
    torch.manual_seed(77)
    a = make()
    n1 = 1 if dev.type == "cpu" else 2          # the host emulator is slow: one step per phase is enough to catch a lost state
    run(*a, n1, x0)
    ck = str(tmp_path / "state.pt")
    save_training_state(ck, a[0], a[3], a[4], iteration=2, extra={"note": "x"})
    run(*a, n1, x0)
    b = make()
    torch.manual_seed(12345)                       # a different RNG position: must be overwritten by the checkpoint
    it, extra = load_training_state(ck, b[0], b[3], b[4])
    assert it == 2 and extra == {"note": "x"}
    run(*b, n1, x0)
    for (k, pa), (_, pb) in zip(a[0].named_parameters(), b[0].named_parameters()):
        assert torch.equal(pa, pb), k
    assert a[3].step_count == b[3].step_count == 2 * n1 and torch.equal(a[3].exp_avg, b[3].exp_avg)
    for k in a[4].shadow:
        assert torch.equal(a[4].shadow[k], b[4].shadow[k]), k
    # the same file is a valid reference-style checkpoint for load_model
    from physicsinformeddiffusionmodels_amd.denoising_utils import load_model
    c = make()[0]
    load_model(ck, c)


def _ema_reference_update(shadow, params, mu):
    """src/denoising_utils.py:174-177, verbatim arithmetic."""
    for k, p in params.items():
        shadow[k] = (1. - mu) * p + mu * shadow[k]


@pytest.mark.parametrize("fused", [True, False])
def test_ema_update_bit_exact_and_lazy_swap(backend, fused):
    """the EMA of main.py:178-183,316.  Five iterations of the main.py loop body (step; update if iteration >
    ema_start; ema(); restore()) with the update inside the Adam kernel (fused) or as the stand-alone flat kernel: the shadow
    equals the reference formula replayed on the same weights BIT FOR BIT (every trainable parameter, the never-used ones
    included); ema()/restore() are pointer flips (no copies) and checkpoints taken in between hold the averaged weights."""
    from physicsinformeddiffusionmodels_amd.denoising_utils import EMA
    L, dev = backend
    lib = L if dev.type == "cpu" else None
    m, diff, res, _ = setup(backend, 8, 16, 100)
    mu, ema_start = 0.99, 1
    ema = EMA(mu)
    ema.register(m)
    opt = FusedClipAdam(m, lr=2e-3, max_norm=1.0, image_size=16, lib=lib, ema=ema if fused else None, ema_start=ema_start)
    ref_shadow = {k: p.detach().clone() for k, p in m.named_parameters() if p.requires_grad}
    x0 = torch.randn(2, 2, 16, 16, generator=torch.Generator().manual_seed(9)).to(dev)
    for iteration in range(5):
        loss, *_ = diff.model_estimation_loss(x0, residual_func=res, c_data=1., c_residual=1e-3)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if iteration > ema_start:
            ema.update(m)
            _ema_reference_update(ref_shadow, {k: p.detach() for k, p in m.named_parameters() if p.requires_grad}, mu)
        train_ptrs = {k: p.data_ptr() for k, p in m.named_parameters()}
        train_vals = {k: p.detach().clone() for k, p in m.named_parameters()}
        ema.ema(res.model)
        for k, p in m.named_parameters():
            if p.requires_grad:
                assert p.data_ptr() == ema.shadow[k].data_ptr(), k      # the parameter IS the shadow now: no copy was made
        sd = m.state_dict()
        for k in ref_shadow:
            assert torch.equal(sd[k], ref_shadow[k]), (iteration, k)    # what save_model would write (main.py:314)
        if iteration == 3:
            with pytest.raises(RuntimeError):
                opt.step()                                                # no optimizer step on swapped-in weights
            opt.step_count -= 1
        ema.restore(res.model)
        for k, p in m.named_parameters():
            assert p.data_ptr() == train_ptrs[k] and torch.equal(p.detach(), train_vals[k]), k
    assert ema._fused_updates == 0
    n_flat = len(ema._flat[2])
    assert n_flat == 265 and len(ref_shadow) > n_flat        # engine parameters flat, the rest updated per tensor
    for k in ref_shadow:
        assert torch.equal(ema.shadow[k], ref_shadow[k]), k
    # a shadow dict loaded from a checkpoint is re-homed into the flat layout at the next update
    ema.load_state_dict({k: v.clone() for k, v in ema.shadow.items()})
    ema.update(m)
    _ema_reference_update(ref_shadow, {k: p.detach() for k, p in m.named_parameters() if p.requires_grad}, mu)
    for k in ref_shadow:
        assert torch.equal(ema.shadow[k], ref_shadow[k]), k
