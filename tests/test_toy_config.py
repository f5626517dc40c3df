"""BASELINE configs[0]: the 2-D toy study (main_toy.py:113-130 -> src/denoising_toy_utils.py).  CPU plumbing, no kernels: the
package's plain-PyTorch restatement against the genuine reference (golden g24: default init under a seed, schedule dictionary,
model_estimation_loss in its three parameterisations x two x0 estimates with injected RNG incl. all gradient norms, a 6-step
p_sample_loop with save_output), plus a short run of main_toy.py's loop body."""
import os

import numpy as np
import pytest
import torch

from oracle import pidm_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def _funcs():
    residual = lambda x: torch.sum(x ** 2, dim=1) - 1.0                     # noqa: E731  (main_toy.py:49-54)

    def ineq(x):                                                             # main_toy.py:56-70, threshold 1, mode 'leq'
        density = torch.sum(torch.abs(x), dim=1)
        return torch.relu(density - 1.0), density
    opt = lambda x: x[:, 0]                                                  # noqa: E731  (main_toy.py:72-77)
    return residual, ineq, opt


class patched:
    def __init__(self, **fns):
        self.fns = fns

    def __enter__(self):
        self.orig = {k: getattr(torch, k) for k in self.fns}
        for k, v in self.fns.items():
            setattr(torch, k, v)

    def __exit__(self, *a):
        for k, v in self.orig.items():
            setattr(torch, k, v)


def test_default_init_and_schedule_match_the_reference():
    import src.denoising_toy_utils as toy
    g = np.load(os.path.join(G, "g24_toy_config.npz"))
    torch.manual_seed(5)
    m = toy.ConditionalModel(2, 100)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["init/names"]]
    assert np.array_equal(np.array([v.double().sum().item() for v in sd.values()]), g["init/sum"])
    assert np.array_equal(np.array([v.double().abs().sum().item() for v in sd.values()]), g["init/abs_sum"])
    assert np.array_equal(torch.rand(3).numpy(), g["init/next_rand"])        # same RNG consumption by the constructor
    dd = toy.create_diff_dict(100, "cpu")
    assert sorted(dd.keys()) == [str(k) for k in g["sched/names"]]
    for k, v in dd.items():
        assert np.array_equal(v.numpy(), g["sched/" + k]), k                 # bit-exact tables


@pytest.mark.parametrize("mode", ["x0", "eps", "mu"])
@pytest.mark.parametrize("est", ["mean", "sample"])
def test_model_estimation_loss(mode, est):
    import src.denoising_toy_utils as toy
    g = np.load(os.path.join(G, "g24_toy_config.npz"))
    residual, ineq, opt = _funcs()
    m = toy.ConditionalModel(2, 100)
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    dd = toy.create_diff_dict(100, "cpu")
    x0, t_half, eps = (torch.from_numpy(g[k]) for k in ("x0", "t_half", "eps"))
    extra = [torch.randn(9, 2, generator=torch.Generator().manual_seed(243 + i)) for i in range(4)]
    it = iter([eps] + extra)
    with patched(randint=lambda *a, **k: t_half.clone(), randn_like=lambda *a, **k: next(it).clone()):
        out = toy.model_estimation_loss(m, x0, 100, dd, model_pred_mode=mode, residual_func=residual, ineq_func=ineq, opt_func=opt,
                                        c_data=1.0, c_residual=0.005, c_ineq=0.1, lambda_opt=0.01, use_ddim_x0=est == "sample",
                                        reduced_ddim_steps=1 if est == "sample" else 0)
    out[0].backward()
    ref = g[f"loss/{mode}/{est}"]
    got = np.array([out[0].item(), out[1], out[2], out[3], out[4]])
    assert all(isinstance(v, float) for v in out[1:])
    assert np.allclose(got, ref, rtol=2e-5, atol=1e-6), (got, ref)
    assert out[1] == out[0].item()          # the reference's in-place accumulation: its "data loss" IS the total (see the module)
    gn = np.array([p.grad.double().norm().item() for p in m.parameters()])
    assert np.allclose(gn, g[f"loss/{mode}/{est}/grad_norms"], rtol=2e-4, atol=1e-6 * gn.max())


def test_p_sample_loop_with_saved_outputs():
    import src.denoising_toy_utils as toy
    g = np.load(os.path.join(G, "g24_toy_config.npz"))
    m = toy.ConditionalModel(2, 6)
    m.load_state_dict(O.fill_state_dict(m.state_dict()))
    dd = toy.create_diff_dict(6, "cpu")
    noises = [torch.from_numpy(n) for n in g["sampler/noises"]]
    used = {"n": 0}

    def draw(*a, **k):
        used["n"] += 1
        return noises[used["n"] - 1].clone()
    with patched(randn=draw, randn_like=draw), torch.no_grad():
        x_seq, outs, x0s = toy.p_sample_loop(m, [7, 2], 6, dd, model_pred_mode="x0", save_output=True, surpress_noise=True,
                                             reduced_ddim_steps=0)
    assert used["n"] == int(g["sampler/draws"])                              # same RNG consumption (z per step + the DDIM map's draws)
    for got, key in ((x_seq, "x_seq"), (outs, "model_outputs"), (x0s, "x0_estimations")):
        ref = g["sampler/" + key]
        assert len(got) == len(ref) == 7
        assert np.allclose(torch.stack(got).numpy(), ref, rtol=2e-5, atol=2e-6), key


def test_main_toy_loop_body_trains():
    """main_toy.py:113-130 for a few iterations at its batch size: the loss falls and the samples move towards the unit circle."""
    import src.denoising_toy_utils as toy
    torch.manual_seed(0)
    np.random.seed(0)
    residual, ineq, opt = _funcs()
    data = torch.tensor(toy.sample_hypersphere(2048, 2)).float()
    model = toy.ConditionalModel(2, 100)
    optimizer = torch.optim.Adam(model.parameters(), lr=5.e-4)
    dd = toy.create_diff_dict(100, toy.device if False else "cpu")
    losses = []
    for it in range(60):
        perm = torch.randperm(data.size(0))
        for i in range(0, data.size(0), 512):
            batch_x = data[perm[i:i + 512]]
            loss, *_ = toy.model_estimation_loss(model, batch_x, 100, dd, model_pred_mode="x0", residual_func=residual, ineq_func=ineq,
                                                 opt_func=opt, c_data=1.0, c_residual=0.005, c_ineq=0.0, lambda_opt=0.0,
                                                 use_ddim_x0=True, reduced_ddim_steps=0)
            optimizer.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            optimizer.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0]
    with torch.no_grad():
        seqs = toy.p_sample_loop(model, [500, 2], 100, dd, model_pred_mode="x0", save_output=False, surpress_noise=True)
    r0 = residual(seqs[0][0]).abs().mean().item()
    r1 = residual(seqs[0][-1]).abs().mean().item()
    assert r1 < 0.6 * r0, (r0, r1)
