"""BASELINE configs[0]: the reference's 2-D toy study (`main_toy.py`, `/root/reference/src/denoising_toy_utils.py`) - a point
diffusion on the unit circle with a 3-layer conditional MLP, batch 128-512, CPU.  It is plumbing: no UNet, no PDE residual kernel,
nothing for the gfx950 engine to accelerate (the whole model is 17 k parameters), so this module is a plain-PyTorch restatement
that keeps `main_toy.py` runnable against this package: same function names, argument meaning, return structures, RNG
consumption and `state_dict` keys (`lin1.lin.*`, `lin1.embed.weight`, ...), pinned by golden g24 (genuine reference, injected
RNG; tests/test_toy_config.py).  The schedule tables are the ones `DenoisingDiffusion` builds (bit-identical to the reference,
golden g1), not a second copy.

Reference lines each piece follows are cited in the docstrings.  Not provided (plotting / IO conveniences of the reference's
module: `hdr_plot_style`, `plot_data`, `plot_diffusion`, `array_to_gif` - matplotlib / imageio conveniences with no arithmetic; a
checkpoint's residual / inequality / optimisation callables are pickled with `dill` when it is installed, else with `pickle`)."""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

device = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')      # denoising_toy_utils.py:26 (main_toy.py uses it)
_LOG_FLOOR = -27.6310211159                                                    # log(1e-12): :382


def fix_seeds(seed=42):
    """:28-31"""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)


def noop(*args, **kwargs):
    """:33-34 (the log function when wandb is off)"""


def exists(x):
    return x is not None


def default(val, d):
    return val if exists(val) else (d() if callable(d) else d)


def right_pad_dims_to(x, t):
    """:93-97"""
    extra = x.ndim - t.ndim
    return t if extra <= 0 else t.view(*t.shape, *((1,) * extra))


def create_diff_dict(n_steps, device):
    """:43-83 - the same cosine schedule and derived tables as the image model's (src/denoising_utils.py:275-335, golden g1)."""
    from .denoising_utils import DenoisingDiffusion
    return DenoisingDiffusion.schedule_tables(n_steps, device)


def make_beta_schedule(schedule='linear', n_timesteps=1000, start=1e-5, end=1e-2):
    """:128-144"""
    from .denoising_utils import DenoisingDiffusion
    return DenoisingDiffusion.make_beta_schedule(None, schedule=schedule, n_timesteps=n_timesteps, start=start, end=end)


# ---- data sets (:99-126): numpy RNG, as the reference -----------------------------------------------------------------------
def sample_zeros(size):
    return np.zeros((size, 2))


def sample_gaussian(size, dim=2):
    return np.random.randn(size, dim)


def sample_hypersphere(size, dim):
    pts = np.random.normal(0, 1, (size, dim))
    return pts / np.linalg.norm(pts, axis=1, keepdims=True)


def sample_two_points(size):
    return np.array([[-0.5, -0.5], [0.5, 0.5]])[np.random.randint(2, size=size)]


def sample_four_points(size):
    return np.array([[-1., -1.], [-1., 1.], [1., -1.], [1., 1.]])[np.random.randint(4, size=size)]


def remove_outliers(data, percentile=0.01, also_lower_bound=False):
    """:513-525 - points whose norm lies strictly inside the (percentile, 100 - percentile) band"""
    if data.size == 0:
        return data
    pct = percentile * 100
    norms = np.linalg.norm(data, axis=1)
    lo = np.percentile(norms, pct) if also_lower_bound else 0.
    hi = np.percentile(norms, 100 - pct)
    return data[(norms > lo) & (norms < hi)]


def extract(input, t, x):
    """:146-150: table[t] shaped to broadcast against x"""
    vals = torch.gather(input, 0, t.to(input.device))
    return vals.reshape(t.shape[0], *([1] * (x.dim() - 1)))


def q_sample(x_0, t, alphas_bar_sqrt, one_minus_alphas_bar_sqrt, noise=None):
    """:153-158"""
    if noise is None:
        noise = torch.randn_like(x_0)
    return extract(alphas_bar_sqrt, t, x_0) * x_0 + extract(one_minus_alphas_bar_sqrt, t, x_0) * noise


# ---- the model (:169-197) --------------------------------------------------------------------------------------------------
class ConditionalLinear(nn.Module):
    """Linear layer whose output is scaled by a per-timestep embedding (uniform [0, 1) init, drawn right after the Linear's)."""

    def __init__(self, num_in, num_out, n_steps):
        super().__init__()
        self.num_out = num_out
        self.lin = nn.Linear(num_in, num_out)
        self.embed = nn.Embedding(n_steps, num_out)
        self.embed.weight.data.uniform_()

    def forward(self, x, y):
        return self.embed(y).view(-1, self.num_out) * self.lin(x)


class ConditionalModel(nn.Module):
    def __init__(self, dim, n_steps):
        super().__init__()
        self.lin1 = ConditionalLinear(dim, 128, n_steps)
        self.lin2 = ConditionalLinear(128, 128, n_steps)
        self.lin3 = nn.Linear(128, dim)

    def forward(self, x, y):
        h = F.softplus(self.lin1(x, y))
        h = F.softplus(self.lin2(h, y))
        return self.lin3(h)


# ---- diffusion algebra (:365-434) ------------------------------------------------------------------------------------------
def normal_kl(mean1, logvar1, mean2, logvar2):
    return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * torch.exp(-logvar2))


def gaussian_log_likelihood(x, means, variance, return_full=False):
    sq = ((x - means) ** 2) / variance
    ll = -0.5 * (sq + torch.log(variance) + torch.log(2 * torch.pi)) if return_full else -0.5 * sq
    return torch.clamp(ll, min=_LOG_FLOOR)


def predict_start_from_noise(x_t, t, noise, diff_dict):
    return (extract(diff_dict['sqrt_recip_alphas_cumprod'], t, x_t) * x_t -
            extract(diff_dict['sqrt_recipm1_alphas_cumprod'], t, x_t) * noise)


def predict_noise_from_mean(x_t, t, mean_t, diff_dict):
    return (extract(diff_dict['sqrt_recip_alphas'], t, mean_t) * x_t - mean_t) / extract(diff_dict['noise_mean_coeff'], t, mean_t)


def loss_variational(output, x_0, x_t, t, diff_dict, base_2=False):
    """:396-434 - KL(q(x_{t-1}|x_t,x_0) || p) with the model's mean and the true (clipped) variance; -log p(x_0|x_1) at t = 0."""
    B = x_0.shape[0]
    true_mean = (extract(diff_dict['posterior_mean_coef1'], t, x_t) * x_0 + extract(diff_dict['posterior_mean_coef2'], t, x_t) * x_t)
    var = extract(diff_dict['posterior_variance_clipped'], t, x_t)
    logvar = torch.log(var)
    kl = normal_kl(true_mean, logvar, output, logvar).view(B, -1).mean(dim=1)
    ll = gaussian_log_likelihood(x_0, means=output, variance=var).view(B, -1).mean(dim=1)
    if base_2:
        kl, ll = kl / np.log(2.), ll / np.log(2.)
    assert not ll.isnan().any(), 'Log likelihood is nan.'
    assert not ll.isinf().any(), 'Log likelihood is inf.'
    return torch.where(t == 0, -ll, kl).mean(-1)


# ---- sampler (:199-363) ----------------------------------------------------------------------------------------------------
def _model_to_mean_and_x0(model_pred, x, t, diff_dict, model_pred_mode):
    """(posterior mean, x0 estimate) from the network output under the three parameterisations (:203-232)"""
    if model_pred_mode == 'eps':
        alpha_t = extract(diff_dict['alphas'], t, x)
        eps_factor = (1 - alpha_t) / extract(diff_dict['one_minus_alphas_bar_sqrt'], t, x)
        return (1 / alpha_t.sqrt()) * (x - eps_factor * model_pred), predict_start_from_noise(x, t, model_pred, diff_dict)
    if model_pred_mode == 'x0':
        mean = extract(diff_dict['posterior_mean_coef1'], t, x) * model_pred + extract(diff_dict['posterior_mean_coef2'], t, x) * x
        return mean, model_pred
    if model_pred_mode == 'mu':
        eps = predict_noise_from_mean(x, t, model_pred, diff_dict)
        return model_pred, predict_start_from_noise(x, t, eps, diff_dict)
    raise ValueError('model_pred_mode not recognized.')


def p_sample(model, x, t, diff_dict, model_pred_mode='eps', save_output=False, surpress_noise=False, use_dynamic_threshold=False,
             reduced_ddim_steps=0):
    """One ancestral step (:199-265).  Returns (sample, model output or None, DDIM x0 estimate or None)."""
    t = torch.tensor([t], device=x.device)
    model_pred = model(x, t)
    model_output = model_pred.clone().detach() if save_output else None
    mean, x0_pred = _model_to_mean_and_x0(model_pred, x, t, diff_dict, model_pred_mode)
    z = torch.randn_like(x, device=x.device)
    sigma_t = extract(diff_dict['betas'], t, x).sqrt()
    keep = (1. - (t == 0).float()) if surpress_noise else 1.
    sample = mean + keep * sigma_t * z
    if use_dynamic_threshold:
        s = torch.quantile(sample.float().flatten(1).abs(), 0.9, dim=-1)     # per-sample 90th percentile, floor 1
        s.clamp_(min=1.0)
        s = right_pad_dims_to(sample, s)
        sample = sample.clamp(-s, s) / s
    x0_estimation = None
    if save_output:
        x0_estimation = (ddim_sample_x0(x, t, model, x.shape, reduced_ddim_steps, 0, diff_dict, model_pred_mode=model_pred_mode)
                         if t > 0 else x0_pred)
    return sample, model_output, x0_estimation


def p_sample_loop(model, shape, n_steps, diff_dict, model_pred_mode='x0', save_output=False, surpress_noise=True,
                  use_dynamic_threshold=False, reduced_ddim_steps=0):
    """:267-288.  Returns (x_seq, model outputs, x0 estimates) - the last two are empty lists unless save_output."""
    cur_x = torch.randn(shape, device=diff_dict['alphas'].device)
    x_seq = [cur_x.detach().cpu()]
    model_outputs = [torch.zeros(shape, device='cpu')] if save_output else []
    x0_estimations = [torch.zeros(shape, device='cpu')] if save_output else []
    for i in reversed(range(n_steps)):
        cur_x, out_i, x0_i = p_sample(model, cur_x.detach(), i, diff_dict, model_pred_mode, save_output, surpress_noise,
                                      use_dynamic_threshold, reduced_ddim_steps=reduced_ddim_steps)
        x_seq.append(cur_x.detach().cpu())
        if save_output:
            model_outputs.append(out_i.detach().cpu())
            x0_estimations.append(x0_i.detach().cpu())
    return x_seq, model_outputs, x0_estimations


def ddim_sample_x0(xt, t, model, shape, reduced_n_steps, ddim_sampling_eta, diff_dict, model_pred_mode='eps'):
    """DDIM map from x_t towards x_0 over `reduced_n_steps` + 2 evenly spaced times per sample (:290-363); with 0 reduced steps it
    is model(x_t, t) -> x0 followed by one more evaluation at t = 0 ... of which the last pair (0, -1) returns the x0 estimate."""
    B, dev, eta = shape[0], diff_dict['alphas'].device, ddim_sampling_eta
    batch_t = (torch.ones(B, device=dev, dtype=torch.long) * t) if len(t) == 1 else t
    batch_t = batch_t.cpu().numpy()
    cur_seq, next_seq = [], []
    for tb in batch_t:
        seq = [int(v) for v in np.linspace(0, tb, reduced_n_steps + 2, endpoint=True, dtype=float)]
        cur_seq.append(seq[::-1])
        next_seq.append(([-1] + seq[:-1])[::-1])
    cur_times = torch.tensor(cur_seq, device=dev).T
    next_times = torch.tensor(next_seq, device=dev).T
    cur_x, x0_pred = xt, None
    for tc, tn in zip(cur_times, next_times):
        same = (tc == tn).float().unsqueeze(-1)
        model_pred = model(cur_x, tc)
        if model_pred_mode == 'eps':
            eps_theta = model_pred
            x0_pred = predict_start_from_noise(cur_x, tc, eps_theta, diff_dict)
        elif model_pred_mode == 'x0':
            x0_pred = model_pred
            mean = (extract(diff_dict['posterior_mean_coef1'], tc, cur_x) * x0_pred +
                    extract(diff_dict['posterior_mean_coef2'], tc, cur_x) * cur_x)
            eps_theta = predict_noise_from_mean(cur_x, tc, mean, diff_dict)
        elif model_pred_mode == 'mu':
            eps_theta = predict_noise_from_mean(cur_x, tc, model_pred, diff_dict)
            x0_pred = predict_start_from_noise(cur_x, tc, eps_theta, diff_dict)
        else:
            raise ValueError('model_pred_mode not recognized.')
        if tn[0] < 0:
            assert torch.all(tn == -1), 'Next timesteps should be -1, otherwise this is inconsistent.'
            cur_x = x0_pred
            continue
        alpha, alpha_next = extract(diff_dict['alphas_prod'], tc, cur_x), extract(diff_dict['alphas_prod'], tn, cur_x)
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        noise = torch.randn_like(cur_x)
        stepped = x0_pred * alpha_next.sqrt() + c * eps_theta + sigma * noise
        cur_x = same * cur_x + (1 - same) * stepped
    return cur_x


# ---- training loss (:436-511) ----------------------------------------------------------------------------------------------
def model_estimation_loss(model, x_0, n_steps, diff_dict, model_pred_mode='eps', residual_func=None, ineq_func=None, opt_func=None,
                          c_data=1., c_residual=0., c_ineq=0., lambda_opt=0., use_ddim_x0=False, reduced_ddim_steps=0):
    """The toy counterpart of DenoisingDiffusion.model_estimation_loss: antithetic timestep draw (t and n_steps - 1 - t), q-sample,
    data term under the chosen parameterisation, then -log p(r | x0 estimate) for the residual, the inequality term and the
    exponential-prior optimisation term.  Returns (loss, data loss, mean |residual|, mean inequality, mean objective) - the
    last four as python floats."""
    B = x_0.shape[0]
    t = torch.randint(0, n_steps, size=(B // 2 + 1,), device=x_0.device)
    t = torch.cat([t, n_steps - t - 1], dim=0)[:B].long()
    a = extract(diff_dict['alphas_bar_sqrt'], t, x_0)
    am1 = extract(diff_dict['one_minus_alphas_bar_sqrt'], t, x_0)
    e = torch.randn_like(x_0, device=x_0.device)
    x = x_0 * a + e * am1
    output = model(x, t)
    if model_pred_mode == 'eps':
        loss = F.mse_loss(output, e)
        x_0_pred = predict_start_from_noise(x, t, output, diff_dict)
    elif model_pred_mode == 'x0':
        per = F.mse_loss(output, x_0, reduction='none').flatten(1).mean(dim=1, keepdim=True)
        loss = (per * extract(diff_dict['p2_loss_weight'], t, per)).mean()
        x_0_pred = output
    elif model_pred_mode == 'mu':
        loss = loss_variational(output, x_0, x, t, diff_dict)
        x_0_pred = predict_start_from_noise(x, t, predict_noise_from_mean(x, t, output, diff_dict), diff_dict)
    else:
        raise ValueError('model_pred_mode not recognized.')
    loss = c_data * loss
    # The reference keeps `data_loss = loss` and then adds the other terms IN PLACE (`loss += ...`, :477-505): the tensor it later
    # reports as the data loss is the same object as the total, so the second return value equals the total loss (golden g24 shows
    # it).  Reproduced: the terms below are accumulated in place.
    data_loss = loss
    target = (ddim_sample_x0(x, t, model, x.shape, reduced_ddim_steps, 0, diff_dict, model_pred_mode=model_pred_mode)
              if use_ddim_x0 else x_0_pred)
    residual = residual_func(target)
    var = extract(diff_dict['posterior_variance_clipped'], t, residual)
    residual_loss = -c_residual * gaussian_log_likelihood(torch.zeros_like(residual), means=residual, variance=var).mean()
    loss += residual_loss
    ineq, _ = ineq_func(target)
    loss += -c_ineq * gaussian_log_likelihood(torch.zeros_like(ineq), means=ineq, variance=var).mean()
    objective = opt_func(target)
    loss += (lambda_opt * objective).mean()
    return loss, data_loss.item(), torch.abs(residual).mean().item(), ineq.mean().item(), opt_func(target).mean().item()


# ---- checkpoints (:527-593) ------------------------------------------------------------------------------------------------
def _pickler():
    try:
        import dill
        return dill
    except ImportError:
        return pickle


def save_model(model, name, diff_dict, step, n_steps, dim, model_pred_mode, residual_func, ineq_func, opt_func):
    save_dir = './trained_models/toy/' + name + '/model'
    os.makedirs(save_dir, exist_ok=True)
    base = f'{save_dir}/checkpoint_{step}'
    torch.save(dict(model=model.state_dict(), n_steps=n_steps, dim=dim, model_pred_mode=model_pred_mode, diff_dict=diff_dict), base + '.pt')
    for tag, fn in (('residual_func', residual_func), ('ineq_func', ineq_func), ('opt_func', opt_func)):
        with open(f'{base}_{tag}.pkl', 'wb') as f:
            _pickler().dump(fn, f)
    print(f'checkpoint saved to {save_dir}')


def load_model(path, strict=True):
    obj = torch.load(path, map_location='cpu', weights_only=False)
    model = ConditionalModel(obj['dim'], obj['n_steps'])
    try:
        model.load_state_dict(obj['model'], strict=strict)
    except RuntimeError:
        print('Failed loading state dict.')
    fns = []
    for tag in ('residual_func', 'ineq_func', 'opt_func'):
        with open(path.replace('.pt', f'_{tag}.pkl'), 'rb') as f:
            fns.append(_pickler().load(f))
    return (model, obj['diff_dict'], obj['n_steps'], obj['dim'], obj['model_pred_mode'], *fns)
