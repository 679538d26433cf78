"""ORACLE -- test infrastructure only.  CPU restatement of the two CogVideoX schedulers and of the CFG combine of
the denoise loop.  Pinned bit-exactly against the imported reference (tests/golden/sched_*.npz).

Reference: diffusers/src/diffusers/schedulers/scheduling_ddim_cogvideox.py (alphas :179-231, zero-SNR :95-123,
set_timesteps :260-303, step :305-402), scheduling_dpm_cogvideox.py (step :306-439),
src/custom_cogvideox_pipe.py:266-296 (CFG, dtype round).
Scalars stay 0-dim float64 torch tensors exactly as in the reference, so torch's type promotion (a 0-dim fp64
scalar times a bf16 tensor is computed as bf16(scalar) * tensor) is reproduced, not re-derived.
"""
import math

import numpy as np
import torch


def alphas_cumprod(snr_shift_scale, num_train=1000, beta_start=0.00085, beta_end=0.012, zero_snr=True):
    betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train, dtype=torch.float64) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
    if zero_snr:
        s = ac.sqrt()
        s0, sT = s[0].clone(), s[-1].clone()
        s -= sT
        s *= s0 / (s0 - sT)
        ac = s**2
    return ac


def trailing_timesteps(n, num_train=1000):
    return (np.round(np.arange(num_train, 0, -num_train / n)).astype(np.int64) - 1)


def cfg_combine(noise_pred, guidance):
    """custom_cogvideox_pipe.py:266-279."""
    noise_pred = noise_pred.float()
    u, c = noise_pred.chunk(2)
    return u + guidance * (c - u)


def dynamic_guidance(guidance, n_steps, i):
    """custom_cogvideox_pipe.py:269-272."""
    return 1 + guidance * ((1 - math.cos(math.pi * ((n_steps - i) / n_steps) ** 5.0)) / 2)


def ddim_step(ac, n_steps, model_output, t, sample, num_train=1000):
    """scheduling_ddim_cogvideox.py:364-394 (v_prediction, set_alpha_to_one).  Returns (prev_sample, x0)."""
    prev_t = t - num_train // n_steps
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    b_t = 1 - a_t
    x0 = (a_t**0.5) * sample - (b_t**0.5) * model_output
    at = ((1 - a_prev) / (1 - a_t)) ** 0.5
    bt = a_prev**0.5 - a_t**0.5 * at
    return at * sample + bt * x0, x0


def dpm_step(ac, n_steps, model_output, old_x0, t, t_back, sample, noise1, noise2, num_train=1000):
    """scheduling_dpm_cogvideox.py:391-434.  noise1/noise2 are the two randn draws (the reference always draws the
    first one, and a second one on multistep steps).  Returns (prev_sample, x0)."""
    prev_t = t - num_train // n_steps
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    a_back = ac[t_back] if t_back is not None else None
    b_t = 1 - a_t
    x0 = (a_t**0.5) * sample - (b_t**0.5) * model_output
    lamb = ((a_t / (1 - a_t)) ** 0.5).log()
    lamb_next = ((a_prev / (1 - a_prev)) ** 0.5).log()
    h = lamb_next - lamb
    m1 = ((1 - a_prev) / (1 - a_t)) ** 0.5 * (-h).exp()
    m2 = (-2 * h).expm1() * a_prev**0.5
    mn = (1 - a_prev) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
    prev = m1 * sample - m2 * x0 + mn * noise1
    if old_x0 is None or prev_t < 0:
        return prev, x0
    lamb_prev = ((a_back / (1 - a_back)) ** 0.5).log()
    r = (lamb - lamb_prev) / h
    m3, m4 = 1 + 1 / (2 * r), 1 / (2 * r)
    d = m3 * x0 - m4 * old_x0
    return m1 * sample - m2 * d + mn * noise2, x0
