"""Host-side mirrors of the reference's scheduler objects (same constructor arguments, attributes and `step`
signatures) whose arithmetic runs in the HIP kernel `sched_step_k` through the C ABI (s2v_sched_step /
s2v_denoise_step).

Reference: diffusers/src/diffusers/schedulers/scheduling_ddim_cogvideox.py:179-231 (alphas), :260-303 (timesteps),
:305-402 (step); scheduling_dpm_cogvideox.py:306-439 (step); protocol use in src/custom_cogvideox_pipe.py:200-205,
250,282-296.

The per-step scalars are evaluated on the host with 0-dim float64 torch tensors, expression by expression as the
reference does, and then cast the way torch's type promotion casts them before the tensor multiply:
  * a scalar that multiplies a model-dtype tensor (sample, noise) is rounded fp64 -> fp32 -> model dtype;
  * a scalar that multiplies an fp32 tensor (model_output, x0) is rounded fp64 -> fp32.
"""
import ctypes

import numpy as np
import torch

from . import _lib, tables


def _f32(x):
    return float(torch.as_tensor(x, dtype=torch.float64).to(torch.float32))


def _as_model(x, dtype):
    t = torch.as_tensor(x, dtype=torch.float64).to(torch.float32)
    return float(t.to(dtype).to(torch.float32))


class _SchedulerBase:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 clip_sample=False, set_alpha_to_one=True, prediction_type="v_prediction",
                 timestep_spacing="trailing", rescale_betas_zero_snr=True, snr_shift_scale=3.0, scalar_semantics="cpu",
                 **unused):
        if beta_schedule != "scaled_linear" or prediction_type != "v_prediction" or timestep_spacing != "trailing" \
                or not rescale_betas_zero_snr or not set_alpha_to_one or clip_sample:
            raise NotImplementedError("only the CogVideoX scheduler configuration is implemented "
                                      "(scaled_linear, v_prediction, trailing, zero-SNR, set_alpha_to_one)")
        # The per-step scalars that multiply model-dtype tensors (sqrt(alpha_t), a_t, m1, mn): torch on the CPU materialises such a
        # 0-dim fp64 scalar in the tensor's dtype (rounded to bf16 for a bf16 sample) -- "cpu", the default, what the committed
        # goldens were generated with; on a CUDA device torch fetches it as the op's fp32 math type and never rounds it -- "cuda".
        if scalar_semantics not in ("cpu", "cuda"):
            raise ValueError("scalar_semantics must be 'cpu' or 'cuda'")
        self.scalar_semantics = scalar_semantics
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           snr_shift_scale=snr_shift_scale)
        self.alphas_cumprod = torch.from_numpy(
            tables.alphas_cumprod(snr_shift_scale, num_train_timesteps, beta_start, beta_end))
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def _sc(self, x, dtype):
        return _as_model(x, dtype) if self.scalar_semantics == "cpu" else _f32(x)

    def set_timesteps(self, num_inference_steps, device=None):
        n_train = self.config["num_train_timesteps"]
        if num_inference_steps > n_train:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than {n_train}")
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(tables.trailing_timesteps(num_inference_steps, n_train)).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alphas(self, timestep):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' first")
        t = int(timestep)
        prev_t = t - self.config["num_train_timesteps"] // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return t, prev_t, a_t, a_prev

    def _run(self, coef, model_output, sample, x0_hist, noise, flags_extra=0):
        """scheduler.step seam: fp32 model_output, model-dtype sample -> (fp32 prev_sample, fp32 x0)."""
        dt = _lib.DTYPE_OF[sample.dtype]
        mo = model_output.float().contiguous()
        sample = sample.contiguous()
        prev = torch.empty(sample.shape, dtype=torch.float32, device=sample.device)
        if x0_hist is None:
            x0_hist = torch.empty(sample.shape, dtype=torch.float32, device=sample.device)
        flags = 2 | 4 | flags_extra  # fp32 model_output in, fp32 un-rounded prev_sample out
        _lib.check(_lib.lib().s2v_sched_step(None, ctypes.byref(coef), _lib.ptr(mo), flags, _lib.ptr(sample),
                                             _lib.ptr(prev), _lib.ptr(x0_hist), _lib.ptr(noise), sample.numel(), dt,
                                             _lib.stream_ptr()))
        return prev, x0_hist


class CogVideoXDDIMScheduler(_SchedulerBase):
    def coef(self, timestep, dtype, guidance=1.0):
        """scalars of scheduling_ddim_cogvideox.py:364-394 for one step"""
        _, _, a_t, a_prev = self._alphas(timestep)
        b_t = 1 - a_t
        at = ((1 - a_prev) / (1 - a_t)) ** 0.5
        bt = a_prev**0.5 - a_t**0.5 * at
        c = _lib.SchedCoefC()
        c.kind, c.guidance = 0, float(np.float32(guidance))
        c.c_x0_x, c.c_x0_v = self._sc(a_t**0.5, dtype), _f32(b_t**0.5)
        c.a_t, c.b_t = self._sc(at, dtype), _f32(bt)
        return c

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        prev, x0 = self._run(self.coef(timestep, sample.dtype), model_output, sample, None, None)
        if not return_dict:
            return (prev, x0)
        return dict(prev_sample=prev, pred_original_sample=x0)


class CogVideoXDPMScheduler(_SchedulerBase):
    def coef(self, timestep, timestep_back, first, dtype, guidance=1.0):
        """scalars of scheduling_dpm_cogvideox.py:391-434; `first` = no old_pred_original_sample yet"""
        _, prev_t, a_t, a_prev = self._alphas(timestep)
        b_t = 1 - a_t
        lamb = ((a_t / (1 - a_t)) ** 0.5).log()
        lamb_next = ((a_prev / (1 - a_prev)) ** 0.5).log()
        h = lamb_next - lamb
        m1 = ((1 - a_prev) / (1 - a_t)) ** 0.5 * (-h).exp()
        m2 = (-2 * h).expm1() * a_prev**0.5
        mn = (1 - a_prev) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
        c = _lib.SchedCoefC()
        c.guidance = float(np.float32(guidance))
        c.c_x0_x, c.c_x0_v = self._sc(a_t**0.5, dtype), _f32(b_t**0.5)
        c.m1, c.m2, c.mn = self._sc(m1, dtype), _f32(m2), self._sc(mn, dtype)
        if first or prev_t < 0:
            c.kind = 1
        else:
            a_back = self.alphas_cumprod[int(timestep_back)]
            lamb_prev = ((a_back / (1 - a_back)) ** 0.5).log()
            r = (lamb - lamb_prev) / h
            c.kind, c.m3, c.m4 = 2, _f32(1 + 1 / (2 * r)), _f32(1 / (2 * r))
        return c

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, eta=0.0,
             use_clipped_model_output=False, generator=None, variance_noise=None, return_dict=False):
        first = old_pred_original_sample is None
        c = self.coef(timestep, timestep_back, first, sample.dtype)
        # the reference draws randn once, and a second time on multistep steps (the first draw is then discarded)
        shape, dev = sample.shape, sample.device
        gdev = generator.device.type if generator is not None else dev.type
        noise = torch.randn(shape, generator=generator, device=gdev if gdev == "cpu" else dev, dtype=sample.dtype)
        if c.kind == 2:
            noise = torch.randn(shape, generator=generator, device=gdev if gdev == "cpu" else dev, dtype=sample.dtype)
        noise = noise.to(dev)
        hist = old_pred_original_sample.float().clone() if not first else None
        prev, x0 = self._run(c, model_output, sample, hist, noise)
        return (prev, x0)
