"""CPU restatement of the three diffusers 0.24.0 schedulers the stage-2 path can hold in
``pipe.scheduler`` (oracle; test infrastructure -- see oracle/__init__.py, PARITY UNPINNED).

Call sites in the reference: src/pipelines/stage2_inpaint_pipeline.py:472-473 (set_timesteps),
:500 (scale_model_input), :519 (step); chosen at stage2_batchtest_inpaint_model.py:132 (UniPC)
and pcdms_kaggle_demo.ipynb cell 15 (DDIM); DDPM at stage2_train_inpaint_model.py:175,361.
Arithmetic: SURVEY.md Appendix A-9 / A-10 / A-11.  All coefficient math is float64 numpy on the
host except where diffusers itself keeps fp32 tables (betas / alphas_cumprod / sigmas).
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch


def _alphas_cumprod(T=1000, beta_start=0.00085, beta_end=0.012, schedule="scaled_linear") -> torch.Tensor:
    if schedule == "scaled_linear":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2
    elif schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, T, dtype=torch.float32)
    else:
        raise NotImplementedError(schedule)
    return torch.cumprod(1.0 - betas, dim=0), betas


class DDIMOracle:
    """Appendix A-9.  ctor args = notebook cell 15."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False,
                 steps_offset=1, prediction_type="epsilon"):
        assert prediction_type == "epsilon" and not clip_sample
        self.T = num_train_timesteps
        self.alphas_cumprod, self.betas = _alphas_cumprod(self.T, beta_start, beta_end, beta_schedule)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.steps_offset = steps_offset
        self.timesteps = torch.arange(self.T - 1, -1, -1)
        self.num_inference_steps = None

    def set_timesteps(self, n: int, device=None):
        self.num_inference_steps = n
        ratio = self.T // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, x, t=None):
        return x

    def coefficients(self, t: int):
        """x_prev = cx * x + ce * eps  (eta = 0)."""
        tp = t - self.T // self.num_inference_steps
        a = float(self.alphas_cumprod[t])
        ap = float(self.alphas_cumprod[tp]) if tp >= 0 else float(self.final_alpha_cumprod)
        return a, ap

    def step(self, eps: torch.Tensor, t, x: torch.Tensor, eta: float = 0.0, generator=None,
             variance_noise=None):
        t = int(t)
        a, ap = self.coefficients(t)
        x0 = (x - math.sqrt(1 - a) * eps) / math.sqrt(a)
        var = (1 - ap) / (1 - a) * (1 - a / ap)
        std = eta * math.sqrt(var)
        prev = math.sqrt(ap) * x0 + math.sqrt(1 - ap - std * std) * eps
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(eps.shape, generator=generator, dtype=eps.dtype)
            prev = prev + std * variance_noise
        return prev


class DDPMOracle:
    """Appendix A-11 (``fixed_small`` variance, epsilon prediction)."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear"):
        self.T = num_train_timesteps
        self.alphas_cumprod, self.betas = _alphas_cumprod(self.T, beta_start, beta_end, beta_schedule)
        self.timesteps = torch.arange(self.T - 1, -1, -1)
        self.num_inference_steps = None

    def set_timesteps(self, n: int, device=None):
        self.num_inference_steps = n
        ratio = self.T // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64))

    def scale_model_input(self, x, t=None):
        return x

    def add_noise(self, x0, noise, t: torch.Tensor):
        a = self.alphas_cumprod[t].to(x0.dtype)
        while a.dim() < x0.dim():
            a = a[..., None]
        return a.sqrt() * x0 + (1 - a).sqrt() * noise

    def step(self, eps, t, x, generator=None, variance_noise=None):
        t = int(t)
        n = self.num_inference_steps or self.T
        tp = t - self.T // n
        a = float(self.alphas_cumprod[t])
        ap = float(self.alphas_cumprod[tp]) if tp >= 0 else 1.0
        cur_alpha = a / ap
        cur_beta = 1 - cur_alpha
        x0 = (x - math.sqrt(1 - a) * eps) / math.sqrt(a)
        mean = (math.sqrt(ap) * cur_beta / (1 - a)) * x0 + (math.sqrt(cur_alpha) * (1 - ap) / (1 - a)) * x
        if t > 0:
            var = max((1 - ap) / (1 - a) * cur_beta, 1e-20)
            if variance_noise is None:
                variance_noise = torch.randn(eps.shape, generator=generator, dtype=eps.dtype)
            mean = mean + math.sqrt(var) * variance_noise
        return mean


class UniPCOracle:
    """Appendix A-10: UniPC (bh2, order 2, predict_x0, lower_order_final, linspace spacing)."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", solver_order=2, solver_type="bh2",
                 lower_order_final=True, prediction_type="epsilon"):
        assert prediction_type == "epsilon" and solver_type in ("bh1", "bh2")
        self.T = num_train_timesteps
        self.alphas_cumprod, _ = _alphas_cumprod(self.T, beta_start, beta_end, beta_schedule)
        self.solver_order = solver_order
        self.solver_type = solver_type
        self.lower_order_final = lower_order_final
        self.set_timesteps(self.T)

    def set_timesteps(self, n: int, device=None):
        self.num_inference_steps = n
        ts = np.linspace(0, self.T - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        ac = self.alphas_cumprod.numpy()
        sig = ((1 - ac) / ac) ** 0.5
        sigmas = np.interp(ts, np.arange(0, len(sig)), sig)
        sigma_last = ((1 - ac[0]) / ac[0]) ** 0.5
        self.sigmas = np.concatenate([sigmas, [sigma_last]]).astype(np.float32)
        self.timesteps = torch.from_numpy(ts)
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = None
        self.this_order = 1

    def scale_model_input(self, x, t=None):
        return x

    @staticmethod
    def _als(sigma):
        sigma = np.float32(sigma)
        alpha = np.float32(1.0) / np.sqrt(sigma * sigma + np.float32(1.0), dtype=np.float32)
        return alpha, sigma * alpha

    def _lam(self, i):
        a, s = self._als(self.sigmas[i])
        return np.log(a) - np.log(s), a, s

    def _rb(self, rks, hh, order):
        h_phi_1 = np.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.solver_type == "bh1" else np.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(np.power(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return np.stack(R), np.array(b), h_phi_1, B_h

    def _predict(self, x, order):
        i = self.step_index
        m0 = self.model_outputs[-1]
        lam_t, alpha_t, sigma_t = self._lam(i + 1)
        lam_s0, _, sigma_s0 = self._lam(i)
        h = lam_t - lam_s0
        rks, D1s = [], []
        for k in range(1, order):
            lam_si, _, _ = self._lam(i - k)
            rk = (lam_si - lam_s0) / h
            rks.append(rk)
            D1s.append((self.model_outputs[-(k + 1)] - m0) / float(rk))
        rks.append(1.0)
        R, b, h_phi_1, B_h = self._rb(np.array(rks, dtype=np.float32), -h, order)
        x_t = float(sigma_t / sigma_s0) * x - float(alpha_t * h_phi_1) * m0
        if D1s:
            rhos = np.array([0.5]) if order == 2 else np.linalg.solve(R[:-1, :-1], b[:-1])
            res = sum(float(r) * d for r, d in zip(rhos, D1s))
            x_t = x_t - float(alpha_t * B_h) * res
        return x_t

    def _correct(self, model_t, last_sample, order):
        i = self.step_index
        m0 = self.model_outputs[-1]
        lam_t, alpha_t, sigma_t = self._lam(i)
        lam_s0, _, sigma_s0 = self._lam(i - 1)
        h = lam_t - lam_s0
        rks, D1s = [], []
        for k in range(1, order):
            lam_si, _, _ = self._lam(i - (k + 1))
            rk = (lam_si - lam_s0) / h
            rks.append(rk)
            D1s.append((self.model_outputs[-(k + 1)] - m0) / float(rk))
        rks.append(1.0)
        R, b, h_phi_1, B_h = self._rb(np.array(rks, dtype=np.float32), -h, order)
        rhos = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
        x_t = float(sigma_t / sigma_s0) * last_sample - float(alpha_t * h_phi_1) * m0
        res = sum(float(r) * d for r, d in zip(rhos[:-1], D1s)) if D1s else 0.0
        return x_t - float(alpha_t * B_h) * (res + float(rhos[-1]) * (model_t - m0))

    def step(self, eps, t, x):
        if self.step_index is None:
            self.step_index = int((self.timesteps == int(t)).nonzero()[0])
        i = self.step_index
        alpha_t, sigma_t = self._als(self.sigmas[i])
        x0 = (x - float(sigma_t) * eps) / float(alpha_t)
        if i > 0 and self.last_sample is not None:
            x = self._correct(x0, self.last_sample, self.this_order)
        for k in range(self.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0
        order = min(self.solver_order, len(self.timesteps) - i) if self.lower_order_final else self.solver_order
        self.this_order = min(order, self.lower_order_nums + 1)
        self.last_sample = x
        prev = self._predict(x, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev


def _betas_for_alpha_bar(T: int, max_beta: float = 0.999) -> torch.Tensor:
    """diffusers ``betas_for_alpha_bar`` (cosine / "squaredcos_cap_v2")."""
    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return torch.tensor([min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), max_beta) for i in range(T)], dtype=torch.float32)


class UnCLIPOracle:
    """diffusers 0.24.0 ``UnCLIPScheduler`` as the stage-1 prior uses it (called at
    /root/reference/src/pipelines/stage1_prior_pipeline.py:439-440 ``set_timesteps`` and :478-483
    ``step(pred, timestep=t, sample=latents, prev_timestep=...)``).  Defaults = the Kandinsky-2.2 prior's
    ``scheduler_config.json`` (prediction_type "sample", clip +-10, "fixed_small_log" variance, cosine betas).
    PARITY UNPINNED ([D-0.24] block, restated from the published algorithm)."""

    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, variance_type="fixed_small_log", clip_sample=True,
                 clip_sample_range=10.0, prediction_type="sample", beta_schedule="squaredcos_cap_v2"):
        assert beta_schedule == "squaredcos_cap_v2" and variance_type == "fixed_small_log"
        assert prediction_type in ("sample", "epsilon")
        self.T = num_train_timesteps
        self.betas = _betas_for_alpha_bar(self.T)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.clip_sample, self.clip_sample_range, self.prediction_type = clip_sample, clip_sample_range, prediction_type
        self.timesteps = torch.arange(self.T - 1, -1, -1)

    def set_timesteps(self, n: int, device=None):
        ratio = (self.T - 1) / (n - 1)
        self.timesteps = torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64))

    def scale_model_input(self, x, t=None):
        return x

    def coefficients(self, t: int, prev_t: Optional[int]):
        """pred_prev = c_x0 * clip(x0) + c_x * x + std * noise (std = 0 at t = 0)."""
        if prev_t is None:
            prev_t = t - 1
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if prev_t == t - 1:
            beta, alpha = float(self.betas[t]), float(self.alphas[t])
        else:
            beta = 1 - a_t / a_prev
            alpha = 1 - beta
        c_x0 = math.sqrt(a_prev) * beta / b_t
        c_x = math.sqrt(alpha) * b_prev / b_t
        std = 0.0
        if t > 0:
            var = max(b_prev / b_t * beta, 1e-20)
            std = math.exp(0.5 * math.log(var))
        return c_x0, c_x, std, a_t

    def step(self, model_output, t, sample, prev_timestep=None, generator=None, variance_noise=None):
        t = int(t)
        prev_t = None if prev_timestep is None else int(prev_timestep)
        c_x0, c_x, std, a_t = self.coefficients(t, prev_t)
        if self.prediction_type == "epsilon":
            x0 = (sample - math.sqrt(1 - a_t) * model_output) / math.sqrt(a_t)
        else:
            x0 = model_output
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        prev = c_x0 * x0 + c_x * sample
        if t > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return prev
