"""MI355X-native stand-ins for the reference's ``pipe.scheduler`` objects.

The reference uses three diffusers 0.24.0 schedulers at this boundary:
``UniPCMultistepScheduler`` (stage2_batchtest_inpaint_model.py:132), ``DDIMScheduler``
(pcdms_kaggle_demo.ipynb cell 15 -- the configuration BASELINE.json's metric names) and
``DDPMScheduler`` (stage2_train_inpaint_model.py:175,361).  Call sites:
src/pipelines/stage2_inpaint_pipeline.py:472-473 (set_timesteps / timesteps), :386
(init_noise_sigma), :494 (order), :500 (scale_model_input), :519 (step; ``eta`` / ``generator`` are
passed only if ``inspect.signature(step)`` has them, :313-321).

Coefficient math is scalar host code (float64, tables in fp32 where diffusers keeps them in fp32);
every tensor update is a hand-written HIP kernel from libpcdm.so (``pcdm_lincomb`` /
``pcdm_cfg_step``).  Tensors must be on the GPU: there is no CPU tensor path.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from . import ops


class SchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample

    def __getitem__(self, i):
        return (self.prev_sample,)[i]


class _Config(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, d=None):
        return getattr(self, k, d)

    def keys(self):
        return self.__dict__.keys()


def _betas(cfg) -> np.ndarray:
    T = cfg.num_train_timesteps
    if cfg.trained_betas is not None:
        return np.asarray(cfg.trained_betas, dtype=np.float32)
    if cfg.beta_schedule == "linear":
        return torch.linspace(cfg.beta_start, cfg.beta_end, T, dtype=torch.float32).numpy()
    if cfg.beta_schedule == "scaled_linear":
        return (torch.linspace(cfg.beta_start ** 0.5, cfg.beta_end ** 0.5, T, dtype=torch.float32) ** 2).numpy()
    if cfg.beta_schedule == "squaredcos_cap_v2":   # diffusers betas_for_alpha_bar (cosine), max_beta 0.999
        def alpha_bar(t):
            return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.asarray([min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), 0.999) for i in range(T)], dtype=np.float32)
    raise NotImplementedError(f"{cfg.beta_schedule} does is not implemented")


def _f32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda and not ops._lib.is_emulator():
        raise RuntimeError("scheduler.step needs GPU tensors (no CPU tensor path in pcdms_amd)")
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class _Base:
    order = 1
    init_noise_sigma = 1.0
    _defaults: dict = {}

    def __init__(self, **kwargs):
        cfg = dict(self._defaults)
        cfg.update(kwargs)   # from_config semantics (diffusers): keys of other, compatible schedulers do not act here but stay in
        self.config = _Config(**cfg)   # .config ("hidden" attributes), so that A.from_config(B.from_config(A.config).config) round-trips
        betas = _betas(self.config)
        self.betas = torch.from_numpy(betas.copy())
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)  # fp32, as diffusers
        self._ac = self.alphas_cumprod.numpy().astype(np.float64)
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.arange(self.config.num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = dict(config.__dict__ if isinstance(config, SimpleNamespace) else config)
        cfg.update(kwargs)
        return cls(**cfg)

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def __len__(self):
        return self.config.num_train_timesteps

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """sqrt(ac_t) x0 + sqrt(1-ac_t) noise with one coefficient pair per batch row."""
        x0, n = _f32(original_samples), _f32(noise)
        out = torch.empty_like(x0)
        ts = [int(v) for v in timesteps.reshape(-1).tolist()]
        if len(ts) == 1:
            ts = ts * x0.shape[0]
        for b, t in enumerate(ts):
            a = float(self._ac[t])
            ops.lincomb(out[b], [x0[b], n[b]], [math.sqrt(a), math.sqrt(1 - a)])
        return out.to(original_samples.dtype)


class DDIMScheduler(_Base):
    """SURVEY.md Appendix A-9.  ``DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
    beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1)`` is the
    notebook's (cell 15) configuration."""

    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                     prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                     clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading", rescale_betas_zero_snr=False)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if c.prediction_type != "epsilon" or c.thresholding or c.rescale_betas_zero_snr:
            raise NotImplementedError("only epsilon prediction without thresholding is on the stage-2 path")
        self.final_alpha_cumprod = 1.0 if c.set_alpha_to_one else float(self._ac[0])

    def set_timesteps(self, num_inference_steps: int, device=None):
        c = self.config
        if num_inference_steps > c.num_train_timesteps:
            raise ValueError("`num_inference_steps` cannot be larger than `num_train_timesteps`")
        self.num_inference_steps = num_inference_steps
        T = c.num_train_timesteps
        if c.timestep_spacing == "leading":
            ratio = T // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + c.steps_offset
        elif c.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif c.timestep_spacing == "trailing":
            ts = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
        else:
            raise ValueError(c.timestep_spacing)
        self.timesteps = torch.from_numpy(ts).to(device)

    def _alphas(self, t: int) -> Tuple[float, float]:
        tp = t - self.config.num_train_timesteps // self.num_inference_steps
        return float(self._ac[t]), (float(self._ac[tp]) if tp >= 0 else self.final_alpha_cumprod)

    def step_coefficients(self, t: int, eta: float = 0.0) -> Tuple[float, float, float, float, float]:
        """(cx, ce, cn, c0x, c0e): x_prev = cx*x + ce*eps + cn*noise ; x0 = c0x*x + c0e*eps."""
        a, ap = self._alphas(int(t))
        var = (1 - ap) / (1 - a) * (1 - a / ap)
        std = eta * math.sqrt(max(var, 0.0))
        c0x, c0e = 1.0 / math.sqrt(a), -math.sqrt(1 - a) / math.sqrt(a)
        if self.config.clip_sample:
            raise NotImplementedError("clip_sample=True is not linear; the stage-2 path uses clip_sample=False")
        dirc = math.sqrt(max(1 - ap - std * std, 0.0))
        return math.sqrt(ap) * c0x, math.sqrt(ap) * c0e + dirc, std, c0x, c0e

    def coefficient_table(self, eta: float = 0.0, device=None) -> torch.Tensor:
        """fp32 [n_steps, 4] rows {cx, ce, cn, 0} in timestep order, for the device-indexed fused step."""
        rows = [list(self.step_coefficients(int(t), eta)[:3]) + [0.0] for t in self.timesteps.tolist()]
        return torch.tensor(rows, dtype=torch.float32, device=device)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise: Optional[torch.Tensor] = None,
             return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        cx, ce, cn, c0x, c0e = self.step_coefficients(int(timestep), eta)
        e, x = _f32(model_output), _f32(sample)
        xs, cs = [x, e], [cx, ce]
        if cn > 0:
            if variance_noise is None:
                variance_noise = torch.randn(e.shape, generator=generator, device=e.device, dtype=torch.float32)
            xs.append(_f32(variance_noise))
            cs.append(cn)
        prev = ops.lincomb(torch.empty_like(x), xs, cs).to(sample.dtype)
        if not return_dict:
            return (prev,)
        x0 = ops.lincomb(torch.empty_like(x), [x, e], [c0x, c0e]).to(sample.dtype)
        return SchedulerOutput(prev, x0)


class DDPMScheduler(_Base):
    """SURVEY.md Appendix A-11 (variance_type fixed_small, epsilon prediction)."""

    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                     thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0,
                     sample_max_value=1.0, timestep_spacing="leading", steps_offset=0)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if self.config.prediction_type != "epsilon" or self.config.variance_type != "fixed_small":
            raise NotImplementedError

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    def step_coefficients(self, t: int) -> Tuple[float, float, float]:
        n = self.num_inference_steps or self.config.num_train_timesteps
        tp = t - self.config.num_train_timesteps // n
        a = float(self._ac[t])
        ap = float(self._ac[tp]) if tp >= 0 else 1.0
        cur_a = a / ap
        cur_b = 1 - cur_a
        c0x, c0e = 1 / math.sqrt(a), -math.sqrt(1 - a) / math.sqrt(a)
        k0, kx = math.sqrt(ap) * cur_b / (1 - a), math.sqrt(cur_a) * (1 - ap) / (1 - a)
        sigma = math.sqrt(max((1 - ap) / (1 - a) * cur_b, 1e-20)) if t > 0 else 0.0
        return k0 * c0x + kx, k0 * c0e, sigma

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict: bool = True):
        if self.config.clip_sample:
            raise NotImplementedError("clip_sample=True is outside the stage-2 path")
        cx, ce, cn = self.step_coefficients(int(timestep))
        e, x = _f32(model_output), _f32(sample)
        xs, cs = [x, e], [cx, ce]
        if cn > 0:
            if variance_noise is None:
                variance_noise = torch.randn(e.shape, generator=generator, device=e.device, dtype=torch.float32)
            xs.append(_f32(variance_noise))
            cs.append(cn)
        prev = ops.lincomb(torch.empty_like(x), xs, cs).to(sample.dtype)
        return (prev,) if not return_dict else SchedulerOutput(prev)


class UniPCMultistepScheduler(_Base):
    """UniPC, B(h) = e^h - 1 ("bh2"), order 2, predict_x0, lower_order_final.

    ``step`` has no ``eta`` / ``generator`` parameters (the pipeline inspects the signature).  Stateful:
    one instance per in-flight sampling run."""

    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, solver_order=2, prediction_type="epsilon", thresholding=False,
                     dynamic_thresholding_ratio=0.995, sample_max_value=1.0, predict_x0=True, solver_type="bh2",
                     lower_order_final=True, disable_corrector=[], solver_p=None, use_karras_sigmas=False,
                     timestep_spacing="linspace", steps_offset=0)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if (c.prediction_type != "epsilon" or c.thresholding or not c.predict_x0 or c.use_karras_sigmas
                or c.solver_p is not None or c.solver_type not in ("bh1", "bh2")):
            raise NotImplementedError("UniPC variant outside the stage-2 path")
        self.set_timesteps(c.num_train_timesteps)

    def set_timesteps(self, num_inference_steps: int, device=None):
        c, T = self.config, self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ratio = T // (num_inference_steps + 1)
            ts = (np.arange(0, num_inference_steps + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + c.steps_offset
        else:
            raise NotImplementedError(c.timestep_spacing)
        ac = self.alphas_cumprod.numpy()
        sig = ((1 - ac) / ac) ** 0.5
        sigmas = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sigmas, [((1 - ac[0]) / ac[0]) ** 0.5]]).astype(np.float32)
        self.timesteps = torch.from_numpy(ts).to(device)
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * c.solver_order
        self.lower_order_nums = 0
        self.last_sample: Optional[torch.Tensor] = None
        self._step_index: Optional[int] = None
        self.this_order = 1
        self._ts_list = [int(v) for v in ts]

    @property
    def step_index(self):
        return self._step_index

    # ---- scalar pieces
    def _als(self, i: int) -> Tuple[np.float32, np.float32, np.float32]:
        s = np.float32(self.sigmas[i])
        alpha = np.float32(1.0) / np.sqrt(s * s + np.float32(1.0), dtype=np.float32)
        sig = s * alpha
        return np.log(alpha) - np.log(sig), alpha, sig

    def _rb(self, rks, hh, order):
        h_phi_1 = np.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.config.solver_type == "bh1" else np.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(np.power(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return np.stack(R), np.array(b), h_phi_1, B_h

    def predictor_coefficients(self, i: int, order: int):
        """x_next = cx*x + sum_k cm[k]*m_{-k}  (m_0 newest x0-prediction)."""
        lam_t, alpha_t, sigma_t = self._als(i + 1)
        lam_s, _, sigma_s = self._als(i)
        h = lam_t - lam_s
        rks = [(self._als(i - k)[0] - lam_s) / h for k in range(1, order)] + [1.0]
        R, b, h_phi_1, B_h = self._rb(np.array(rks, dtype=np.float32), -h, order)
        cm = [0.0] * order
        cm[0] = -float(alpha_t * h_phi_1)
        if order > 1:
            rhos = np.array([0.5]) if order == 2 else np.linalg.solve(R[:-1, :-1], b[:-1])
            for k in range(1, order):
                cf = -float(alpha_t * B_h) * float(rhos[k - 1]) / float(rks[k - 1])
                cm[k] += cf      # D1_k = (m_{-k} - m_0) / r_k
                cm[0] -= cf
        return float(sigma_t / sigma_s), cm

    def corrector_coefficients(self, i: int, order: int):
        """x_corrected = cl*last_sample + sum_k cm[k]*m_{-k} + ct*model_t (m_0 = newest stored, before model_t)."""
        lam_t, alpha_t, sigma_t = self._als(i)
        lam_s, _, sigma_s = self._als(i - 1)
        h = lam_t - lam_s
        rks = [(self._als(i - (k + 1))[0] - lam_s) / h for k in range(1, order)] + [1.0]
        R, b, h_phi_1, B_h = self._rb(np.array(rks, dtype=np.float32), -h, order)
        rhos = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
        cm = [0.0] * order
        cm[0] = -float(alpha_t * h_phi_1)
        for k in range(1, order):
            cf = -float(alpha_t * B_h) * float(rhos[k - 1]) / float(rks[k - 1])
            cm[k] += cf
            cm[0] -= cf
        ct = -float(alpha_t * B_h) * float(rhos[-1])
        cm[0] -= ct
        return float(sigma_t / sigma_s), cm, ct

    def coefficient_table(self, device=None) -> torch.Tensor:
        """fp32 [n, 12] rows for ``pcdm_unipc_step`` (include/pcdm.h): the scalars ``step`` would compute at step i of a fresh run --
        the order bookkeeping (``lower_order_nums``, ``this_order``, ``lower_order_final``) depends on i and n only, so the whole
        multistep update is a per-step linear map on (x, eps, m1, m2, last_sample) with host-known coefficients."""
        c, n = self.config, len(self._ts_list)
        if c.solver_order > 2:
            raise NotImplementedError("fused UniPC: solver_order <= 2")
        rows = np.zeros((n, 12), dtype=np.float64)
        lower, this_order = 0, 1
        for i in range(n):
            _, alpha_t, sigma_t = self._als(i)
            rows[i, 0], rows[i, 1] = 1.0 / float(alpha_t), -float(sigma_t) / float(alpha_t)
            if i > 0 and (i - 1) not in c.disable_corrector:
                cl, cm, ct = self.corrector_coefficients(i, this_order)
                rows[i, 2], rows[i, 3], rows[i, 6] = 1.0, cl, ct
                rows[i, 4:4 + len(cm)] = cm
            order = min(c.solver_order, n - i) if c.lower_order_final else c.solver_order
            this_order = min(order, lower + 1)
            cx, cm = self.predictor_coefficients(i, this_order)
            rows[i, 7] = cx
            rows[i, 8:8 + len(cm)] = cm
            if lower < c.solver_order:
                lower += 1
        return torch.from_numpy(rows.astype(np.float32)).to(device)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            self._step_index = self._ts_list.index(int(timestep))
        i = self._step_index
        c = self.config
        e, x = _f32(model_output), _f32(sample)
        _, alpha_t, sigma_t = self._als(i)
        m_t = ops.lincomb(torch.empty_like(x), [x, e], [1.0 / float(alpha_t), -float(sigma_t) / float(alpha_t)])
        use_corrector = i > 0 and (i - 1) not in c.disable_corrector and self.last_sample is not None
        if use_corrector:
            cl, cm, ct = self.corrector_coefficients(i, self.this_order)
            xs = [self.last_sample] + [self.model_outputs[-(k + 1)] for k in range(self.this_order)] + [m_t]
            x = ops.lincomb(torch.empty_like(x), xs, [cl] + cm + [ct])
        for k in range(c.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = m_t
        order = min(c.solver_order, len(self._ts_list) - i) if c.lower_order_final else c.solver_order
        self.this_order = min(order, self.lower_order_nums + 1)
        self.last_sample = x
        cx, cm = self.predictor_coefficients(i, self.this_order)
        xs = [x] + [self.model_outputs[-(k + 1)] for k in range(self.this_order)]
        prev = ops.lincomb(torch.empty_like(x), xs, [cx] + cm).to(sample.dtype)
        if self.lower_order_nums < c.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return (prev,) if not return_dict else SchedulerOutput(prev)


class UnCLIPScheduler(_Base):
    """Stand-in for diffusers 0.24.0 ``UnCLIPScheduler`` at the stage-1 prior's call sites
    (/root/reference/src/pipelines/stage1_prior_pipeline.py:439-440 ``set_timesteps`` / ``timesteps``, :279
    ``init_noise_sigma``, :478-483 ``step(pred, timestep=t, sample=latents, prev_timestep=...)``).
    Class defaults are diffusers'; the Kandinsky-2.2 prior's ``scheduler_config.json`` (prediction_type "sample",
    clip_sample_range 10, "fixed_small_log") arrives through ``from_config`` / ``from_pretrained``.
    Scalar coefficient math on the host (float64), the tensor update is one HIP kernel (``pcdm_unclip_step``)."""

    _defaults = dict(num_train_timesteps=1000, variance_type="fixed_small_log", clip_sample=True, clip_sample_range=1.0,
                     prediction_type="epsilon", beta_schedule="squaredcos_cap_v2", trained_betas=None)
    KANDINSKY22_PRIOR = dict(num_train_timesteps=1000, variance_type="fixed_small_log", clip_sample=True,
                             clip_sample_range=10.0, prediction_type="sample", beta_schedule="squaredcos_cap_v2")

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if c.variance_type != "fixed_small_log" or c.prediction_type not in ("epsilon", "sample"):
            raise NotImplementedError("UnCLIP variant outside the stage-1 path (learned_range variance / v-prediction)")
        self._betas64 = self.betas.numpy().astype(np.float64)

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = (self.config.num_train_timesteps - 1) / (num_inference_steps - 1)
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device)

    def step_coefficients(self, t: int, prev_t: Optional[int] = None):
        """(p_x, p_e, clip, c_x0, c_x, c_noise): x0 = clamp(p_x x + p_e pred, +-clip); prev = c_x0 x0 + c_x x + c_noise z."""
        c = self.config
        if prev_t is None:
            prev_t = t - 1
        a_t = float(self._ac[t])
        a_prev = float(self._ac[prev_t]) if prev_t >= 0 else 1.0
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if prev_t == t - 1:
            beta = float(self._betas64[t])
            alpha = 1 - beta
        else:
            beta = 1 - a_t / a_prev
            alpha = 1 - beta
        if c.prediction_type == "epsilon":
            p_x, p_e = 1 / math.sqrt(a_t), -math.sqrt(1 - a_t) / math.sqrt(a_t)
        else:
            p_x, p_e = 0.0, 1.0
        clip = float(c.clip_sample_range) if c.clip_sample else 0.0
        std = math.exp(0.5 * math.log(max(b_prev / b_t * beta, 1e-20))) if t > 0 else 0.0
        return p_x, p_e, clip, math.sqrt(a_prev) * beta / b_t, math.sqrt(alpha) * b_prev / b_t, std

    def step(self, model_output, timestep, sample, prev_timestep=None, generator=None, variance_noise=None,
             return_dict: bool = True):
        t = int(timestep)
        co = self.step_coefficients(t, None if prev_timestep is None else int(prev_timestep))
        e, x = _f32(model_output), _f32(sample)
        noise = None
        if co[5] > 0:
            if variance_noise is None:
                variance_noise = torch.randn(e.shape, generator=generator, dtype=torch.float32,
                                             device=generator.device if generator is not None else e.device)
            noise = _f32(variance_noise.to(e.device))
        prev = ops.unclip_step(e, False, 0.0, x, noise, torch.empty_like(x), (*co, 1.0, 0.0)).to(sample.dtype)
        return (prev,) if not return_dict else SchedulerOutput(prev)
