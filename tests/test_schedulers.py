"""Schedulers: oracle known answers and product (HIP tensor updates) vs oracle."""
from __future__ import annotations

import inspect

import numpy as np
import pytest
import torch

from oracle.schedulers import DDIMOracle, DDPMOracle, UniPCOracle
from pcdms_amd.schedulers import DDIMScheduler, DDPMScheduler, UniPCMultistepScheduler

SD21 = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
            clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon",
            skip_prk_steps=True, trained_betas=None)  # SD-2.1-base scheduler_config.json (PNDM keys are ignored)


def test_oracle_known_answers():
    d = DDIMOracle()
    ac = d.alphas_cumprod
    kat = {0: 0.9991499782, 1: 0.9982960224, 20: 0.9813143015, 21: 0.9803806543, 500: 0.2763324678,
           961: 0.0072817220, 981: 0.0057754959, 999: 0.0046600951}
    for i, v in kat.items():
        assert abs(float(ac[i]) - v) < 2e-7
    d.set_timesteps(50)
    assert d.timesteps.tolist() == list(range(981, 0, -20))
    x, e = torch.tensor([0.5, -1, 2.0]), torch.tensor([0.1, 0.2, -0.3])
    a = float(ac[981])
    x0 = (x - np.sqrt(1 - a) * e) / np.sqrt(a)
    assert torch.allclose(x0, torch.tensor([5.2671928, -15.7825527, 30.2530613]), atol=2e-4)
    assert torch.allclose(d.step(e, 981, x), torch.tensor([0.5491006, -1.1475022, 2.2826788]), atol=1e-5)
    d.set_timesteps(20)
    assert d.timesteps.tolist() == list(range(951, 0, -50))
    u = UniPCOracle()
    u.set_timesteps(20)
    assert u.timesteps.tolist() == [999, 949, 899, 849, 799, 749, 699, 649, 599, 549, 500, 450, 400, 350, 300, 250,
                                    200, 150, 100, 50]
    assert np.allclose(u.sigmas[:3], [14.614647, 10.90424, 8.302806], rtol=1e-6)
    assert np.allclose(u.sigmas[-3:], [0.34393182, 0.22558255, 0.02916753], rtol=1e-6)


def test_oracle_unipc_solves_linear_ode():
    """With an exact eps-model for a point-mass data distribution the x0-prediction is constant and every
    exponential-integrator order is exact: x_final = a_last*c + (sigma ratio)*(x_T - a_T c)."""
    u = UniPCOracle()
    u.set_timesteps(20)
    ac = u.alphas_cumprod
    c = torch.tensor([0.7, -0.3])
    xT = torch.tensor([1.5, -0.25])
    x = xT.clone()
    for t in u.timesteps:
        a = float(ac[int(t)])
        x = u.step((x - np.sqrt(a) * c) / np.sqrt(1 - a), t, x)
    aT, a0 = float(ac[999]), float(ac[0])
    expect = np.sqrt(a0) * c + np.sqrt(1 - a0) / np.sqrt(1 - aT) * (xT - np.sqrt(aT) * c)
    assert torch.allclose(x, expect.float(), atol=2e-4)


def test_step_signatures_match_what_the_pipeline_inspects():
    """ref stage2_inpaint_pipeline.py:313-321 passes eta/generator only if step() names them."""
    assert {"eta", "generator"} <= set(inspect.signature(DDIMScheduler.step).parameters)
    p = set(inspect.signature(UniPCMultistepScheduler.step).parameters)
    assert "eta" not in p and "generator" not in p
    assert "generator" in inspect.signature(DDPMScheduler.step).parameters
    s = UniPCMultistepScheduler.from_config(SD21)
    assert s.config.steps_offset == 1 and s.config.timestep_spacing == "linspace" and s.order == 1
    assert s.init_noise_sigma == 1.0


def test_host_tables_match_oracle():
    s = DDIMScheduler.from_config(SD21)
    s.set_timesteps(50)
    o = DDIMOracle()
    o.set_timesteps(50)
    assert s.timesteps.tolist() == o.timesteps.tolist()
    cx, ce, cn, c0x, c0e = s.step_coefficients(981)
    x, e = torch.tensor([0.5, -1, 2.0]), torch.tensor([0.1, 0.2, -0.3])
    assert torch.allclose(cx * x + ce * e, o.step(e, 981, x), atol=1e-6) and cn == 0.0
    u = UniPCMultistepScheduler.from_config(SD21)
    u.set_timesteps(20)
    uo = UniPCOracle()
    uo.set_timesteps(20)
    assert u.timesteps.tolist() == uo.timesteps.tolist() and np.array_equal(u.sigmas, uo.sigmas)


def test_schedulers_vs_oracle(backend):
    dev = backend.device
    g = torch.Generator().manual_seed(0)
    shape = (2, 4, 6, 10)
    for n in (5, 20):
        prods = [DDIMScheduler.from_config(SD21), UniPCMultistepScheduler.from_config(SD21)]
        oras = [DDIMOracle(), UniPCOracle()]
        for s, o in zip(prods, oras):
            s.set_timesteps(n, device=dev)
            o.set_timesteps(n)
            x = torch.randn(shape, generator=g)
            xo = x.clone()
            xd = x.to(dev)
            for t in s.timesteps:
                e = torch.randn(shape, generator=g) * 0.5 + 0.3 * xo
                xo = o.step(e, int(t), xo)
                xd = s.step(e.to(dev), t, xd, return_dict=False)[0]
            backend.sync()
            assert torch.allclose(xd.cpu(), xo, rtol=2e-4, atol=2e-4), type(s).__name__
    # DDIM eta > 0 and DDPM ancestral with injected noise
    s, o = DDIMScheduler.from_config(SD21), DDIMOracle()
    s.set_timesteps(10, device=dev)
    o.set_timesteps(10)
    x, e, z = (torch.randn(shape, generator=g) for _ in range(3))
    a = s.step(e.to(dev), 901, x.to(dev), eta=0.7, variance_noise=z.to(dev)).prev_sample
    assert torch.allclose(a.cpu(), o.step(e, 901, x, eta=0.7, variance_noise=z), atol=1e-5)
    s2 = DDPMScheduler(**{**{k: v for k, v in SD21.items() if k in DDPMScheduler._defaults}, "clip_sample": False})
    o2 = DDPMOracle()
    s2.set_timesteps(10, device=dev)
    o2.set_timesteps(10)
    b = s2.step(e.to(dev), 900, x.to(dev), variance_noise=z.to(dev)).prev_sample
    assert torch.allclose(b.cpu(), o2.step(e, 900, x, variance_noise=z), atol=1e-5)
    ts = torch.tensor([10, 900])
    an = s2.add_noise(x.to(dev), z.to(dev), ts)
    assert torch.allclose(an.cpu(), o2.add_noise(x, z, ts), atol=1e-6)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 20])
def test_unipc_fused_step_kernel_equals_step(backend, n):
    """``pcdm_unipc_step`` on static history slots + ``coefficient_table`` (the graph-captured form of the shipped driver's
    scheduler, ref stage2_batchtest_inpaint_model.py:132) == the stateful ``UniPCMultistepScheduler.step`` over whole runs of
    1 .. 20 steps (order ramp-up 1 -> 2, corrector from step 1, ``lower_order_final`` at the end), with CFG folded in."""
    from pcdms_amd import ops
    dev = backend.device
    g = torch.Generator().manual_seed(5 + n)
    shape = (2, 4, 4, 6)
    x0 = torch.randn(shape, generator=g)
    eps_all = [torch.randn((4, 4, 4, 6), generator=g) for _ in range(n)]     # [uncond | cond] halves
    gs = 2.0
    ref_s = UniPCMultistepScheduler.from_config(SD21)
    ref_s.set_timesteps(n)
    x = x0.to(dev)
    for i, t in enumerate(ref_s.timesteps):
        e = eps_all[i].to(dev)
        guided = e[:2] + gs * (e[2:] - e[:2])
        x = ref_s.step(guided.contiguous(), t, x, return_dict=False)[0]
    fs = UniPCMultistepScheduler.from_config(SD21)
    fs.set_timesteps(n)
    coef = fs.coefficient_table(device=dev)
    assert coef.shape == (n, 12)
    lat = x0.clone().to(dev)
    m1, m2, last = (torch.zeros(shape, device=dev) for _ in range(3))
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    for i in range(n):
        ops.unipc_step(eps_all[i].to(dev).contiguous(), True, gs, lat, m1, m2, last, coef, step)
        ops.advance_step(step)
    backend.sync()
    assert torch.allclose(lat.cpu(), x.cpu(), rtol=2e-5, atol=2e-5 * float(x.abs().max())), (lat.cpu() - x.cpu()).abs().max()
