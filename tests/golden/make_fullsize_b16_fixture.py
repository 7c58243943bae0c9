#!/usr/bin/env python3
"""Second full-size oracle state for the per-GPU share of BASELINE.json configs[2] (8 pairs' worth of samples per GPU: N = 8,
UNet batch 16): the guided eps of ONE forward in the MIDDLE of the schedule (step 25 of 50, t = 491), complementing
``fullsize_config2.npz``'s ``b16_eps`` (step 0, pure-noise input).

Input latents: ``sqrt(a_t) x0 + sqrt(1 - a_t) n`` with seeded ``x0 ~ N(0, 0.9^2)`` (seed 21) and ``n ~ N(0,1)`` (seed 22) --
the marginal a DDIM trajectory has at that step (ref: DDPMScheduler.add_noise).  Same seeded
868.9 M-parameter weights and ``synth_inputs(cfg, 64, 88, 8)`` conditioning as the other full-size fixtures.  The fp32 oracle
(oracle/unet.py; restates /root/reference/src/models/stage2_inpaint_unet_2d_condition.py:579-825) runs the 16 batch rows in
chunks of 4 (rows are independent).  ~2-4 min on the 8 build-container cores.

    python tests/golden/make_fullsize_b16_fixture.py
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
STEP, STEPS, N = 25, 50, 8


def mid_state_latents(alpha_t: float, N: int, h: int, w: int) -> torch.Tensor:
    x0 = torch.randn(N, 4, h, w, generator=torch.Generator().manual_seed(21)) * 0.9
    n = torch.randn(N, 4, h, w, generator=torch.Generator().manual_seed(22))
    return alpha_t ** 0.5 * x0 + (1 - alpha_t) ** 0.5 * n


def main():
    from oracle.pipeline import build_conditioning, synth_inputs
    from oracle.schedulers import DDIMOracle
    from oracle.unet import UNetConfig, synth_state_dict, unet_forward
    t0 = time.time()
    cfg = UNetConfig()
    h, w = 64, 88
    sd = synth_state_dict(cfg, seed=0, random_affine=True)
    inp = synth_inputs(cfg, h, w, N)
    c = build_conditioning(inp["masked_latents"], inp["s_img_proj_f"], inp["st_pose_f"], inp["pred_t_img_embed"], N, True)
    sch = DDIMOracle()
    sch.set_timesteps(STEPS)
    t = int(sch.timesteps[STEP])
    lat = mid_state_latents(float(sch.alphas_cumprod[t]), N, h, w)
    x = torch.cat([torch.cat([lat] * 2), c["mask"], c["masked_latents"]], 1)
    pose = c["pose_cond"]
    outs = []
    with torch.no_grad():
        for r in range(0, 2 * N, 4):
            sl = slice(r, r + 4)
            outs.append(unet_forward(sd, cfg, x[sl], torch.tensor(t), c["feature_f"][sl], c["prior_embed"][sl],
                                     pose if pose.shape[0] == 1 else pose[sl]))
            print(f"rows {r}..{r + 3} done ({time.time() - t0:.0f} s)", flush=True)
    eps = torch.cat(outs)
    u, cn = eps.chunk(2)
    g = u + 2.0 * (cn - u)
    out = ROOT / "tests" / "golden" / "fullsize_b16_mid.npz"
    np.savez_compressed(out, torch_version=np.array(torch.__version__), step=STEP, steps=STEPS, t=t, N=N,
                        eps=g.numpy().astype(np.float16), lat_checksum=float(lat.double().abs().sum()))
    print(f"wrote {out} ({out.stat().st_size / 1e6:.2f} MB) in {time.time() - t0:.0f} s; |eps| {g.norm():.3f} std {g.std():.4f}")


if __name__ == "__main__":
    main()
