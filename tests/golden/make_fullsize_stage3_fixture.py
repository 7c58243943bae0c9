#!/usr/bin/env python3
"""Full-size oracle states for the STAGE-3 refinement UNet (BASELINE.json configs[3]'s third stage): the stock
865.9 M-parameter SD-2.1 topology with ``in_channels = 8`` and no class embedding / pose
(/root/reference/stage3_batchtest_refined_model.py:121-126), latent 64 x 44 (one 512 x 352 image -- NOT divisible by 8: levels
64x44 -> 32x22 -> 16x11 -> 8x6, the odd-size stride-2 / upsample-to-skip-size path of
/root/reference/src/models/stage2_inpaint_unet_2d_condition.py:625-633), N = 8 samples under CFG (UNet batch 16).

Contents (made by the fp32 oracle ``oracle/unet.py`` + ``oracle/pipeline.py::stage3_sample``, which restate
/root/reference/src/pipelines/stage3_refined_pipeline.py:483-557):
  * ``eps_0``   guided eps of ONE forward at step 0 of a 20-step DDIM schedule (pure-noise latents), N = 8;
  * ``eps_mid`` the same at step 10 (latents = the schedule's marginal at that step, seeded), N = 8;
  * ``lat_final_n2`` the final latents of a complete 20-step DDIM run with N = 2 (the first two of the N = 8 noise samples): the
    GPU test runs N = 8 under the hipGraph and its first two samples must reproduce this trajectory (samples are independent).
Seeded inputs: weights ``synth_state_dict(cfg3, seed=0, random_affine=True)``; ``synth_stage3_inputs`` below.  ~10 min on 8 cores.

    python tests/golden/make_fullsize_stage3_fixture.py
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
H, W, N, STEPS, MID = 64, 44, 8, 20, 10


def stage3_config():
    from oracle.unet import UNetConfig
    return UNetConfig(in_channels=8, class_embed_type=None, projection_class_embeddings_input_dim=None)


def synth_stage3_inputs(n: int = N):
    g = lambda s: torch.Generator().manual_seed(s)   # noqa: E731
    return dict(latents=torch.randn(n, 4, H, W, generator=g(41)),
                gen_t_img_latents=torch.randn(1, 4, H, W, generator=g(42)) * 0.18215 * 5,
                s_img_proj_f=torch.randn(1, 257, 1024, generator=g(43)))


def mid_latents(alpha_t: float, n: int = N) -> torch.Tensor:
    x0 = torch.randn(n, 4, H, W, generator=torch.Generator().manual_seed(44)) * 0.9
    nz = torch.randn(n, 4, H, W, generator=torch.Generator().manual_seed(45))
    return alpha_t ** 0.5 * x0 + (1 - alpha_t) ** 0.5 * nz


def guided_eps(sd, cfg, inp, lat, t, chunk=4):
    from oracle.unet import unet_forward
    n = lat.shape[0]
    feat = inp["s_img_proj_f"].repeat(n, 1, 1)
    gl = inp["gen_t_img_latents"].repeat(n, 1, 1, 1)
    feat = torch.cat([torch.zeros_like(feat), feat])
    gl = torch.cat([torch.zeros_like(gl), gl])
    x = torch.cat([torch.cat([lat] * 2), gl], 1)
    outs = []
    with torch.no_grad():
        for r in range(0, 2 * n, chunk):
            outs.append(unet_forward(sd, cfg, x[r:r + chunk], torch.tensor(int(t)), feat[r:r + chunk], None, None))
    u, c = torch.cat(outs).chunk(2)
    return u + 2.0 * (c - u)


def main():
    from oracle.pipeline import stage3_sample
    from oracle.schedulers import DDIMOracle
    from oracle.unet import synth_state_dict
    t0 = time.time()
    cfg = stage3_config()
    sd = synth_state_dict(cfg, seed=0, random_affine=True)
    inp = synth_stage3_inputs()
    sch = DDIMOracle()
    sch.set_timesteps(STEPS)
    e0 = guided_eps(sd, cfg, inp, inp["latents"] * sch.init_noise_sigma, sch.timesteps[0])
    print(f"eps_0 done ({time.time() - t0:.0f} s)", flush=True)
    tm = int(sch.timesteps[MID])
    em = guided_eps(sd, cfg, inp, mid_latents(float(sch.alphas_cumprod[tm])), tm)
    print(f"eps_mid done ({time.time() - t0:.0f} s)", flush=True)
    with torch.no_grad():
        fin = stage3_sample(sd, cfg, DDIMOracle(), gen_t_img_latents=inp["gen_t_img_latents"], s_img_proj_f=inp["s_img_proj_f"],
                            latents=inp["latents"][:2], num_images_per_prompt=2, guidance_scale=2.0, num_inference_steps=STEPS)
    out = ROOT / "tests" / "golden" / "fullsize_stage3.npz"
    np.savez_compressed(out, torch_version=np.array(torch.__version__), steps=STEPS, mid=MID, t_mid=tm, N=N,
                        eps_0=e0.numpy().astype(np.float16), eps_mid=em.numpy().astype(np.float16), lat_final_n2=fin.numpy(),
                        lat_checksum=float(inp["latents"].double().abs().sum()))
    print(f"wrote {out} ({out.stat().st_size / 1e6:.2f} MB) in {time.time() - t0:.0f} s; |eps_0| {e0.norm():.3f} |eps_mid| {em.norm():.3f}")


if __name__ == "__main__":
    main()
