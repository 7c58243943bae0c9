#!/usr/bin/env python3
"""Full-size, full-length parity fixture for BASELINE.json configs[1] ("stage2 inpaint, 352x512, batch=4,
50 DDIM steps") and the per-GPU share of configs[2] (batch 8 => UNet batch 16, one forward).

Runs the fp32 CPU oracle (oracle/ -- the restatement of /root/reference/src/pipelines/stage2_inpaint_pipeline.py:
494-532 and src/models/stage2_inpaint_unet_2d_condition.py:579-825) on the SEEDED full-size synthetic weights
(868.9 M parameters, `synth_state_dict(UNetConfig(), seed=0, random_affine=True)`) and seeded inputs
(`synth_inputs(cfg, 64, 88, 4)`), 50 DDIM steps, guidance 2.0, and stores

  lat_<i>   fp32 [4,4,64,88]   latents BEFORE step i, i in CHECK (and `lat_final` after step 49)
  eps_<i>   fp16 [4,4,64,88]   guided eps of step i
  img_0/3   uint8 [512,704,3]  canvases 0 and 3 after oracle/vae.py decode (full SD-2.1 VAE topology, seeded weights) +
                               VaeImageProcessor.postprocess;  img_mean / img_std of all four canvases
  b16_eps   fp16 [8,4,64,88]   guided eps of ONE forward at N = 8 (UNet batch 16; configs[2]'s per-GPU share), step 0

Everything the GPU box needs is in the .npz (weights / inputs are regenerated there from the same seeds with the same
torch build; the torch version is recorded).  ~35 min on the 8 build-container cores (475 TFLOP of fp32 at ~0.22 TFLOP/s).

    python tests/golden/make_fullsize_config2_fixture.py [--steps 50] [--out tests/golden/fullsize_config2.npz]
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

CHECK = (0, 10, 25, 49)
SEED_W, SEED_VAE = 0, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden" / "fullsize_config2.npz"))
    ap.add_argument("--no-b16", action="store_true")
    args = ap.parse_args()

    from oracle.pipeline import build_conditioning, stage2_sample, synth_inputs
    from oracle.schedulers import DDIMOracle
    from oracle.unet import UNetConfig, synth_state_dict, unet_forward
    from oracle import vae as ovae

    torch.manual_seed(0)
    cfg = UNetConfig()
    h, w, N = 64, 88, 4
    t0 = time.time()
    sd = synth_state_dict(cfg, seed=SEED_W, random_affine=True)
    inp = synth_inputs(cfg, h, w, N)
    print(f"weights + inputs: {time.time() - t0:.1f} s", flush=True)
    out = {"torch_version": np.array(torch.__version__), "steps": np.array(args.steps), "check": np.array(CHECK)}

    def hook(i, t, eps, lat):
        print(f"step {i:2d} t={t:4d} |eps| {eps.norm():.3f} |lat| {lat.norm():.3f}  ({time.time() - t0:.0f} s)", flush=True)
        if i in CHECK:
            out[f"lat_{i}"] = lat.numpy().copy()
            out[f"eps_{i}"] = eps.numpy().astype(np.float16)

    with torch.no_grad():
        lat = stage2_sample(sd, cfg, DDIMOracle(), num_images_per_prompt=N, guidance_scale=2.0,
                            num_inference_steps=args.steps, eps_hook=hook, **inp)
        out["lat_final"] = lat.numpy().copy()
        # ---- VAE decode of the final latents (ref :528-532)
        vcfg = ovae.VAEConfig()
        vsd = ovae.synth_state_dict(vcfg, SEED_VAE)
        imgs = []
        for k in range(N):
            img = ovae.decode(vsd, vcfg, lat[k:k + 1] / vcfg.scaling_factor)
            imgs.append(ovae.postprocess_uint8(img)[0])
            print(f"decoded canvas {k} ({time.time() - t0:.0f} s)", flush=True)
        u8 = torch.stack(imgs).numpy()
        out["img_0"], out["img_3"] = u8[0], u8[3]
        out["img_mean"] = u8.reshape(N, -1).astype(np.float64).mean(1)
        out["img_std"] = u8.reshape(N, -1).astype(np.float64).std(1)
        # ---- configs[2] per-GPU share: one forward at N = 8 (UNet batch 16), step 0
        if not args.no_b16:
            N8 = 8
            inp8 = synth_inputs(cfg, h, w, N8)
            c = build_conditioning(inp8["masked_latents"], inp8["s_img_proj_f"], inp8["st_pose_f"], inp8["pred_t_img_embed"],
                                   N8, True)
            sch = DDIMOracle()
            sch.set_timesteps(args.steps)
            t = sch.timesteps[0]
            x = torch.cat([inp8["latents"]] * 2)
            eps = unet_forward(sd, cfg, torch.cat([x, c["mask"], c["masked_latents"]], 1), t, c["feature_f"], c["prior_embed"],
                               c["pose_cond"])
            u, cn = eps.chunk(2)
            out["b16_eps"] = (u + 2.0 * (cn - u)).numpy().astype(np.float16)
            print(f"batch-16 forward done ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(args.out, **out)
    print(f"wrote {args.out} ({Path(args.out).stat().st_size / 1e6:.2f} MB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
