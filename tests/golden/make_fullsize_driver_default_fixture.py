#!/usr/bin/env python3
"""Full-size oracle run of the configuration the reference's stage-2 driver SHIPS WITH
(/root/reference/stage2_batchtest_inpaint_model.py:132 UniPCMultistepScheduler, :196 num_images_per_prompt = 4,
:256-260 guidance_scale 2.0 / num_inference_steps 20 / img 512x512): canvas 1024x512 => latent 64x128, UNet batch 8
(M = 65 536 rows at level 0, self-attention over N = 8192 tokens), 20 UniPC steps (bh2, order 2, lower_order_final).

The fp32 oracle (oracle/pipeline.py::stage2_sample over oracle/unet.py + oracle/schedulers.py::UniPCOracle) runs the COMPLETE
call on the seeded 868.9 M-parameter weights of the other full-size fixtures
(``synth_state_dict(UNetConfig(), seed=0, random_affine=True)``) and ``synth_inputs(cfg, 64, 128, 4)``.  Stored: the latents
before steps 0 / 1 / 5 / 10 / 19 (fp32), the guided eps of steps 0 / 10 / 19 (fp16), the final latents.  ~310 TFLOP of fp32:
about 25 min on the 8 build-container cores.

    python tests/golden/make_fullsize_driver_default_fixture.py
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
STEPS, N, H, W = 20, 4, 64, 128
CHECK = (0, 1, 5, 10, 19)
EPS_AT = (0, 10, 19)


def main():
    from oracle.pipeline import stage2_sample, synth_inputs
    from oracle.schedulers import UniPCOracle
    from oracle.unet import UNetConfig, synth_state_dict
    t0 = time.time()
    cfg = UNetConfig()
    sd = synth_state_dict(cfg, seed=0, random_affine=True)
    inp = synth_inputs(cfg, H, W, N)
    out = {"torch_version": np.array(torch.__version__), "steps": np.array(STEPS), "check": np.array(CHECK),
           "eps_at": np.array(EPS_AT)}

    def hook(i, t, eps, lat):
        print(f"step {i:2d} t={t:4d} |eps| {eps.norm():.3f} |lat| {lat.norm():.3f}  ({time.time() - t0:.0f} s)", flush=True)
        if i in CHECK:
            out[f"lat_{i}"] = lat.numpy().copy()
        if i in EPS_AT:
            out[f"eps_{i}"] = eps.numpy().astype(np.float16)

    with torch.no_grad():
        lat = stage2_sample(sd, cfg, UniPCOracle(), num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=STEPS,
                            eps_hook=hook, **inp)
    out["lat_final"] = lat.numpy().copy()
    path = ROOT / "tests" / "golden" / "fullsize_driver_default.npz"
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({path.stat().st_size / 1e6:.2f} MB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
