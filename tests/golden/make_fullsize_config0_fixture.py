#!/usr/bin/env python3
"""Full-size, full-length oracle run of BASELINE.json configs[0] -- the reference's own CPU-runnable case: ONE 256x256 pair
(canvas 512x256, latent 32x64), ``num_images_per_prompt = 1``, 20 DDIM steps, guidance 2.0, fp32 on the CPU
(/root/reference/stage2_batchtest_inpaint_model.py with ``--num_inference_steps 20`` at the 256 resolution the checkpoints
``s2_256.pt`` are for; loop: src/pipelines/stage2_inpaint_pipeline.py:494-525).

The fp32 oracle (oracle/pipeline.py::stage2_sample over oracle/unet.py) runs the COMPLETE call on the seeded 868.9 M-parameter
weights (``synth_state_dict(UNetConfig(), seed=0, random_affine=True)`` -- the weights of the other full-size fixtures) and the
seeded inputs ``synth_inputs(cfg, 32, 64, 1)``; stored: the latents before steps 5 / 10 / 15, the final latents, the guided eps
of steps 0 / 10 / 19 (fp16).  80 s on the 8 build-container cores.

    python tests/golden/make_fullsize_config0_fixture.py
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
STEPS, N, H, W = 20, 1, 32, 64
CHECK = (0, 5, 10, 15, 19)


def main():
    from oracle.pipeline import stage2_sample, synth_inputs
    from oracle.schedulers import DDIMOracle
    from oracle.unet import UNetConfig, synth_state_dict
    t0 = time.time()
    cfg = UNetConfig()
    sd = synth_state_dict(cfg, seed=0, random_affine=True)
    inp = synth_inputs(cfg, H, W, N)
    out = {"torch_version": np.array(torch.__version__), "steps": np.array(STEPS), "check": np.array(CHECK), "lat_0": inp["latents"].numpy().copy()}

    def hook(i, t, eps, lat):
        print(f"step {i:2d} t={t:4d} |eps| {eps.norm():.3f} |lat| {lat.norm():.3f}  ({time.time() - t0:.0f} s)", flush=True)
        if i in CHECK:
            out[f"lat_{i}"] = lat.numpy().copy()
            out[f"eps_{i}"] = eps.numpy().astype(np.float16)

    with torch.no_grad():
        lat = stage2_sample(sd, cfg, DDIMOracle(), num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=STEPS,
                            eps_hook=hook, **inp)
    out["lat_final"] = lat.numpy().copy()
    path = ROOT / "tests" / "golden" / "fullsize_config0.npz"
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({path.stat().st_size / 1e3:.0f} kB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
