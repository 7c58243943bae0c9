#!/usr/bin/env python3
"""A REAL-IMAGE fixture of the stage-2 path: every other full-size fixture feeds Gaussian latents and a constant-zero
masked-latent target half.  Here the inputs are real pictures -- the reference's own sample source image and pose maps
(/root/reference/imgs/img1.png, pose1.png, pose2.png; build container only) -- taken through the driver's canvas preparation
(/root/reference/stage2_batchtest_inpaint_model.py:150-174):

    s_img = img1 -> RGB, 256 x 256 BICUBIC;  [s_img | black] canvas 512 x 256;  [s_pose | t_pose] canvas;  ToTensor + Normalize(0.5, 0.5)
    masked_latents = vae.encode(canvas).latent_dist.sample() * scaling_factor        (src/pipelines/stage2_inpaint_pipeline.py:443-444;
                     posterior noise INJECTED so that the run is reproducible: the right half is the VAE's code of black, not the constant 0)
    st_pose_f = pose_proj(pose canvas)                                               (ControlNetConditioningEmbedding, :173-174)

then 3 DDIM steps of the full-size UNet (868.9 M seeded parameters, N = 1, guidance 2.0, latent 32 x 64), VAE decode, uint8 -- all by the
fp32 CPU oracle.  No checkpoint exists offline: VAE / pose net / UNet weights are the seeded synthetic ones of the other fixtures, the DINOv2 /
stage-1 conditioning is the seeded synthetic one too; what is real is the IMAGE STATISTICS entering the VAE, the pose net and through them the
UNet (smooth regions, hard edges, a constant black half, sparse stick figures on black).

Only ARRAYS travel (uint8 canvases, latents, eps, pixels): no PNG, no reference text.  tests/test_fullsize_parity.py::test_real_image_inputs
compares the HIP chain (VAE encode -> pose net -> 3 sampling steps -> decode) with them on the GPU.

    python tests/golden/make_real_image_fixture.py        (build container; ~1 min)
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REF_IMGS = Path("/root/reference/imgs")
H = W = 256
STEPS, N = 3, 1
SEED_POSE_NET = 11


def canvases():
    """the driver's canvas preparation (:150-162) on the reference's sample images -> uint8 [256, 512, 3] x 2"""
    from PIL import Image
    s_img = Image.open(REF_IMGS / "img1.png").convert("RGB").resize((W, H), Image.BICUBIC)
    black = Image.new("RGB", s_img.size, (0, 0, 0))
    cv = Image.new("RGB", (s_img.width * 2, s_img.height))
    cv.paste(s_img, (0, 0))
    cv.paste(black, (s_img.width, 0))
    s_pose = Image.open(REF_IMGS / "pose1.png").convert("RGB").resize((W, H), Image.BICUBIC)
    t_pose = Image.open(REF_IMGS / "pose2.png").convert("RGB").resize((W, H), Image.BICUBIC)
    pc = Image.new("RGB", (s_pose.width * 2, s_pose.height))
    pc.paste(s_pose, (0, 0))
    pc.paste(t_pose, (s_pose.width, 0))
    return np.asarray(cv, dtype=np.uint8).copy(), np.asarray(pc, dtype=np.uint8).copy()


def to_model_input(u8: np.ndarray) -> torch.Tensor:
    """transforms.ToTensor() + Normalize([0.5], [0.5]) (:91-94) of a uint8 HWC canvas -> fp32 [1, 3, H, W] in [-1, 1]"""
    return (torch.from_numpy(u8).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5


def seeded():
    """everything that is NOT the pictures: seeded tensors shared with the GPU test"""
    g = torch.Generator().manual_seed(77)
    return dict(post_noise=torch.randn(1, 4, H // 8, 2 * W // 8, generator=g), latents=torch.randn(N, 4, H // 8, 2 * W // 8, generator=g))


def main():
    from oracle import cond as OC
    from oracle import vae as OV
    from oracle.pipeline import stage2_sample, synth_inputs
    from oracle.schedulers import DDIMOracle
    from oracle.unet import UNetConfig, synth_state_dict
    t0 = time.time()
    canvas_u8, pose_u8 = canvases()
    sdd = seeded()
    cfg = UNetConfig()
    vcfg = OV.VAEConfig()
    vsd = OV.synth_state_dict(vcfg, 0)
    psd = OC.synth(OC.pose_param_shapes(), seed=SEED_POSE_NET)
    out = {"torch_version": np.array(torch.__version__), "canvas_u8": canvas_u8, "pose_u8": pose_u8}
    with torch.no_grad():
        x = to_model_input(canvas_u8)[None]
        moments = OV.encode_moments(vsd, vcfg, x)
        ml = OV.sample_latents(moments, sdd["post_noise"]) * vcfg.scaling_factor          # ref :443-444
        st_pose_f = OC.pose_embedding(psd, to_model_input(pose_u8)[None])                   # ref :173-174
        print(f"masked latents: left half std {ml[..., :32].std():.3f} mean {ml[..., :32].mean():+.3f} | right half (black) std {ml[..., 32:].std():.3f} "
              f"mean {ml[..., 32:].mean():+.3f};  st_pose_f std {st_pose_f.std():.3f}  ({time.time() - t0:.0f} s)", flush=True)
        syn = synth_inputs(cfg, H // 8, 2 * W // 8, N)          # DINOv2 tokens / stage-1 embedding: seeded synthetic (no encoders offline)
        sd = synth_state_dict(cfg, seed=0, random_affine=True)
        eps_l, lat_l = [], []

        def hook(i, t, eps, lat):
            eps_l.append(eps.numpy().astype(np.float16))
            lat_l.append(lat.numpy().copy())
            print(f"step {i} t={t} |eps| {eps.norm():.3f}  ({time.time() - t0:.0f} s)", flush=True)

        lat = stage2_sample(sd, cfg, DDIMOracle(), masked_latents=ml, s_img_proj_f=syn["s_img_proj_f"], st_pose_f=st_pose_f,
                            pred_t_img_embed=syn["pred_t_img_embed"], latents=sdd["latents"], num_images_per_prompt=N, guidance_scale=2.0,
                            num_inference_steps=STEPS, eps_hook=hook)
        img = OV.postprocess_uint8(OV.decode(vsd, vcfg, lat / vcfg.scaling_factor))[0]
    out.update(masked_latents=ml.numpy(), st_pose_f_sub=st_pose_f[:, :, ::4, ::4].numpy().astype(np.float16),
               st_pose_f_norm=np.array(float(st_pose_f.norm())), eps=np.stack(eps_l), lat_before=np.stack(lat_l), lat_final=lat.numpy(), img=img.numpy())
    path = ROOT / "tests" / "golden" / "real_image.npz"
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({path.stat().st_size / 1e3:.0f} kB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
