#!/usr/bin/env python3
"""BASELINE.json configs[3]: stage-1 prior + stage-2 + stage-3 refine, 352x512, 8 samples per pair, one MI355X -- every stage at its
full size with seeded random weights (no checkpoints offline), from pixels to pixels, timed per stage with HIP events.

Per pair, following the three drivers: CLIP ViT-H/14 embed of the source -> stage-1 prior (N = 1, 20 UnCLIP steps, guidance 0;
stage1_batchtest_prior_model.py:105-113) -> DINOv2-giant + ImageProjModel_p, pose canvas -> ControlNetConditioningEmbedding, VAE encode of
the [source | black] canvas -> stage-2 (N = 8 -> UNet batch 16, 50 DDIM steps, guidance 2; stage2_batchtest_inpaint_model.py:188-200) -> VAE
decode -> stage-3 on the target half of one sample (N = 4, 20 steps, guidance 2; stage3_batchtest_refined_model.py:161-171) -> uint8.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pcdms_amd as P  # noqa: E402
from oracle import cond as OC  # noqa: E402  (synthetic weights only)
from oracle import prior as OP  # noqa: E402
from oracle import vae as OV  # noqa: E402
from oracle.unet import UNetConfig, synth_state_dict  # noqa: E402
from tests.test_schedulers import SD21  # noqa: E402
from tests.test_unet import _kwargs  # noqa: E402


def _rand_sd(model, g):
    sd = {}
    for k, shp in model.expected_shapes().items():
        if len(shp) >= 2 and "position" not in k and "cls_token" not in k and "mask_token" not in k:
            fan = shp[1] * (shp[2] * shp[3] if len(shp) == 4 else 1)
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) / fan ** 0.5
        elif k.endswith("lambda1") or (k.endswith(".weight") and len(shp) == 1):
            sd[k] = torch.ones(shp)
        else:
            sd[k] = torch.randn(shp, generator=g) * 0.02
    return sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n2", type=int, default=8, help="stage-2 samples per pair")
    ap.add_argument("--iters", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    t0 = time.time()
    clip = P.CLIPVisionModelWithProjection(); clip.load_state_dict(_rand_sd(clip, g)); clip.to(dev)
    dino = P.Dinov2Model(); dino.load_state_dict(_rand_sd(dino, g)); dino.to(dev)
    prior = P.Stage1_PriorTransformer(num_embeddings=2, embedding_dim=1024); prior.load_state_dict(OP.synth_state_dict(OP.PriorConfig(), 1)); prior.to(dev)
    iproj = P.ImageProjModel_p(1536, 768, 1024); iproj.load_state_dict(OC.synth(OC.image_proj_param_shapes(), 2, 1.0)); iproj.to(dev)
    pose_proj = P.ControlNetConditioningEmbedding(320, 3, (16, 32, 96, 256)); pose_proj.load_state_dict(OC.synth(OC.pose_param_shapes(), 3)); pose_proj.to(dev)
    vae = P.AutoencoderKL(); vae.load_state_dict(OV.synth_state_dict(OV.VAEConfig(), 4)); vae.to(dev)
    ucfg = UNetConfig()
    unet2 = P.Stage2_InapintUNet2DConditionModel(**_kwargs(ucfg)); unet2.load_state_dict(synth_state_dict(ucfg, seed=5)); unet2.to(dev)
    u3cfg = UNetConfig(in_channels=8, class_embed_type=None, projection_class_embeddings_input_dim=None)
    unet3 = P.UNet2DConditionModel(**_kwargs(u3cfg)); unet3.load_state_dict(synth_state_dict(u3cfg, seed=6)); unet3.to(dev)
    pipe1 = P.Stage1_PriorPipeline(prior).to(dev)
    pipe2 = P.Stage2_InpaintDiffusionPipeline(unet2, P.DDIMScheduler.from_config(SD21), vae=vae)
    pipe3 = P.Stage3_RefinedDiffusionPipeline(unet3, P.DDIMScheduler.from_config(SD21), vae=vae)
    load_s = time.time() - t0
    H, W = 512, 352
    s_img = (torch.rand(1, 3, H, W, generator=g) * 2 - 1).to(dev)
    pose = (torch.rand(1, 3, H, 2 * W, generator=g) * 2 - 1).to(dev)
    pix = torch.randn(1, 3, 224, 224, generator=g).to(dev)
    kp_s, kp_t = torch.rand(1, 1, 36, generator=g).to(dev), torch.rand(1, 1, 36, generator=g).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def one_pair(timed):
        marks = [ev() for _ in range(6)]
        marks[0].record()
        s_embed = clip(pix).image_embeds.unsqueeze(1)
        pred = pipe1(s_embed=s_embed, s_pose=kp_s, t_pose=kp_t, num_images_per_prompt=1, num_inference_steps=20, generator=gen,
                     guidance_scale=0)[0].unsqueeze(1)
        marks[1].record()
        feat = iproj(dino(pix).last_hidden_state)
        st_pose_f = pose_proj(pose)
        canvas = torch.cat([s_img, -torch.ones_like(s_img)], dim=3)
        marks[2].record()
        out2 = pipe2(height=H, width=2 * W, vae_image=canvas, s_img_proj_f=feat, st_pose_f=st_pose_f, pred_t_img_embed=pred,
                     num_images_per_prompt=a.n2, guidance_scale=2.0, generator=gen, num_inference_steps=50, output_type="pt")
        marks[3].record()
        gen_t = (out2.images[:1, :, :, W:] * 2 - 1).contiguous()          # target half of one sample (the driver picks the best-SSIM one)
        marks[4].record()
        out3 = pipe3(height=H, width=W, vae_gen_t_image=gen_t, s_img_proj_f=feat, num_images_per_prompt=4, guidance_scale=2.0, generator=gen,
                     num_inference_steps=20, output_type="uint8")
        marks[5].record()
        torch.cuda.synchronize()
        assert out3.images.shape == (4, H, W, 3) and out2.images.shape == (a.n2, 3, H, 2 * W) and torch.isfinite(out2.latents).all()
        names = ["stage1 (CLIP-H + prior, 20 steps)", "conditioning (DINOv2-g, image proj, pose embed)", f"stage2 (VAE enc, N={a.n2}, 50 DDIM, VAE dec)",
                 "glue", "stage3 (VAE enc, N=4, 20 steps, VAE dec + uint8)"]
        return {n: marks[i].elapsed_time(marks[i + 1]) for i, n in enumerate(names)}
    one_pair(False)   # packs weights, autotunes new shapes, captures the stage-2 graph
    one_pair(False)
    acc = {}
    for _ in range(a.iters):
        for k, v in one_pair(True).items():
            acc[k] = acc.get(k, 0.0) + v / a.iters
    total = sum(acc.values())
    print(json.dumps(dict(metric="three-stage pair latency (BASELINE.json configs[3])", total_ms=round(total, 1),
                          stage2_images_per_s=round(a.n2 / (total * 1e-3), 3), per_stage_ms={k: round(v, 1) for k, v in acc.items()},
                          load_and_pack_s=round(load_s, 1), data="synthetic", weights="seeded random, full-size")))


if __name__ == "__main__":
    main()
