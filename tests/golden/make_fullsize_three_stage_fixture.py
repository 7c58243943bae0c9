#!/usr/bin/env python3
"""The fp32 CPU oracle chain of BASELINE.json configs[3] at FULL model sizes (tests/three_stage_common.py: seeded weights, one pair,
N = 2 per stage, 20 / 10 / 5 steps), for ``tests/test_three_stage_flow.py::test_three_stage_full_size`` (the
hand-overs stage-1 embedding -> stage-2 class label and stage-2 pixels -> stage-3 VAE encode were only ever checked on tiny models).

Encoders: ``transformers``' own CLIPVisionModelWithProjection / Dinov2Model (the classes the reference's drivers call) with the
seeded weights loaded; everything else: oracle/.  Stored: every hand-over tensor (stage-1 embedding, image-projection tokens
(fp16), pose feature statistics, masked latents, stage-2 latents, decoded target half (uint8), stage-3 conditioning latents, stage-3
latents, final uint8 images).  About 20 min on the 8 build-container cores.

    python tests/golden/make_fullsize_three_stage_fixture.py
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def main():
    from transformers import CLIPVisionConfig, Dinov2Config
    from transformers import CLIPVisionModelWithProjection as HFCLIP
    from transformers import Dinov2Model as HFDino
    from oracle import cond as OC
    from oracle import prior as OP
    from oracle import vae as OV
    from oracle.pipeline import stage2_sample, stage3_sample
    from oracle.schedulers import DDIMOracle, UnCLIPOracle
    from pcdms_amd.encoders import CLIP_VIT_H14_CONFIG, DINOV2_GIANT_CONFIG
    from tests import three_stage_common as T
    t0 = time.time()
    Wt, I = T.weights(), T.inputs()
    out = {"torch_version": np.array(torch.__version__)}

    def log(msg):
        print(f"[{time.time() - t0:6.0f} s] {msg}", flush=True)
    with torch.no_grad():
        hf_clip = HFCLIP(CLIPVisionConfig(**CLIP_VIT_H14_CONFIG)).eval()
        missing, unexpected = hf_clip.load_state_dict(Wt["clip"], strict=False)
        assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
        o_embed = hf_clip(I["pix224"]).image_embeds.unsqueeze(1)
        del hf_clip
        log(f"CLIP-H embed |.| {o_embed.norm():.3f}")
        hf_dino = HFDino(Dinov2Config(**{k: v for k, v in DINOV2_GIANT_CONFIG.items()})).eval()
        missing, unexpected = hf_dino.load_state_dict(Wt["dino"], strict=False)
        assert not unexpected and not missing, (missing, unexpected)
        o_feat = OC.image_proj_p(Wt["iproj"], hf_dino(I["pix224"]).last_hidden_state)
        del hf_dino
        log(f"DINOv2-g -> image proj {tuple(o_feat.shape)} |.| {o_feat.norm():.3f}")
        o_pred = OP.stage1_sample(Wt["prior"], Wt["pcfg"], UnCLIPOracle(), s_embed=o_embed, s_pose=I["s_kp"], t_pose=I["t_kp"], latents=I["s1_lat"],
                                  noises=I["s1_noise"], num_inference_steps=T.S1_STEPS, guidance_scale=0).unsqueeze(1)
        log(f"stage 1 done |pred| {o_pred.norm():.3f}")
        o_pose = OC.pose_embedding(Wt["pose"], I["pose"])
        vsd, vcfg = Wt["vae"], Wt["vcfg"]
        o_ml = OV.sample_latents(OV.encode_moments(vsd, vcfg, I["canvas"]), I["post_noise"]) * vcfg.scaling_factor
        log("pose embedding + VAE encode done")
        o_lat2 = stage2_sample(Wt["unet2"], Wt["ucfg"], DDIMOracle(), masked_latents=o_ml, s_img_proj_f=o_feat, st_pose_f=o_pose,
                               pred_t_img_embed=o_pred, latents=I["s2_lat"], num_images_per_prompt=T.N2, guidance_scale=2.0,
                               num_inference_steps=T.S2_STEPS, eps_hook=lambda i, t, e, l: log(f"  stage 2 step {i}"))
        o_img2 = torch.cat([OV.decode(vsd, vcfg, o_lat2[k:k + 1] / vcfg.scaling_factor) for k in range(T.N2)])
        log("stage 2 + decode done")
        gen_t = o_img2[:1, :, :, T.W:].clamp(-1, 1).contiguous()            # target half of sample 0 (the driver picks the best-SSIM one)
        o_gl = OV.sample_latents(OV.encode_moments(vsd, vcfg, gen_t), I["post_noise3"]) * vcfg.scaling_factor
        o_lat3 = stage3_sample(Wt["unet3"], Wt["u3cfg"], DDIMOracle(), gen_t_img_latents=o_gl, s_img_proj_f=o_feat, latents=I["s3_lat"],
                               num_images_per_prompt=T.N3, guidance_scale=2.0, num_inference_steps=T.S3_STEPS)
        o_u8 = OV.postprocess_uint8(torch.cat([OV.decode(vsd, vcfg, o_lat3[k:k + 1] / vcfg.scaling_factor) for k in range(T.N3)]))
        log("stage 3 + decode done")
    out.update(embed=o_embed.numpy(), pred=o_pred.numpy(), feat=o_feat.numpy().astype(np.float16),
               pose_mean=o_pose.mean((0, 2, 3)).numpy(), pose_sub=o_pose[:, :, ::8, ::8].numpy().astype(np.float16),
               ml=o_ml.numpy(), lat2=o_lat2.numpy(), img2_u8=OV.postprocess_uint8(o_img2)[:, ::2, ::2].numpy(),
               gen_t_u8=OV.postprocess_uint8(gen_t).numpy(), gl=o_gl.numpy(), lat3=o_lat3.numpy(), u8=o_u8.numpy())
    path = ROOT / "tests" / "golden" / "fullsize_three_stage.npz"
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({path.stat().st_size / 1e6:.2f} MB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
