"""Seeded weights and inputs of the FULL-SIZE three-stage chain (BASELINE.json configs[3]), shared by the fixture generator
(tests/golden/make_fullsize_three_stage_fixture.py: the CPU oracle chain) and the ``-m gpu`` test
(tests/test_three_stage_flow.py::test_three_stage_full_size: the HIP chain).  Test infrastructure.

Chain (the reference's three drivers: stage1_batchtest_prior_model.py:80-113, stage2_batchtest_inpaint_model.py:150-200,
stage3_batchtest_refined_model.py:140-171), every model at the reference's size with seeded random weights:
CLIP ViT-H/14 -> stage-1 prior (20 UnCLIP steps) -> [DINOv2-giant -> ImageProjModel_p, pose embedding, VAE encode of the
[source | black] canvas] -> stage 2 (N = 2, ``S2_STEPS`` DDIM steps, guidance 2) -> VAE decode -> target half of sample 0 -> VAE encode
-> stage 3 (N = 2, ``S3_STEPS`` steps) -> VAE decode -> uint8.  The step counts are cut (10 / 5 instead of 50 / 20) so that the fp32
oracle chain takes minutes, not an hour, on the 8 build-container cores: the hand-overs, shapes and models are the full ones.
"""
from __future__ import annotations

import torch

H, W = 512, 352             # one person image; canvas 512 x 704 -> latent 64 x 88; stage 3 on 512 x 352 -> latent 64 x 44
N2, N3 = 2, 2
S1_STEPS, S2_STEPS, S3_STEPS = 20, 10, 5


def rand_sd(shapes: dict, seed: int) -> dict:
    """weights for an encoder given its ``expected_shapes()`` (as tools/bench_three_stage.py)"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in shapes.items():
        if len(shp) >= 2 and "position" not in k and "cls_token" not in k and "mask_token" not in k:
            fan = shp[1] * (shp[2] * shp[3] if len(shp) == 4 else 1)
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) / fan ** 0.5
        elif k.endswith("lambda1") or (k.endswith(".weight") and len(shp) == 1):
            sd[k] = torch.ones(shp)
        else:
            sd[k] = torch.randn(shp, generator=g) * 0.02
    return sd


def inputs():
    g = torch.Generator().manual_seed(0)
    d = dict(
        s_img=torch.rand(1, 3, H, W, generator=g) * 2 - 1,
        pose=torch.rand(1, 3, H, 2 * W, generator=g) * 2 - 1,
        pix224=torch.randn(1, 3, 224, 224, generator=g),
        s_kp=torch.rand(1, 1, 36, generator=g), t_kp=torch.rand(1, 1, 36, generator=g),
        s1_lat=torch.randn(1, 1024, generator=g),
        s1_noise=[torch.randn(1, 1024, generator=g) for _ in range(S1_STEPS)],
        post_noise=torch.randn(1, 4, H // 8, 2 * W // 8, generator=g),
        s2_lat=torch.randn(N2, 4, H // 8, 2 * W // 8, generator=g),
        post_noise3=torch.randn(1, 4, H // 8, W // 8, generator=g),
        s3_lat=torch.randn(N3, 4, H // 8, W // 8, generator=g),
    )
    d["canvas"] = torch.cat([d["s_img"], -torch.ones_like(d["s_img"])], dim=3)      # [source | black]
    return d


def weights():
    """state dicts of the seven models (CPU fp32)"""
    import pcdms_amd as P
    from oracle import cond as OC
    from oracle import prior as OP
    from oracle import vae as OV
    from oracle.unet import UNetConfig, synth_state_dict
    ucfg = UNetConfig()
    u3cfg = UNetConfig(in_channels=8, class_embed_type=None, projection_class_embeddings_input_dim=None)
    return dict(
        clip=rand_sd(P.CLIPVisionModelWithProjection().expected_shapes(), 11),
        dino=rand_sd(P.Dinov2Model().expected_shapes(), 12),
        prior=OP.synth_state_dict(OP.PriorConfig(), 1), pcfg=OP.PriorConfig(),
        iproj=OC.synth(OC.image_proj_param_shapes(), 2, 1.0),
        pose=OC.synth(OC.pose_param_shapes(), 3),
        vae=OV.synth_state_dict(OV.VAEConfig(), 4), vcfg=OV.VAEConfig(),
        unet2=synth_state_dict(ucfg, seed=5, random_affine=True), ucfg=ucfg,
        unet3=synth_state_dict(u3cfg, seed=6, random_affine=True), u3cfg=u3cfg,
    )
