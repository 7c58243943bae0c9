"""CPU restatement of the stage-2 sampling loop (oracle; test infrastructure).

Follows /root/reference/src/pipelines/stage2_inpaint_pipeline.py:420-525 for ONE (source,target)
pair with ``num_images_per_prompt = N`` samples.  VAE encode/decode are
outside the hot path, so ``masked_latents`` (= vae.encode(...)*scaling_factor,
ref :443-444) and ``latents`` (ref :478-487) are inputs.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from .unet import UNetConfig, unet_forward


def left_half_mask(bs: int, h: int, w: int) -> torch.Tensor:
    """ref :434-437 -- ones over the source (left) half of the canvas, zeros over the target."""
    return torch.cat([torch.ones(bs, 1, h, w // 2), torch.zeros(bs, 1, h, w // 2)], dim=3)


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    """ref :52-63."""
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg


def build_conditioning(masked_latents, s_img_proj_f, st_pose_f, pred_t_img_embed, N: int, cfg_on: bool,
                       mask=None, use_prior_embed: bool = True, uncond_feature=None):
    """ref :430-466: CFG-doubled conditioning for one pair.  Returns dict of fp32 tensors."""
    bs = s_img_proj_f.shape[0]
    assert bs == 1, "reference semantics hold for one pair per call (SURVEY.md Appendix C-1)"
    h, w = masked_latents.shape[-2:]
    rep = 2 * N if cfg_on else N
    pose_cond = torch.cat([st_pose_f] * rep)
    if mask is None:
        mask = left_half_mask(bs, h, w)
    mask = torch.cat([mask] * rep)
    ml = torch.cat([masked_latents] * rep)
    if use_prior_embed:
        feature_f = torch.cat([s_img_proj_f, pred_t_img_embed], dim=1).repeat(bs * N, 1, 1)
        prior_embed = pred_t_img_embed.repeat(bs * N, 1, 1)
    else:   # Simple_Stage2_InpaintDiffusionPipeline (ref :812): source-image tokens only, no class_labels
        feature_f = s_img_proj_f.repeat(bs * N, 1, 1)
        prior_embed = None
    if cfg_on:
        # ref :455-458: literal zeros; the notebook caller (src/pipelines/PCDMs_pipeline.py:1061-1062, pcdms_kaggle_demo.ipynb cell 37)
        # passes negative_prompt_embeds = image_proj_model(zeros) instead: ``uncond_feature`` [1, L, D]
        unc = torch.zeros_like(feature_f) if uncond_feature is None else uncond_feature.repeat(bs * N, 1, 1)
        feature_f = torch.cat([unc, feature_f], dim=0)
        if prior_embed is not None:
            prior_embed = torch.cat([torch.zeros_like(prior_embed), prior_embed], dim=0)
    # NOTE ref :464-466 repeats feature_f a second time when CFG is off (only valid for N=1);
    # the evident intent (N rows) is implemented.
    return dict(pose_cond=pose_cond, mask=mask, masked_latents=ml, feature_f=feature_f, prior_embed=prior_embed)


def stage2_sample(sd: Dict[str, torch.Tensor], cfg: UNetConfig, scheduler, *, masked_latents, s_img_proj_f,
                  st_pose_f, pred_t_img_embed, latents, num_images_per_prompt: int = 4,
                  guidance_scale: float = 2.0, num_inference_steps: int = 50, guidance_rescale: float = 0.0,
                  mask=None, eps_hook: Optional[Callable] = None,
                  unet: Optional[Callable] = None, use_prior_embed: bool = True, uncond_feature=None) -> torch.Tensor:
    """Returns final latents [N,4,h,w] (before vae.decode, ref :528).

    ``unet(sample, t, encoder_hidden_states=, class_labels=, my_pose_cond=) -> eps`` may replace
    the oracle UNet (used to run the oracle loop on top of another UNet implementation)."""
    N = num_images_per_prompt
    cfg_on = guidance_scale > 1.0
    c = build_conditioning(masked_latents, s_img_proj_f, st_pose_f, pred_t_img_embed, N, cfg_on, mask, use_prior_embed, uncond_feature)
    scheduler.set_timesteps(num_inference_steps)
    latents = latents * scheduler.init_noise_sigma
    if unet is None:
        def unet(x, t, encoder_hidden_states, class_labels, my_pose_cond):
            return unet_forward(sd, cfg, x, t, encoder_hidden_states, class_labels, my_pose_cond)
    for i, t in enumerate(scheduler.timesteps):
        x = torch.cat([latents] * 2) if cfg_on else latents
        x = scheduler.scale_model_input(x, t)
        inp = torch.cat([x, c["mask"], c["masked_latents"]], dim=1)
        eps = unet(inp, t, encoder_hidden_states=c["feature_f"], class_labels=c["prior_embed"],
                   my_pose_cond=c["pose_cond"])
        if cfg_on:
            u, cnd = eps.chunk(2)
            eps = u + guidance_scale * (cnd - u)
            if guidance_rescale > 0.0:
                eps = rescale_noise_cfg(eps, cnd, guidance_rescale)
        if eps_hook is not None:
            eps_hook(i, int(t), eps, latents)
        latents = scheduler.step(eps, t, latents)
    return latents


def synth_inputs(cfg: UNetConfig, h: int, w: int, N: int, L_img: int = 257):
    """Seeded synthetic conditioning of SURVEY.md §8(d) (CPU generators; seeds 1..5)."""
    def g(seed):
        return torch.Generator(device="cpu").manual_seed(seed)
    ctx = cfg.cross_attention_dim
    latents = torch.randn(N, 4, h, w, generator=g(1))
    ml = torch.zeros(1, 4, h, w)
    ml[..., : w // 2] = torch.randn(1, 4, h, w // 2, generator=g(2)) * 0.18215 * 5
    st_pose_f = torch.randn(1, cfg.block_out_channels[0], h, w, generator=g(3)) * 0.1
    s_img_proj_f = torch.randn(1, L_img, ctx, generator=g(4))
    pred = torch.randn(1, 1, cfg.projection_class_embeddings_input_dim or ctx, generator=g(5)) * 0.4
    return dict(latents=latents, masked_latents=ml, st_pose_f=st_pose_f, s_img_proj_f=s_img_proj_f,
                pred_t_img_embed=pred)


def stage3_sample(sd, cfg: UNetConfig, scheduler, *, gen_t_img_latents, s_img_proj_f, latents, num_images_per_prompt: int = 1,
                  guidance_scale: float = 2.0, num_inference_steps: int = 20) -> torch.Tensor:
    """Stage-3 refinement loop, /root/reference/src/pipelines/stage3_refined_pipeline.py:483-557, with the batch bug of
    the reference FIXED (it does not repeat the conditioning for num_images_per_prompt in the
    CFG branch): conditioning is repeated to N rows; the uncond half has zero context AND zero refine latents (:491-497).
    UNet input = cat([latents, gen_t_img_latents], 1) (8 channels, :538); stock UNet (no class_labels / pose)."""
    N = num_images_per_prompt
    cfg_on = guidance_scale > 1.0
    feat = s_img_proj_f.repeat(N, 1, 1)
    gl = gen_t_img_latents.repeat(N, 1, 1, 1)
    if cfg_on:
        feat = torch.cat([torch.zeros_like(feat), feat])
        gl = torch.cat([torch.zeros_like(gl), gl])
    scheduler.set_timesteps(num_inference_steps)
    latents = latents * scheduler.init_noise_sigma
    for t in scheduler.timesteps:
        x = torch.cat([latents] * 2) if cfg_on else latents
        eps = unet_forward(sd, cfg, torch.cat([x, gl], 1), t, feat, None, None)
        if cfg_on:
            u, c = eps.chunk(2)
            eps = u + guidance_scale * (c - u)
        latents = scheduler.step(eps, t, latents)
    return latents
