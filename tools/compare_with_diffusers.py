#!/usr/bin/env python3
"""Close the "parity unpinned" gap when the real third-party code is available (SURVEY.md §8c f).

The diffusers 0.24.0 blocks the reference composes cannot be imported in the offline build container, so the in-repo
oracle restates them (oracle/__init__.py).  On a machine that has `diffusers==0.24.0` (and a checkout of
tencent-ailab/PCDMs) this script compares pcdms_amd on the MI355X directly against the real implementation:

    python tools/compare_with_diffusers.py --reference-root /path/to/PCDMs [--sd21 /path/to/stable-diffusion-2-1-base]
                                           [--tiny] [--device cuda]

  1. UNet: the reference's own ``Stage2_InapintUNet2DConditionModel`` (fp32, random init from the SD-2.1 config or
     ``--sd21`` weights) vs ``pcdms_amd.Stage2_InapintUNet2DConditionModel`` loaded from its state_dict.
  2. Schedulers: diffusers DDIM / UniPC / DDPM / UnCLIP ``step`` vs pcdms_amd's on random tensors, every timestep.
  3. VAE: diffusers ``AutoencoderKL`` encode moments / decode vs ``pcdms_amd.AutoencoderKL``.
  4. Stage-1 prior: the reference's ``Stage1_PriorTransformer`` vs ``pcdms_amd.Stage1_PriorTransformer``.

STATUS: written against the diffusers 0.24.0 API from memory; NOT executed in the build container (diffusers is not
installable offline).  Exit code 0 iff every comparison is within the tolerance stated in the tests (rel-L2 3e-2 for
the bf16 networks, 2e-5 for the schedulers).
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

SD21_UNET = dict(sample_size=64, in_channels=9, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True,
                 class_embed_type="projection", projection_class_embeddings_input_dim=1024)
TINY_UNET = dict(SD21_UNET, block_out_channels=(64, 64, 128, 128), attention_head_dim=(1, 1, 2, 2), sample_size=8)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference-root", required=True, help="checkout of tencent-ailab/PCDMs")
    ap.add_argument("--sd21", default=None, help="stable-diffusion-2-1-base directory (optional: real weights)")
    ap.add_argument("--tiny", action="store_true", help="small channel counts (fast CPU reference)")
    ap.add_argument("--device", default="cuda")
    a = ap.parse_args()
    import diffusers
    print("diffusers", diffusers.__version__, "(the reference pins 0.24.0)")
    sys.path.insert(0, a.reference_root)
    import pcdms_amd as P
    dev = torch.device(a.device)
    g = torch.Generator().manual_seed(0)
    report = {}

    # ---- 1. stage-2 UNet ------------------------------------------------------------------------------------------
    from src.models.stage2_inpaint_unet_2d_condition import Stage2_InapintUNet2DConditionModel as RefUNet
    cfg = dict(TINY_UNET if a.tiny else SD21_UNET)
    if a.sd21 and not a.tiny:
        ref_unet = RefUNet.from_pretrained(a.sd21, subfolder="unet", in_channels=9, class_embed_type="projection",
                                           projection_class_embeddings_input_dim=1024, low_cpu_mem_usage=False,
                                           ignore_mismatched_sizes=True)
    else:
        ref_unet = RefUNet(**cfg)
    ref_unet = ref_unet.float().eval()
    mine = P.Stage2_InapintUNet2DConditionModel(**cfg)
    mine.load_state_dict(ref_unet.state_dict())
    mine.to(dev)
    B, h, w, L = 2, (16 if a.tiny else 32), (16 if a.tiny else 64), 258
    x = torch.randn(B, 9, h, w, generator=g)
    ehs = torch.randn(B, L, 1024, generator=g)
    ehs[: B // 2] = 0
    cl = torch.randn(B, 1, 1024, generator=g) * 0.4
    pose = torch.randn(1, cfg["block_out_channels"][0], h, w, generator=g) * 0.1
    with torch.no_grad():
        want = ref_unet(x, torch.tensor(500), encoder_hidden_states=ehs, class_labels=cl, my_pose_cond=pose.repeat(B, 1, 1, 1),
                        return_dict=False)[0]
    got = mine(x.to(dev), torch.tensor(500), encoder_hidden_states=ehs.to(dev), class_labels=cl.to(dev), my_pose_cond=pose.to(dev),
               return_dict=False)[0]
    report["unet_forward"] = (rel(got, want), 3e-2)

    # ---- 2. schedulers --------------------------------------------------------------------------------------------
    sd21_sched = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    pairs = [("ddim", diffusers.DDIMScheduler(**sd21_sched, clip_sample=False, set_alpha_to_one=False, steps_offset=1),
              P.DDIMScheduler(**sd21_sched, clip_sample=False, set_alpha_to_one=False, steps_offset=1), 50),
             ("unipc", diffusers.UniPCMultistepScheduler(**sd21_sched), P.UniPCMultistepScheduler(**sd21_sched), 20),
             ("unclip", diffusers.UnCLIPScheduler(**P.UnCLIPScheduler.KANDINSKY22_PRIOR), P.UnCLIPScheduler(**P.UnCLIPScheduler.KANDINSKY22_PRIOR), 20)]
    for name, theirs, ours, n in pairs:
        theirs.set_timesteps(n)
        ours.set_timesteps(n)
        assert [int(t) for t in theirs.timesteps] == [int(t) for t in ours.timesteps], name
        xs_t = xs_o = torch.randn(2, 4, 8, 8, generator=g)
        worst = 0.0
        ts = [int(t) for t in theirs.timesteps]
        for i, t in enumerate(ts):
            eps = torch.randn(2, 4, 8, 8, generator=g)
            noise = torch.randn(2, 4, 8, 8, generator=g)
            if name == "unclip":
                prev = None if i + 1 == len(ts) else ts[i + 1]
                torch.manual_seed(1234 + i)   # diffusers draws the variance noise itself: reproduce it
                noise = torch.randn(eps.shape)
                torch.manual_seed(1234 + i)
                nt = theirs.step(eps, t, xs_t, prev_timestep=prev).prev_sample
                no = ours.step(eps.to(dev), t, xs_o.to(dev), prev_timestep=prev, variance_noise=noise.to(dev)).prev_sample
            else:
                nt = theirs.step(eps, t, xs_t).prev_sample
                no = ours.step(eps.to(dev), t, xs_o.to(dev)).prev_sample
            worst = max(worst, rel(no, nt))
            xs_t, xs_o = nt, nt   # re-synchronise every step: per-step comparison
        report[f"scheduler_{name}"] = (worst, 2e-5)

    # ---- 3. VAE ---------------------------------------------------------------------------------------------------
    vcfg = dict(block_out_channels=(64, 64, 128, 128) if a.tiny else (128, 256, 512, 512), layers_per_block=2, latent_channels=4,
                in_channels=3, out_channels=3, norm_num_groups=32, down_block_types=("DownEncoderBlock2D",) * 4,
                up_block_types=("UpDecoderBlock2D",) * 4)
    ref_vae = (diffusers.AutoencoderKL.from_pretrained(a.sd21, subfolder="vae") if a.sd21 and not a.tiny else diffusers.AutoencoderKL(**vcfg)).float().eval()
    my_vae = P.AutoencoderKL(block_out_channels=vcfg["block_out_channels"])
    my_vae.load_state_dict(ref_vae.state_dict())
    my_vae.to(dev)
    img = torch.rand(1, 3, 128, 192, generator=g) * 2 - 1
    z = torch.randn(1, 4, 16, 24, generator=g)
    with torch.no_grad():
        want_m = ref_vae.encode(img).latent_dist.parameters
        want_d = ref_vae.decode(z, return_dict=False)[0]
    report["vae_encode_moments"] = (rel(my_vae.encode(img.to(dev)).latent_dist.parameters, want_m), 3e-2)
    report["vae_decode"] = (rel(my_vae.decode(z.to(dev), return_dict=False)[0], want_d), 3e-2)

    # ---- 4. stage-1 prior -----------------------------------------------------------------------------------------
    from src.models.stage1_prior_transformer import Stage1_PriorTransformer as RefPrior
    pk = dict(num_attention_heads=2 if a.tiny else 32, attention_head_dim=64, num_layers=2 if a.tiny else 20, embedding_dim=1024,
              num_embeddings=2, additional_embeddings=4)
    ref_prior = RefPrior(**pk).float().eval()
    with torch.no_grad():
        ref_prior.positional_embedding.normal_(0, 0.3, generator=g)
        ref_prior.prd_embedding.normal_(0, 0.3, generator=g)
    my_prior = P.Stage1_PriorTransformer(**pk)
    my_prior.load_state_dict(ref_prior.state_dict())
    my_prior.to(dev)
    xt, emb = torch.randn(2, 1, 1024, generator=g), torch.randn(2, 1, 1024, generator=g) * 0.4
    sp, tp = torch.rand(2, 1, 36, generator=g), torch.rand(2, 1, 36, generator=g)
    with torch.no_grad():
        want_p = ref_prior(xt, timestep=473, proj_embedding=emb, encoder_hidden_states=sp, encoder_hidden_states1=tp).predicted_image_embedding
    got_p = my_prior(xt.to(dev), 473, emb.to(dev), sp.to(dev), tp.to(dev)).predicted_image_embedding
    report["prior_forward"] = (rel(got_p, want_p), 3e-2)

    bad = 0
    for k, (v, tol) in report.items():
        ok = v <= tol
        bad += not ok
        print(f"{k:24s} rel-L2 {v:.3e}  (tolerance {tol:.0e})  {'ok' if ok else 'FAIL'}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
