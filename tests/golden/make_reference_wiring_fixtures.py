#!/usr/bin/env python3
"""Generate tests/golden/ref_wiring_*.npz by running the REFERENCE's own code in this container.

    python tests/golden/make_reference_wiring_fixtures.py        (needs /root/reference; dev container only)

What is executed from /root/reference (imported, never copied):
  * ``Stage2_InapintUNet2DConditionModel.__init__`` / ``.forward``
    (src/models/stage2_inpaint_unet_2d_condition.py:66-448, :579-825)
  * ``Stage2_InpaintDiffusionPipeline.__call__`` (src/pipelines/stage2_inpaint_pipeline.py:389-541)
on top of tests/golden/diffusers_stub.py (block internals = oracle.unet, schedulers = oracle.schedulers,
because diffusers 0.24.0 itself is not available offline).  The fixtures therefore pin the reference's
own wiring; see diffusers_stub.py for exactly what they do and do not pin.

Fixtures hold inputs + expected outputs only (a few KB); weights are regenerated from
``synth_state_dict(cfg, seed)`` and guarded by a checksum stored in the fixture.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")

from oracle.pipeline import synth_inputs  # noqa: E402
from oracle.unet import UNetConfig, synth_state_dict  # noqa: E402

FIX_CFG = dict(block_out_channels=(64, 64, 128, 128), attention_head_dim=(1, 1, 2, 2), cross_attention_dim=64,
               projection_class_embeddings_input_dim=64, sample_size=8)
SEED = 7


def weights_checksum(sd) -> float:
    return float(sum(v.double().abs().sum().item() * (1 + (i % 7)) for i, (k, v) in enumerate(sorted(sd.items()))))


def build_reference_unet(cfg: UNetConfig, sd):
    from tests.golden import diffusers_stub
    diffusers_stub.install()
    sys.path.insert(0, str(REF))
    from src.models.stage2_inpaint_unet_2d_condition import Stage2_InapintUNet2DConditionModel as RefUNet
    unet = RefUNet(sample_size=cfg.sample_size, in_channels=9, block_out_channels=cfg.block_out_channels,
                   attention_head_dim=cfg.attention_head_dim, cross_attention_dim=cfg.cross_attention_dim,
                   use_linear_projection=True, class_embed_type="projection",
                   projection_class_embeddings_input_dim=cfg.projection_class_embeddings_input_dim)
    missing, unexpected = unet.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    # the reference pipeline hard-casts inputs to fp16 (:501); the stub blocks up-cast, the three torch modules
    # the reference constructs itself get an up-casting pre-hook (no reference code is modified)
    for m in (unet.conv_in, unet.conv_norm_out, unet.conv_out):
        m.register_forward_pre_hook(lambda mod, a: (a[0].float(),))
    return unet.eval()


def main():
    cfg = UNetConfig(**FIX_CFG)
    sd = synth_state_dict(cfg, seed=SEED, random_affine=True)
    unet = build_reference_unet(cfg, sd)
    meta = dict(seed=SEED, weights_checksum=weights_checksum(sd), torch_version=torch.__version__,
                **{k: np.array(v) for k, v in FIX_CFG.items()})

    # ---- fixture 1: UNet.forward (fp32 inputs)
    g = torch.Generator().manual_seed(11)
    B, h, w, L = 4, 8, 16, 6
    sample = torch.randn(B, 9, h, w, generator=g)
    ehs = torch.randn(B, L, 64, generator=g)
    ehs[:2] = 0
    cl = torch.randn(B, 1, 64, generator=g) * 0.4
    pose = torch.randn(B, 64, h, w, generator=g) * 0.1
    with torch.no_grad():
        eps = unet(sample, torch.tensor(621), encoder_hidden_states=ehs, class_labels=cl, my_pose_cond=pose,
                   return_dict=False)[0]
        eps_obj = unet(sample, 621, encoder_hidden_states=ehs, class_labels=cl, my_pose_cond=pose).sample
    assert torch.equal(eps, eps_obj)
    if not (HERE / "ref_wiring_unet.npz").exists() or "--force" in sys.argv:
        np.savez_compressed(HERE / "ref_wiring_unet.npz", sample=sample.numpy(), timestep=621, ehs=ehs.numpy(),
                            class_labels=cl.numpy(), pose=pose.numpy(), eps=eps.float().numpy(), **meta)

    # ---- fixture 2/3: the reference pipeline __call__ (DDIM 4 steps / UniPC 4 steps), one pair, N=2, guidance 2.0
    from src.pipelines.stage2_inpaint_pipeline import Stage2_InpaintDiffusionPipeline as RefPipe
    from tests.golden import diffusers_stub
    N, h, w, L = 2, 8, 16, 5
    inp = synth_inputs(cfg, h, w, N, L_img=L)
    # (kind, steps, guidance_rescale, file suffix); the third run exercises ``rescale_noise_cfg`` inside the reference's own loop
    # (ref :510-516; the shipped driver passes 0.0).  Existing files are kept unless --force (np.savez output is not byte-stable).
    for kind, steps, gr, suffix in (("ddim", 4, 0.0, "ddim"), ("unipc", 4, 0.0, "unipc"), ("ddim", 4, 0.7, "ddim_gr07")):
        if (HERE / f"ref_wiring_pipeline_{suffix}.npz").exists() and "--force" not in sys.argv:
            continue
        vae = diffusers_stub.FakeVAE(inp["masked_latents"] / 0.18215)
        pipe = RefPipe(vae=vae, unet=unet, scheduler=diffusers_stub.make_scheduler(kind))
        trace = []
        with torch.no_grad():
            out = pipe(height=h * 8, width=w * 8, vae_image=torch.zeros(1, 3, h * 8, w * 8),
                       s_img_proj_f=inp["s_img_proj_f"], st_pose_f=inp["st_pose_f"],
                       pred_t_img_embed=inp["pred_t_img_embed"], num_images_per_prompt=N, guidance_scale=2.0,
                       generator=None, num_inference_steps=steps, latents=inp["latents"].clone(),
                       guidance_rescale=gr, output_type="pt",
                       callback=lambda i, t, lat: trace.append(lat.float().clone()), callback_steps=1)
        final = trace[-1]
        # decode() is the identity in the stub, so .images == final latents / scaling_factor
        assert torch.allclose(out.images.float() * 0.18215, final, atol=1e-6)
        np.savez_compressed(HERE / f"ref_wiring_pipeline_{suffix}.npz", steps=steps, N=N, guidance_rescale=gr,
                            final_latents=final.numpy(), trace=torch.stack(trace).numpy(),
                            **{k: v.numpy() for k, v in inp.items()}, **meta)
        print(kind, "final latents std", float(final.std()))
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
