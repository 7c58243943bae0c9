"""AutoencoderKL: pcdms_amd.vae (HIP) vs the fp32 CPU oracle (oracle/vae.py).

Stated tolerance: bf16 activations through ~30 conv / 30 GroupNorm layers vs fp32: rel-L2 <= 3e-2 on the latent
moments and on the decoded image; uint8 pixels: mean abs diff <= 1.5 levels, <= 1% of pixels off by more than 8.
"""
from __future__ import annotations

import math

import pytest
import torch

from oracle import vae as O
from pcdms_amd.vae import AutoencoderKL


def _rel(a, b):
    a, b = a.float().cpu(), b.float()
    return ((a - b).norm() / b.norm()).item()


def _build(backend, cfg, seed=0):
    sd = O.synth_state_dict(cfg, seed)
    m = AutoencoderKL(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
                      latent_channels=cfg.latent_channels, norm_num_groups=cfg.norm_num_groups)
    m.load_state_dict(sd)
    return sd, m.to(backend.device)


def test_param_contract():
    cfg = O.VAEConfig()
    m = AutoencoderKL()
    exp = m.expected_shapes()
    assert exp == {k: tuple(v) for k, v in O.param_shapes(cfg)}
    assert sum(math.prod(s) for s in exp.values()) == 83_653_863   # the SD AutoencoderKL parameter count
    assert m.config.scaling_factor == 0.18215 and len(m.config.block_out_channels) == 4
    with pytest.raises(RuntimeError):
        m.load_state_dict({"quant_conv.weight": torch.zeros(8, 8, 1, 1)})


def test_vae_softmax_rows_and_helpers(backend):
    from pcdms_amd import _lib, ops
    dev = backend.device
    g = torch.Generator().manual_seed(0)
    rows, cols = (5, 300) if backend.is_emu else (5632, 5632)
    s = torch.randn(rows, cols, generator=g) * 6
    out = ops.softmax_rows(s.to(dev), torch.empty(rows, cols, dtype=torch.bfloat16, device=dev), 0.3)
    backend.sync()
    ref = torch.softmax(s * 0.3, -1)
    assert (out.float().cpu() - ref).abs().max() <= 4e-3 * ref.max()
    # gaussian sample + uint8 postprocess
    B, zc, h, w = 2, 4, 4, 6
    mom = torch.randn(B, 2 * zc, h, w, generator=g)
    noise = torch.randn(B, zc, h, w, generator=g)
    from pcdms_amd.vae import DiagonalGaussianDistribution
    z = DiagonalGaussianDistribution(mom.to(dev)).sample(noise=noise.to(dev))
    backend.sync()
    assert torch.allclose(z.cpu(), O.sample_latents(mom, noise), atol=1e-5, rtol=1e-5)
    img = torch.randn(B, 4, 8, 8, generator=g).to(dev)
    u8 = torch.empty(B, 8, 8, 3, dtype=torch.uint8, device=dev)
    ops._chk(_lib.lib().pcdm_image_to_uint8(img.data_ptr(), u8.data_ptr(), B, 4, 64, ops._stream(img)), "u8")
    backend.sync()
    assert (u8.cpu().int() - O.postprocess_uint8(img.cpu()[:, :3]).int()).abs().max() <= 1


def test_vae_tiny_encode_decode(backend):
    cfg = O.VAEConfig.tiny()
    H, W = (64, 64) if backend.is_emu else (128, 192)
    B = 1 if backend.is_emu else 2
    sd, m = _build(backend, cfg)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    mom = m.encode(x.to(backend.device)).latent_dist.parameters
    backend.sync()
    ref_mom = O.encode_moments(sd, cfg, x)
    assert _rel(mom, ref_mom) <= 3e-2, _rel(mom, ref_mom)
    z = torch.randn(B, 4, H // 8, W // 8, generator=g)
    img = m.decode(z.to(backend.device), return_dict=False)[0]
    backend.sync()
    ref_img = O.decode(sd, cfg, z)
    assert img.shape == ref_img.shape and _rel(img, ref_img) <= 3e-2, _rel(img, ref_img)
    u8 = m.decode_to_uint8(z.to(backend.device))
    backend.sync()
    d = (u8.cpu().int() - O.postprocess_uint8(ref_img).int()).abs()
    assert d.float().mean() <= 1.5 and (d > 8).float().mean() <= 0.01


@pytest.mark.gpu
def test_vae_full_size_decode(gpu_backend):
    """Full 83.65 M-parameter SD-2.1 VAE topology: decode one 64x88 latent (704x512 canvas) and encode it back."""
    cfg = O.VAEConfig()
    sd, m = _build(gpu_backend, cfg, seed=3)
    g = torch.Generator().manual_seed(2)
    z = torch.randn(1, 4, 64, 88, generator=g)
    img = m.decode(z.to(gpu_backend.device), return_dict=False)[0].clone()
    ref = O.decode(sd, cfg, z)
    assert _rel(img, ref) <= 3e-2, _rel(img, ref)
    x = ref.clamp(-1, 1)
    mom = m.encode(x.to(gpu_backend.device)).latent_dist.parameters
    assert _rel(mom, O.encode_moments(sd, cfg, x)) <= 3e-2
    # the driver's default batch (num_images_per_prompt = 8, stage2_batchtest_inpaint_model.py:196): 2.9 M output rows x 256 channels in
    # the last upsampling conv -- the 2 GiB range of the 32-bit buffer offsets is checked per element size (round 3: the check assumed
    # fp32 outputs everywhere and refused this launch).  Samples are independent: entry 0 of the batch == the single decode
    z8 = torch.cat([z, torch.randn(7, 4, 64, 88, generator=g)]).to(gpu_backend.device)
    img8 = m.decode(z8, return_dict=False)[0]
    assert img8.shape == (8, 3, 512, 704) and torch.isfinite(img8).all()
    r_ref, r_one = _rel(img8[:1], ref), _rel(img8[:1], img.cpu())   # (other tile choices at M x 8: bf16 re-rounding only)
    assert r_ref <= 3e-2 and r_one <= 2e-2, (r_ref, r_one)
    r_last = _rel(img8[7:], O.decode(sd, cfg, z8[7:].cpu()))         # the last sample lives at the far end of every buffer
    assert r_last <= 3e-2, r_last


@pytest.mark.gpu
def test_pipeline_with_vae_pixels(gpu_backend):
    """vae_image -> encode -> sampling loop -> decode -> uint8 pixels (ref stage2_inpaint_pipeline.py:443-445,528-532)
    end to end on the GPU vs the same chain through the oracles (tiny UNet + tiny VAE, injected posterior noise)."""
    from oracle.pipeline import stage2_sample, synth_inputs
    from oracle.schedulers import DDIMOracle
    from oracle.unet import UNetConfig, synth_state_dict
    from pcdms_amd import DDIMScheduler, Stage2_InapintUNet2DConditionModel, Stage2_InpaintDiffusionPipeline
    from tests.test_schedulers import SD21
    from tests.test_unet import _kwargs
    dev = gpu_backend.device
    ucfg, vcfg = UNetConfig.tiny(), O.VAEConfig.tiny()
    usd, vsd = synth_state_dict(ucfg, seed=0, random_affine=True), O.synth_state_dict(vcfg, 5)
    unet = Stage2_InapintUNet2DConditionModel(**_kwargs(ucfg))
    unet.load_state_dict(usd)
    unet.to(dev)
    vae = AutoencoderKL(block_out_channels=vcfg.block_out_channels)
    vae.load_state_dict(vsd)
    vae.to(dev)
    N, h, w, steps = 2, 16, 16, 4
    inp = synth_inputs(ucfg, h, w, N, L_img=7)
    g = torch.Generator().manual_seed(4)
    vae_image = torch.rand(1, 3, h * 8, w * 8, generator=g) * 2 - 1
    post_noise = torch.randn(1, 4, h, w, generator=g)
    # oracle chain
    ml = O.sample_latents(O.encode_moments(vsd, vcfg, vae_image), post_noise) * vcfg.scaling_factor
    inp_o = dict(inp, masked_latents=ml)
    lat = stage2_sample(usd, ucfg, DDIMOracle(), num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=steps, **inp_o)
    ref_u8 = O.postprocess_uint8(O.decode(vsd, vcfg, lat / vcfg.scaling_factor))
    # product chain: the pipeline calls vae.encode(...).latent_dist.sample(generator); inject the same posterior noise
    class _FixedNoiseVAE:
        config = vae.config

        def encode(self, x):
            d = vae.encode(x).latent_dist
            return type("E", (), {"latent_dist": type("D", (), {"sample": staticmethod(lambda generator=None: d.sample(noise=post_noise.to(dev)))})})

        decode = staticmethod(vae.decode)
        decode_to_uint8 = staticmethod(vae.decode_to_uint8)
    pipe = Stage2_InpaintDiffusionPipeline(unet, DDIMScheduler.from_config(SD21), vae=_FixedNoiseVAE())
    out = pipe(height=h * 8, width=w * 8, vae_image=vae_image.to(dev), s_img_proj_f=inp["s_img_proj_f"].to(dev),
               st_pose_f=inp["st_pose_f"].to(dev), pred_t_img_embed=inp["pred_t_img_embed"].to(dev),
               latents=inp["latents"].to(dev), num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=steps,
               output_type="uint8")
    assert _rel(out.latents, lat) <= 3e-2
    d = (out.images.cpu().int() - ref_u8.int()).abs().float()
    assert d.mean() <= 2.0 and (d > 12).float().mean() <= 0.02, (d.mean(), (d > 12).float().mean())
    pil = pipe._postprocess(out.latents, "pil")
    assert len(pil) == N and pil[0].size == (w * 8, h * 8)
