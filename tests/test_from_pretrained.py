"""The loading half of the drop-in boundary: ``from_pretrained`` / ``from_config`` on directories laid out like the
checkpoints the reference's drivers read (diffusers ``config.json`` + ``diffusion_pytorch_model.{safetensors,bin}``, transformers
``config.json`` + ``model.safetensors``), with the keyword arguments the drivers pass.  CPU only: loading never touches the GPU."""
from __future__ import annotations

import json

import pytest
import torch
from safetensors.torch import save_file

from oracle import prior as OP
from oracle import vae as OV
from oracle.unet import UNetConfig, synth_state_dict
import pcdms_amd as P

# stable-diffusion-2-1-base/unet/config.json, with tiny channel counts
SD21_UNET_JSON = {
    "_class_name": "UNet2DConditionModel", "_diffusers_version": "0.10.0.dev0", "act_fn": "silu", "attention_head_dim": [1, 2, 4, 4],
    "block_out_channels": [64, 128, 256, 256], "center_input_sample": False, "cross_attention_dim": 64,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"], "downsample_padding": 1,
    "dual_cross_attention": False, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 4, "layers_per_block": 2,
    "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": 32, "num_class_embeds": None, "only_cross_attention": False,
    "out_channels": 4, "sample_size": 16, "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    "use_linear_projection": True, "upcast_attention": False}


@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_unet_from_pretrained_like_the_stage2_driver(tmp_path, fmt):
    """stage2_batchtest_inpaint_model.py:125-130: stock 4-channel SD UNet on disk -> 9 input channels + class projection."""
    d = tmp_path / "sd21" / "unet"
    d.mkdir(parents=True)
    (d / "config.json").write_text(json.dumps(SD21_UNET_JSON))
    stock = UNetConfig.tiny(in_channels=4, class_embed_type=None, projection_class_embeddings_input_dim=None)
    sd = synth_state_dict(stock, seed=1)
    if fmt == "safetensors":
        save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "diffusion_pytorch_model.safetensors"))
    else:
        torch.save(sd, str(d / "diffusion_pytorch_model.bin"))
    kw = dict(subfolder="unet", in_channels=9, class_embed_type="projection", projection_class_embeddings_input_dim=64,
              torch_dtype=torch.float16, low_cpu_mem_usage=False)
    with pytest.raises(ValueError):
        P.Stage2_InapintUNet2DConditionModel.from_pretrained(tmp_path / "sd21", **kw)           # conv_in 4 -> 9 channels
    m = P.Stage2_InapintUNet2DConditionModel.from_pretrained(tmp_path / "sd21", ignore_mismatched_sizes=True, **kw)
    assert m.config.in_channels == 9 and m.config.class_embed_type == "projection" and m.dtype == torch.float16
    assert m.config.sample_size == 16 and m.config._diffusers_version
    got = m.state_dict()
    assert got["conv_in.weight"].shape == (64, 9, 3, 3)                                          # re-initialised
    assert got["class_embedding.linear_1.weight"].shape == (256, 64)                             # absent on disk: fresh
    for k in ("mid_block.resnets.0.conv1.weight", "up_blocks.3.attentions.2.transformer_blocks.0.attn2.to_k.weight", "conv_out.bias"):
        assert torch.equal(got[k], sd[k].float()), k
    # ... then the driver overwrites everything with the trained checkpoint (strict)
    full = synth_state_dict(UNetConfig.tiny(), seed=2)
    m.load_state_dict(full)
    assert torch.equal(m.state_dict()["conv_in.weight"], full["conv_in.weight"])
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in full.items() if k != "conv_out.bias"})
    # stage-3: the stock class, 8 input channels (stage3_batchtest_refined_model.py:121-126)
    m3 = P.UNet2DConditionModel.from_pretrained(tmp_path / "sd21", subfolder="unet", in_channels=8, low_cpu_mem_usage=False,
                                                ignore_mismatched_sizes=True)
    assert m3.state_dict()["conv_in.weight"].shape == (64, 8, 3, 3) and m3.config.class_embed_type is None


def test_vae_from_pretrained(tmp_path):
    d = tmp_path / "sd21" / "vae"
    d.mkdir(parents=True)
    cfg = OV.VAEConfig.tiny()
    (d / "config.json").write_text(json.dumps({
        "_class_name": "AutoencoderKL", "_diffusers_version": "0.10.0.dev0", "act_fn": "silu", "block_out_channels": list(cfg.block_out_channels),
        "down_block_types": ["DownEncoderBlock2D"] * 4, "in_channels": 3, "latent_channels": 4, "layers_per_block": 2, "norm_num_groups": 32,
        "out_channels": 3, "sample_size": 768, "up_block_types": ["UpDecoderBlock2D"] * 4, "scaling_factor": 0.18215}))
    sd = OV.synth_state_dict(cfg, 3)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "diffusion_pytorch_model.safetensors"))
    vae = P.AutoencoderKL.from_pretrained(tmp_path / "sd21", subfolder="vae", torch_dtype=torch.float16)
    assert tuple(vae.config.block_out_channels) == cfg.block_out_channels and vae.config.scaling_factor == 0.18215
    assert len(vae.config.block_out_channels) == 4                          # -> vae_scale_factor 8 (stage2_inpaint_pipeline.py:138)
    assert all(torch.equal(vae.state_dict()[k], v.float()) for k, v in sd.items())
    with pytest.raises(FileNotFoundError):
        P.AutoencoderKL.from_pretrained(tmp_path / "nowhere", subfolder="vae")


def test_prior_from_pretrained_like_the_stage1_driver(tmp_path):
    """stage1_batchtest_prior_model.py:55-59: Kandinsky-2.2 prior directory (embedding_dim 1280, 77 + 4 tokens), re-shaped by kwargs."""
    d = tmp_path / "k22" / "prior"
    d.mkdir(parents=True)
    (d / "config.json").write_text(json.dumps({"_class_name": "PriorTransformer", "num_attention_heads": 2, "attention_head_dim": 64,
                                               "num_layers": 2, "embedding_dim": 1280, "num_embeddings": 77, "additional_embeddings": 4,
                                               "dropout": 0.0}))
    ours = OP.synth_state_dict(OP.PriorConfig.tiny(), 4)
    on_disk = {k: v for k, v in ours.items() if k.startswith("transformer_blocks") or k.startswith("norm_out") or k.startswith("time_embedding")}
    on_disk["proj_in.weight"] = torch.randn(128, 1280)                      # embedding_dim 1280 on disk -> mismatched
    on_disk["positional_embedding"] = torch.randn(1, 81, 128)               # 77 + 4 tokens on disk -> mismatched
    save_file({k: v.contiguous() for k, v in on_disk.items()}, str(d / "diffusion_pytorch_model.safetensors"))
    kw = dict(subfolder="prior", num_embeddings=2, embedding_dim=1024, low_cpu_mem_usage=False)
    with pytest.raises(RuntimeError):
        P.Stage1_PriorTransformer.from_pretrained(tmp_path / "k22", **kw)
    m = P.Stage1_PriorTransformer.from_pretrained(tmp_path / "k22", ignore_mismatched_sizes=True, **kw)
    got = m.state_dict()
    assert m.config.embedding_dim == 1024 and m.num_tokens == 6 and got["proj_in.weight"].shape == (128, 1024)
    assert got["positional_embedding"].shape == (1, 6, 128) and "pose_encoder.net.0.weight" in got
    assert torch.equal(got["transformer_blocks.1.ff.net.2.weight"], ours["transformer_blocks.1.ff.net.2.weight"])
    m.load_state_dict(ours)                                                 # then the trained checkpoint, strict (:58-59)


def test_encoders_from_pretrained_transformers_layout(tmp_path):
    transformers = pytest.importorskip("transformers")
    from tests.test_encoders import TINY, _hf, _hf_clip
    _, hf = _hf(TINY, seed=5)
    hf.save_pretrained(tmp_path / "dino")
    m = P.Dinov2Model.from_pretrained(tmp_path / "dino")
    assert m.config.hidden_size == 128 and m.config.use_swiglu_ffn
    assert all(torch.equal(m.state_dict()[k], v) for k, v in hf.state_dict().items())
    _, hfc = _hf_clip(dict(hidden_size=320, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4, image_size=28, patch_size=14,
                           hidden_act="gelu", projection_dim=64), seed=6)
    hfc.save_pretrained(tmp_path / "clip")
    c = P.CLIPVisionModelWithProjection.from_pretrained(tmp_path / "clip")
    assert c.config.projection_dim == 64 and c.config.num_attention_heads == 4
    assert all(torch.equal(c.state_dict()[k], v) for k, v in hfc.state_dict().items() if not k.endswith("position_ids"))


def test_scheduler_from_config_of_another_scheduler():
    """stage2_batchtest_inpaint_model.py:132: UniPCMultistepScheduler.from_config(pipe.scheduler.config) where the pipeline on disk
    holds SD-2.1's PNDM scheduler config: shared keys are taken, foreign ones do not act but stay in ``.config`` (diffusers keeps
    them as hidden attributes, which is what makes scheduler swapping round-trip)."""
    pndm = {"_class_name": "PNDMScheduler", "_diffusers_version": "0.10.0.dev0", "beta_end": 0.012, "beta_schedule": "scaled_linear",
            "beta_start": 0.00085, "clip_sample": False, "num_train_timesteps": 1000, "prediction_type": "epsilon", "set_alpha_to_one": False,
            "skip_prk_steps": True, "steps_offset": 1, "trained_betas": None}
    u = P.UniPCMultistepScheduler.from_config(pndm)
    assert u.config.beta_schedule == "scaled_linear" and u.config.beta_end == 0.012 and u.config.solver_order == 2
    assert u.config.skip_prk_steps is True and u.config.clip_sample is False
    d = P.DDIMScheduler.from_config(u.config)                               # config objects round-trip between schedulers
    assert d.config.steps_offset == 1 and d.config.clip_sample is False and d.init_noise_sigma == 1.0
    k = P.UnCLIPScheduler.from_config({"_class_name": "UnCLIPScheduler", "clip_sample": True, "clip_sample_range": 10.0,
                                       "num_train_timesteps": 1000, "prediction_type": "sample", "variance_type": "fixed_small_log"})
    assert k.config.prediction_type == "sample" and k.config.clip_sample_range == 10.0


def test_module_surface_the_drivers_touch():
    """``.eval() / .half() / .float() / .requires_grad_(False) / .parameters() / .modules()`` on every model object (the drivers chain
    ``.to(device).eval()``, stage2_batchtest_inpaint_model.py:95-99); training is out of scope and says so."""
    objs = [P.Stage2_InapintUNet2DConditionModel(**{k: v for k, v in SD21_UNET_JSON.items() if k in ("block_out_channels", "attention_head_dim", "cross_attention_dim")}),
            P.AutoencoderKL(block_out_channels=(64, 64, 128, 128)), P.Stage1_PriorTransformer(num_attention_heads=2, num_layers=1, embedding_dim=1024, num_embeddings=2),
            P.Dinov2Model(hidden_size=128, num_hidden_layers=1, num_attention_heads=2), P.CLIPVisionModelWithProjection(hidden_size=128, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2),
            P.ControlNetConditioningEmbedding(64), P.ImageProjModel_p(128, 64, 64)]
    for m in objs:
        assert m.eval() is m and m.requires_grad_(False) is m and m.half() is m and m.float() is m and m.train(False) is m
        assert isinstance(list(m.parameters()), list) and isinstance(list(m.modules()), list)
        with pytest.raises(NotImplementedError):
            m.train()
    iproj = objs[-1]
    iproj.load_state_dict({k: torch.zeros(s) for k, s in iproj.expected_shapes().items()})
    assert sum(p.numel() for p in iproj.parameters()) == 128 * 64 + 64 + 2 * 64 + 64 * 64 + 64
