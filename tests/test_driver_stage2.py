"""The reference's stage-2 driver flow end to end on the GPU (tools/stage2_batchtest_inpaint_model.py): checkpoint directories and
``mp_rank_00_model_states.pt`` in the layouts the reference reads, PNG / JPG inputs, ``test_`` (stage-1 embeddings from .npy) and
``train_`` (CLIP embedding of the target) json modes, grid and best-SSIM outputs.  Tiny models with random weights: this checks the
plumbing a user switching from the reference depends on (every network is parity-tested on its own elsewhere)."""
from __future__ import annotations

import importlib.util
import json
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image
from safetensors.torch import save_file

pytest.importorskip("transformers")


def _load_driver(name="stage2_batchtest_inpaint_model"):
    p = Path(__file__).resolve().parent.parent / "tools" / f"{name}.py"
    spec = importlib.util.spec_from_file_location(name, p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
def test_stage2_driver_flow(gpu_backend, tmp_path):
    from oracle import cond as OC
    from oracle import vae as OV
    from oracle.unet import UNetConfig, synth_state_dict
    from tests.test_encoders import TINY, _hf, _hf_clip
    from tests.test_from_pretrained import SD21_UNET_JSON
    drv = _load_driver()
    # ---- stable-diffusion-2-1-base-like directory
    sd21 = tmp_path / "sd21"
    for sub in ("unet", "vae", "scheduler"):
        (sd21 / sub).mkdir(parents=True)
    (sd21 / "unet" / "config.json").write_text(json.dumps(SD21_UNET_JSON))
    stock = UNetConfig.tiny(in_channels=4, class_embed_type=None, projection_class_embeddings_input_dim=None)
    save_file({k: v.contiguous() for k, v in synth_state_dict(stock, seed=1).items()}, str(sd21 / "unet" / "diffusion_pytorch_model.safetensors"))
    vcfg = OV.VAEConfig.tiny()
    (sd21 / "vae" / "config.json").write_text(json.dumps({"block_out_channels": list(vcfg.block_out_channels), "in_channels": 3, "out_channels": 3,
                                                          "latent_channels": 4, "layers_per_block": 2, "norm_num_groups": 32, "scaling_factor": 0.18215}))
    save_file({k: v.contiguous() for k, v in OV.synth_state_dict(vcfg, 2).items()}, str(sd21 / "vae" / "diffusion_pytorch_model.safetensors"))
    (sd21 / "scheduler" / "scheduler_config.json").write_text(json.dumps({
        "_class_name": "PNDMScheduler", "beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085, "clip_sample": False,
        "num_train_timesteps": 1000, "prediction_type": "epsilon", "set_alpha_to_one": False, "skip_prk_steps": True, "steps_offset": 1}))
    # ---- encoders (transformers layout)
    _, hf_dino = _hf(TINY, seed=3)
    hf_dino.save_pretrained(tmp_path / "dinov2")
    _, hf_clip = _hf_clip(dict(hidden_size=320, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4, image_size=224, patch_size=14,
                               hidden_act="gelu", projection_dim=64), seed=4)
    hf_clip.save_pretrained(tmp_path / "clip")
    # ---- the trained stage-2 checkpoint (DeepSpeed layout: {"module": {prefix.key: tensor}})
    ucfg = UNetConfig.tiny()
    module = {"unet." + k: v for k, v in synth_state_dict(ucfg, seed=5, random_affine=True).items()}
    module.update({"pose_proj." + k: v for k, v in OC.synth(OC.pose_param_shapes(ucfg.block_out_channels[0], 3, (16, 32, 96, 256)), 6).items()})
    module.update({"image_proj_model_p." + k: v for k, v in OC.synth(OC.image_proj_param_shapes(128, 64, ucfg.cross_attention_dim), 7, 1.0).items()})
    (tmp_path / "ckpt").mkdir()
    torch.save({"module": module}, tmp_path / "ckpt" / "mp_rank_00_model_states.pt")
    # ---- data: two pairs, poses, stage-1 embeddings
    rng = np.random.default_rng(0)
    for d in ("img", "pose", "embed"):
        (tmp_path / d).mkdir()
    names = ["a", "b", "c"]
    for n in names:
        Image.fromarray(rng.integers(0, 255, (150, 90, 3), dtype=np.uint8)).save(tmp_path / "img" / f"{n}.png")
        Image.fromarray(rng.integers(0, 255, (150, 90, 3), dtype=np.uint8)).save(tmp_path / "pose" / f"{n}_pose.jpg")
    pairs = [{"source_image": "a.jpg", "target_image": "b.jpg"}, {"source_image": "b.jpg", "target_image": "c.jpg"}]
    other_first = {"source_image": "c.jpg", "target_image": "a.jpg"}
    for p in pairs + [other_first]:
        np.save(tmp_path / "embed" / (p["source_image"].replace(".jpg", "_to_") + p["target_image"].replace(".jpg", ".npy")),
                rng.standard_normal((1, 64)).astype(np.float32) * 0.4)
    W, H = 64, 128
    base = ["--pretrained_model_name_or_path", str(sd21), "--image_encoder_g_path", str(tmp_path / "clip"), "--image_encoder_p_path",
            str(tmp_path / "dinov2"), "--img_path", str(tmp_path / "img") + "/", "--pose_path", str(tmp_path / "pose") + "/",
            "--target_embed_path", str(tmp_path / "embed") + "/", "--save_path", str(tmp_path / "out"), "--num_inference_steps", "3",
            "--img_width", str(W), "--img_height", str(H), "--weights_name", str(tmp_path / "ckpt")]
    # "test" json: stage-1 embeddings from disk, grid output (2 rows x 3: source|target, poses, 4 samples)
    (tmp_path / "test_data.json").write_text(json.dumps(pairs))
    args = drv.build_parser().parse_args(base + ["--json_path", str(tmp_path / "test_data.json")])
    drv.inference(args, 0, pairs)
    show = tmp_path / "out" / "show_guidancescale2.0_seed42_numsteps3"
    grids = sorted(show.glob("*.png"))
    assert [g.name for g in grids] == ["a_to_b.png", "b_to_c.png"]
    assert Image.open(grids[0]).size == (3 * 2 * W, 2 * H)
    # pair 2's result must not depend on WHICH pair came first (the driver re-uses one pipe / one UNet for every pair,
    # /root/reference/stage2_batchtest_inpaint_model.py:141-200; round 1 re-used pair 1's cross-attention K/V for pair 2).
    # Same tensor shapes => the shared generator is in the same state when pair 2 starts, so the PNG must be IDENTICAL.
    args_b = drv.build_parser().parse_args(base[:base.index("--save_path")] + ["--save_path", str(tmp_path / "out_b")] +
                                           base[base.index("--save_path") + 2:] + ["--json_path", str(tmp_path / "test_data.json")])
    drv.inference(args_b, 0, [other_first, pairs[1]])
    g_a = np.asarray(Image.open(grids[1]))
    g_b = np.asarray(Image.open(tmp_path / "out_b" / "show_guidancescale2.0_seed42_numsteps3" / "b_to_c.png"))
    assert np.array_equal(g_a, g_b), float(np.abs(g_a.astype(int) - g_b.astype(int)).mean())
    g_first = np.asarray(Image.open(tmp_path / "out_b" / "show_guidancescale2.0_seed42_numsteps3" / "c_to_a.png"))
    assert not np.array_equal(g_first[H:], np.asarray(Image.open(grids[0]))[H:])   # (the first pairs did differ)
    # "train" json: CLIP embedding of the target image, best-SSIM sample saved as the 64 x 128 target half
    (tmp_path / "train_data.json").write_text(json.dumps(pairs[:1]))
    args = drv.build_parser().parse_args(base + ["--json_path", str(tmp_path / "train_data.json"), "--calculate_metrics"])
    ssims = drv.inference(args, 0, pairs[:1])
    best = tmp_path / "out" / "guidancescale2.0_seed42_numsteps3" / "a_to_b.png"
    assert len(ssims) == 1 and -1.0 <= ssims[0] <= 1.0 and Image.open(best).size == (W, H)


def test_ssim_restatement_properties():
    drv = _load_driver()
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 255, (40, 30, 3))
    assert abs(drv.ssim_gaussian(a, a) - 1.0) < 1e-12
    b = np.clip(a + rng.normal(0, 25, a.shape), 0, 255)
    s1, s2 = drv.ssim_gaussian(a, b), drv.ssim_gaussian(a, np.clip(a + rng.normal(0, 80, a.shape), 0, 255))
    assert 0 < s2 < s1 < 1


@pytest.mark.gpu
def test_three_drivers_chained(gpu_backend, tmp_path):
    """stage-1 driver -> <s>_to_<t>.npy -> stage-2 driver (test json, best-SSIM png) -> stage-3 driver, on fabricated checkpoints in the
    directory layouts of the three reference drivers (Kandinsky-2.2 prior dir, SD-2.1 dir, transformers encoder dirs, DeepSpeed .pt)."""
    from oracle import cond as OC
    from oracle import prior as OP
    from oracle import vae as OV
    from oracle.unet import UNetConfig, synth_state_dict
    from tests.test_encoders import TINY, _hf, _hf_clip
    from tests.test_from_pretrained import SD21_UNET_JSON
    d1, d2, d3 = (_load_driver(n) for n in ("stage1_batchtest_prior_model", "stage2_batchtest_inpaint_model", "stage3_batchtest_refined_model"))
    rng = np.random.default_rng(0)
    for d in ("img", "pose", "posetxt", "ck1", "ck2", "ck3"):
        (tmp_path / d).mkdir()
    for n in ("a", "b"):
        Image.fromarray(rng.integers(0, 255, (150, 90, 3), dtype=np.uint8)).save(tmp_path / "img" / f"{n}.png")
        Image.fromarray(rng.integers(0, 255, (150, 90, 3), dtype=np.uint8)).save(tmp_path / "pose" / f"{n}_pose.jpg")
        (tmp_path / "posetxt" / f"{n}.txt").write_text("\n".join(f"{x:.4f} {y:.4f}" for x, y in rng.uniform(0, 1, (18, 2))))
    pairs = [{"source_image": "a.jpg", "target_image": "b.jpg"}]
    (tmp_path / "test_data.json").write_text(json.dumps(pairs))
    # ---- encoders
    _, hf_dino = _hf(TINY, seed=3)
    hf_dino.save_pretrained(tmp_path / "dinov2")
    _, hf_clip = _hf_clip(dict(hidden_size=320, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4, image_size=224, patch_size=14,
                               hidden_act="gelu", projection_dim=1024), seed=4)
    hf_clip.save_pretrained(tmp_path / "clip")
    # ---- stage 1: kandinsky-2-2-prior-like dir (config of the stock prior + scheduler), DeepSpeed checkpoint of the trained prior
    k22 = tmp_path / "k22"
    (k22 / "prior").mkdir(parents=True)
    (k22 / "scheduler").mkdir()
    (k22 / "prior" / "config.json").write_text(json.dumps({"num_attention_heads": 2, "attention_head_dim": 64, "num_layers": 2, "embedding_dim": 1280,
                                                           "num_embeddings": 77, "additional_embeddings": 4}))
    (k22 / "scheduler" / "scheduler_config.json").write_text(json.dumps({"_class_name": "UnCLIPScheduler", "clip_sample": True, "clip_sample_range": 10.0,
                                                                         "num_train_timesteps": 1000, "prediction_type": "sample",
                                                                         "variance_type": "fixed_small_log"}))
    torch.save({"module": OP.synth_state_dict(OP.PriorConfig.tiny(), 1)}, tmp_path / "ck1" / "mp_rank_00_model_states.pt")
    a1 = d1.build_parser().parse_args(["--pretrained_model_name_or_path", str(k22), "--image_encoder_path", str(tmp_path / "clip"), "--img_path",
                                       str(tmp_path / "img") + "/", "--pose_path", str(tmp_path / "posetxt") + "/", "--save_path", str(tmp_path / "s1"),
                                       "--num_inference_steps", "4", "--weights_name", str(tmp_path / "ck1")])
    sims = d1.main(a1, 0, pairs)
    emb_dir = tmp_path / "s1" / "guidancescale0_seed42_numsteps4"
    emb = np.load(emb_dir / "a_to_b.npy")
    assert emb.shape == (1, 1024) and np.isfinite(emb).all() and len(sims) == 1 and (emb_dir / "a_results.txt").exists()
    # ---- stage 2 (context / class-projection width 1024 = the prior's embedding width)
    sd21 = tmp_path / "sd21"
    for sub in ("unet", "vae", "scheduler"):
        (sd21 / sub).mkdir(parents=True)
    (sd21 / "unet" / "config.json").write_text(json.dumps(dict(SD21_UNET_JSON, cross_attention_dim=1024)))
    vcfg = OV.VAEConfig.tiny()
    (sd21 / "vae" / "config.json").write_text(json.dumps({"block_out_channels": list(vcfg.block_out_channels), "scaling_factor": 0.18215}))
    save_file({k: v.contiguous() for k, v in OV.synth_state_dict(vcfg, 2).items()}, str(sd21 / "vae" / "diffusion_pytorch_model.safetensors"))
    (sd21 / "scheduler" / "scheduler_config.json").write_text(json.dumps({"_class_name": "PNDMScheduler", "beta_end": 0.012, "beta_schedule": "scaled_linear",
                                                                          "beta_start": 0.00085, "num_train_timesteps": 1000, "steps_offset": 1}))
    ucfg = UNetConfig.tiny(cross_attention_dim=1024, projection_class_embeddings_input_dim=1024)
    iproj = OC.synth(OC.image_proj_param_shapes(128, 64, 1024), 7, 1.0)
    module = {"unet." + k: v for k, v in synth_state_dict(ucfg, seed=5, random_affine=True).items()}
    module.update({"pose_proj." + k: v for k, v in OC.synth(OC.pose_param_shapes(ucfg.block_out_channels[0], 3, (16, 32, 96, 256)), 6).items()})
    module.update({"image_proj_model_p." + k: v for k, v in iproj.items()})
    torch.save({"module": module}, tmp_path / "ck2" / "mp_rank_00_model_states.pt")
    W, H = 64, 128
    common = ["--pretrained_model_name_or_path", str(sd21), "--image_encoder_p_path", str(tmp_path / "dinov2"), "--img_path", str(tmp_path / "img") + "/",
              "--pose_path", str(tmp_path / "pose") + "/", "--json_path", str(tmp_path / "test_data.json"), "--num_inference_steps", "3",
              "--img_width", str(W), "--img_height", str(H), "--calculate_metrics"]
    a2 = d2.build_parser().parse_args(common + ["--image_encoder_g_path", str(tmp_path / "clip"), "--target_embed_path", str(emb_dir) + "/",
                                                "--save_path", str(tmp_path / "s2"), "--weights_name", str(tmp_path / "ck2")])
    d2.inference(a2, 0, pairs)
    s2_dir = tmp_path / "s2" / "guidancescale2.0_seed42_numsteps3"
    assert Image.open(s2_dir / "a_to_b.png").size == (W, H)
    # ---- stage 3 (stock UNet, 8 input channels) on the stage-2 result
    u3 = UNetConfig.tiny(in_channels=8, cross_attention_dim=1024, class_embed_type=None, projection_class_embeddings_input_dim=None)
    module3 = {"unet." + k: v for k, v in synth_state_dict(u3, seed=8, random_affine=True).items()}
    module3.update({"image_proj_model_p." + k: v for k, v in iproj.items()})
    torch.save({"module": module3}, tmp_path / "ck3" / "mp_rank_00_model_states.pt")
    a3 = d3.build_parser().parse_args(common + ["--gen_t_img_path", str(s2_dir) + "/", "--save_path", str(tmp_path / "s3"), "--weights_name", str(tmp_path / "ck3")])
    ss = d3.inference(a3, 0, pairs)
    out = tmp_path / "s3" / "guidancescale2.0_seed42_numsteps3" / "a_to_b.png"
    assert len(ss) == 1 and Image.open(out).size == (W, H)
