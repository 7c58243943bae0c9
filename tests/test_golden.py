"""Golden fixtures produced by the REFERENCE's own forward / __call__ (tests/golden/README in
make_reference_wiring_fixtures.py).  CPU: the oracle reproduces them.  GPU: the HIP path reproduces them.

Tolerances: the UNet fixture used fp32 inputs -> oracle must agree to fp32 round-off (1e-5).  The pipeline
fixtures include the reference's hard-coded ``.half()`` casts of every UNet input (Appendix C-3), so the
fp32 oracle agrees to 1e-2 rel-L2 over 4 steps; the bf16 HIP path to 5e-2.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest
import torch

from oracle.pipeline import stage2_sample
from oracle.schedulers import DDIMOracle, UniPCOracle
from oracle.unet import UNetConfig, synth_state_dict, unet_forward

G = Path(__file__).resolve().parent / "golden"


def _load(name):
    z = np.load(G / name)
    cfg = UNetConfig(block_out_channels=tuple(int(v) for v in z["block_out_channels"]),
                     attention_head_dim=tuple(int(v) for v in z["attention_head_dim"]),
                     cross_attention_dim=int(z["cross_attention_dim"]),
                     projection_class_embeddings_input_dim=int(z["projection_class_embeddings_input_dim"]),
                     sample_size=int(z["sample_size"]))
    sd = synth_state_dict(cfg, seed=int(z["seed"]), random_affine=True)
    from tests.golden.make_reference_wiring_fixtures import weights_checksum
    if abs(weights_checksum(sd) - float(z["weights_checksum"])) > 1e-6 * float(z["weights_checksum"]):
        pytest.skip(f"seeded weight generation differs from the fixture's (torch {z['torch_version']} vs {torch.__version__})")
    return z, cfg, sd


def _t(z, k):
    return torch.from_numpy(np.asarray(z[k]))


def _rel(a, b):
    return ((a.float().cpu() - b).norm() / b.norm()).item()


def test_oracle_reproduces_reference_forward():
    z, cfg, sd = _load("ref_wiring_unet.npz")
    eps = unet_forward(sd, cfg, _t(z, "sample"), torch.tensor(int(z["timestep"])), _t(z, "ehs"), _t(z, "class_labels"),
                       _t(z, "pose"))
    assert torch.allclose(eps, _t(z, "eps"), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("kind", ["ddim", "unipc", "ddim_gr07"])
def test_oracle_reproduces_reference_pipeline(kind):
    """``ddim_gr07``: the reference's loop with guidance_rescale=0.7, i.e. its own rescale_noise_cfg (ref :510-516) inside the loop."""
    z, cfg, sd = _load(f"ref_wiring_pipeline_{kind}.npz")
    sch = UniPCOracle() if kind == "unipc" else DDIMOracle()
    trace = []
    out = stage2_sample(sd, cfg, sch, masked_latents=_t(z, "masked_latents"), s_img_proj_f=_t(z, "s_img_proj_f"),
                        st_pose_f=_t(z, "st_pose_f"), pred_t_img_embed=_t(z, "pred_t_img_embed"),
                        latents=_t(z, "latents"), num_images_per_prompt=int(z["N"]), guidance_scale=2.0,
                        num_inference_steps=int(z["steps"]), guidance_rescale=float(z["guidance_rescale"]) if "guidance_rescale" in z else 0.0)
    assert _rel(out, _t(z, "final_latents")) < 1e-2, _rel(out, _t(z, "final_latents"))


def _product_unet(cfg, sd, dev):
    from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel
    from tests.test_unet import _kwargs
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(sd)
    return m.to(dev)


@pytest.mark.gpu
def test_hip_reproduces_reference_forward(gpu_backend):
    z, cfg, sd = _load("ref_wiring_unet.npz")
    dev = gpu_backend.device
    m = _product_unet(cfg, sd, dev)
    eps = m(_t(z, "sample").to(dev), torch.tensor(int(z["timestep"]), device=dev), _t(z, "ehs").to(dev),
            class_labels=_t(z, "class_labels").to(dev), my_pose_cond=_t(z, "pose").to(dev)).sample
    assert _rel(eps, _t(z, "eps")) < 2.5e-2


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ddim", "unipc", "ddim_gr07"])
def test_hip_reproduces_reference_pipeline(gpu_backend, kind):
    from pcdms_amd.pipeline import Stage2_InpaintDiffusionPipeline
    from pcdms_amd.schedulers import DDIMScheduler, UniPCMultistepScheduler
    from tests.test_schedulers import SD21
    z, cfg, sd = _load(f"ref_wiring_pipeline_{kind}.npz")
    dev = gpu_backend.device
    m = _product_unet(cfg, sd, dev)
    sch = (UniPCMultistepScheduler if kind == "unipc" else DDIMScheduler).from_config(SD21)
    pipe = Stage2_InpaintDiffusionPipeline(m, sch)
    h, w = z["latents"].shape[-2:]
    out = pipe(height=h * 8, width=w * 8, masked_latents=_t(z, "masked_latents").to(dev),
               s_img_proj_f=_t(z, "s_img_proj_f").to(dev), st_pose_f=_t(z, "st_pose_f").to(dev),
               pred_t_img_embed=_t(z, "pred_t_img_embed").to(dev), latents=_t(z, "latents").to(dev),
               num_images_per_prompt=int(z["N"]), guidance_scale=2.0, num_inference_steps=int(z["steps"]),
               guidance_rescale=float(z["guidance_rescale"]) if "guidance_rescale" in z else 0.0, output_type="latent").latents
    assert _rel(out, _t(z, "final_latents")) < 5e-2, _rel(out, _t(z, "final_latents"))


def test_reference_precision_budget_fixture_and_emulation_hook():
    """tests/golden/fp16_budget.npz (make_fp16_budget.py: the reference's own fp16 numerics against the fp32 oracle) is self-consistent, its fp32
    leg IS the committed configs[0] oracle run, and ``oracle.unet.ROUND_DTYPE`` is a no-op unless set: ``None`` reproduces the plain fp32 forward
    bit for bit, fp16 / bf16 change it by their rounding (tiny configuration, CPU)."""
    import json

    import oracle.unet as OU
    bx = np.load(G / "fp16_budget.npz")
    budget = json.loads(str(bx["json"]))
    c0 = np.load(G / "fullsize_config0.npz")
    assert np.abs(bx["config0_final_fp32"] - c0["lat_final"]).max() <= 1e-5 * np.abs(c0["lat_final"]).max()
    for i in ("0", "10", "25"):
        b = budget["forward_configs1"][i]
        assert 5e-4 < b["fp16ref_vs_fp32"] < 5e-3 < b["bf16ref_vs_fp32"] < 4e-2 and b["fp32_vs_stored_fixture"] < 1e-3   # (the stored eps are fp16)
        assert 4 < b["bf16ref_vs_fp32"] / b["fp16ref_vs_fp32"] < 12                     # 3 mantissa bits
    f16, b16 = budget["config0"]["fp16"], budget["config0"]["bf16"]
    assert f16["final_latents"] < b16["final_latents"] and f16["per_step_latents"][0] < f16["per_step_latents"][-1]
    ref = torch.from_numpy(bx["config0_final_fp32"])
    for key, r in (("config0_final_fp16", f16["final_latents"]), ("config0_final_bf16", b16["final_latents"])):
        got = ((torch.from_numpy(bx[key]) - ref).norm() / ref.norm()).item()
        assert abs(got - r) <= 1e-3 * r
    # the hook
    cfg = UNetConfig.tiny()
    sd = synth_state_dict(cfg, seed=3, random_affine=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 9, 16, 16, generator=g)
    ehs = torch.randn(2, 5, cfg.cross_attention_dim, generator=g)
    cl = torch.randn(2, 1, cfg.projection_class_embeddings_input_dim, generator=g)
    pose = torch.randn(2, cfg.block_out_channels[0], 16, 16, generator=g) * 0.1
    assert OU.ROUND_DTYPE is None
    with torch.no_grad():
        base = unet_forward(sd, cfg, x, 500, ehs, cl, pose)
        outs = {}
        for dt in (torch.float16, torch.bfloat16):
            OU.ROUND_DTYPE = dt
            try:
                sdq = {k: v.to(dt).float() for k, v in sd.items()}
                outs[dt] = unet_forward(sdq, cfg, x.to(dt).float(), 500, ehs.to(dt).float(), cl.to(dt).float(), pose.to(dt).float())
            finally:
                OU.ROUND_DTYPE = None
        again = unet_forward(sd, cfg, x, 500, ehs, cl, pose)
    assert torch.equal(base, again)
    r16 = ((outs[torch.float16] - base).norm() / base.norm()).item()
    rbf = ((outs[torch.bfloat16] - base).norm() / base.norm()).item()
    assert 1e-4 < r16 < 3e-3 < rbf < 1e-1 and 4 < rbf / r16 < 12 and torch.equal(outs[torch.float16], outs[torch.float16].half().float())
