"""UNet forward parity: pcdms_amd.Stage2_InapintUNet2DConditionModel (HIP) vs the fp32 CPU oracle.

Tolerance (stated, SURVEY.md §7 iv): the HIP path computes bf16 x bf16 -> fp32 with bf16 activations
between kernels; against the fp32 oracle on identical weights/inputs we require
rel-L2(eps) <= 2.5e-2 and max-abs <= 6% of max|eps| for one forward (61 norm layers deep).
"""
from __future__ import annotations

import pytest
import torch

from oracle.unet import UNetConfig, param_count, param_shapes, synth_state_dict, unet_forward
from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel

REL_L2_TOL = 2.5e-2
MAX_ABS_TOL = 6e-2


def _kwargs(cfg: UNetConfig):
    return dict(in_channels=cfg.in_channels, block_out_channels=cfg.block_out_channels,
                attention_head_dim=cfg.attention_head_dim, cross_attention_dim=cfg.cross_attention_dim,
                use_linear_projection=True, class_embed_type=cfg.class_embed_type,
                projection_class_embeddings_input_dim=cfg.projection_class_embeddings_input_dim,
                sample_size=cfg.sample_size)


def _inputs(cfg, B, h, w, L, seed=0):
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(B, 9, h, w, generator=g)
    ehs = torch.randn(B, L, cfg.cross_attention_dim, generator=g)
    ehs[: B // 2] = 0  # uncond rows, as the pipeline builds them (ref stage2_inpaint_pipeline.py:455-458)
    cl = torch.randn(B, 1, cfg.projection_class_embeddings_input_dim, generator=g) * 0.4
    pose = torch.randn(1, cfg.block_out_channels[0], h, w, generator=g) * 0.1
    return sample, ehs, cl, pose


def _check(out, ref):
    out, ref = out.float().cpu(), ref.float()
    assert torch.isfinite(out).all()
    rel = ((out - ref).norm() / ref.norm()).item()
    mx = ((out - ref).abs().max() / ref.abs().max()).item()
    assert rel <= REL_L2_TOL and mx <= MAX_ABS_TOL, f"rel-L2 {rel:.4f}, max-abs/scale {mx:.4f}"
    return rel, mx


def test_state_dict_contract():
    """Key names / shapes / parameter count agree between product and oracle (SURVEY.md §8c a)."""
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(UNetConfig()))
    exp = m.expected_shapes()
    ref = dict(param_shapes(UNetConfig()))
    assert exp == {k: tuple(v) for k, v in ref.items()}
    assert sum(torch.Size(s).numel() for s in exp.values()) == 868_876_804 == param_count(UNetConfig())
    with pytest.raises(RuntimeError):
        m.load_state_dict({"conv_in.weight": torch.zeros(320, 9, 3, 3)})
    with pytest.raises(NotImplementedError):
        Stage2_InapintUNet2DConditionModel(addition_embed_type="text")


def test_forward_errors_and_cpu_refusal():
    cfg = UNetConfig.tiny()
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(synth_state_dict(cfg, 0))
    s, e, c, p = _inputs(cfg, 2, 8, 8, 5)
    with pytest.raises(ValueError):
        m(s, 10, e, class_labels=None, my_pose_cond=p)
    with pytest.raises(NotImplementedError):
        m(s, 10, e, class_labels=c, my_pose_cond=p, attention_mask=torch.ones(2, 5))
    from pcdms_amd import _lib
    _lib._lib = None  # product library (or none at all): CPU tensors must be refused, never silently computed
    try:
        with pytest.raises(RuntimeError):
            m(s, 10, e, class_labels=c, my_pose_cond=p)
    finally:
        _lib._lib = None


def _run(backend, cfg, B, h, w, L, random_affine=True, t=981):
    sd = synth_state_dict(cfg, seed=0, random_affine=random_affine)
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(sd)
    m.to(backend.device)
    s, e, c, p = _inputs(cfg, B, h, w, L)
    dev = backend.device
    out = m(s.to(dev), torch.tensor(t, device=dev), e.to(dev), class_labels=c.to(dev), my_pose_cond=p.to(dev),
            return_dict=False)[0]
    backend.sync()
    ref = unet_forward(sd, cfg, s, torch.tensor(t), e, c, p)
    return m, out, ref, (s, e, c, p)


def test_unet_tiny(backend):
    cfg = UNetConfig.tiny()
    B, h, w, L = (2, 8, 8, 5) if backend.is_emu else (4, 16, 24, 10)
    m, out, ref, (s, e, c, p) = _run(backend, cfg, B, h, w, L)
    _check(out, ref)
    # second call with the same conditioning tensors exercises the step-invariant caches
    dev = backend.device
    s2 = torch.randn(s.shape, generator=torch.Generator().manual_seed(5))
    ehs_d, cl_d, pose_d = e.to(dev), c.to(dev), p.to(dev)
    o2 = m(s2.to(dev), 500, ehs_d, class_labels=cl_d, my_pose_cond=pose_d).sample.clone()
    backend.sync()
    if not backend.is_emu:   # run-to-run determinism (skipped under the emulator: ~20 s per forward)
        assert torch.equal(o2, m(s2.to(dev), 500, ehs_d, class_labels=cl_d, my_pose_cond=pose_d).sample)
    sd = m.state_dict()
    _check(o2, unet_forward(sd, cfg, s2, 500, e, c, p))


def test_ff_out_fusion_matches_the_two_launches(backend, monkeypatch):
    """``ff.net.2 (+ residual) -> proj_out (+ residual)`` as one two-source GEMM against [Wp W2 | Wp] (pcdms_amd/unet.py FUSE_FF_OUT; the Linears
    composed at stage2_inpaint_unet_2d_condition.py:321-361): the composed weight is the exact product of the two layers (fp64 check), a
    model packed with the fusion launches one GEMM fewer per transformer block, both forms stay inside the forward tolerance against the
    fp32 oracle, and the fused form is no further from it than the two launches (it skips one bf16 rounding of the residual state)."""
    from pcdms_amd import ops, unet as U
    cfg = UNetConfig.tiny()
    B, h, w, L = (2, 8, 8, 5) if backend.is_emu else (4, 16, 24, 10)
    sd = synth_state_dict(cfg, seed=0, random_affine=True)
    s, e, c, p = _inputs(cfg, B, h, w, L)
    dev = backend.device
    outs, launches = {}, {}
    for fused in (True, False):
        monkeypatch.setattr(U, "FUSE_FF_OUT", fused)
        m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
        m.load_state_dict(sd)
        m.to(dev)
        m._pack()
        blocks = [a for a in m._w.values() if isinstance(a, dict) and "ff1" in a]
        assert blocks and all(("ffo" in a) == fused and ("ff2" in a) != fused and ("proj_out" in a) != fused for a in blocks)
        if fused:   # [Wp W2 | Wp], bias Wp b2 + bp -- to bf16 rounding of the packed copy
            a = blocks[0]
            pre = next(k for k, v in m._w.items() if v is a)
            w2, b2 = sd[pre + "transformer_blocks.0.ff.net.2.weight"].double(), sd[pre + "transformer_blocks.0.ff.net.2.bias"].double()
            wp, bp = sd[pre + "proj_out.weight"].double().reshape(a["c"], a["c"]), sd[pre + "proj_out.bias"].double()
            ref_w = torch.cat([wp @ w2, wp], 1)
            got = a["ffo"].w.float().cpu()[: a["c"]].double()
            assert got.shape == ref_w.shape and ((got - ref_w).norm() / ref_w.norm()).item() < 4e-3
            assert torch.allclose(a["ffo"].bias.cpu()[: a["c"]].double(), wp @ b2 + bp, atol=1e-6)
        count = {"n": 0}
        orig_gemm = ops.gemm

        def counting(*args, **kw):
            count["n"] += 1
            return orig_gemm(*args, **kw)
        monkeypatch.setattr(ops, "gemm", counting)
        out = m(s.to(dev), torch.tensor(981, device=dev), e.to(dev), class_labels=c.to(dev), my_pose_cond=p.to(dev), return_dict=False)[0]
        backend.sync()
        monkeypatch.setattr(ops, "gemm", orig_gemm)
        outs[fused], launches[fused] = out.float().cpu(), count["n"]
    n_blocks = len(blocks)
    assert launches[False] - launches[True] == n_blocks, (launches, n_blocks)
    ref = unet_forward(sd, cfg, s, torch.tensor(981), e, c, p)
    rel_f, _ = _check(outs[True], ref)
    rel_u, _ = _check(outs[False], ref)
    assert rel_f <= rel_u * 1.25 + 1e-4, (rel_f, rel_u)
    assert ((outs[True] - outs[False]).norm() / outs[False].norm()).item() < 1e-2


def test_shortcut_fusion_matches_the_two_launches(backend, monkeypatch):
    """``conv2(h) + conv_shortcut(x)`` of every channel-changing ResnetBlock2D as one contraction (pcdms_amd/unet.py FUSE_SHORTCUT; resnet.py's
    forward as composed at stage2_inpaint_unet_2d_condition.py:321-344,407-430): a model packed with it launches one GEMM fewer per such
    block, both forms stay inside the forward tolerance against the fp32 oracle, and they agree with each other to bf16 rounding of the
    residual the fused form no longer rounds."""
    from pcdms_amd import ops, unet as U
    cfg = UNetConfig.tiny()
    B, h, w, L = (2, 8, 8, 5) if backend.is_emu else (4, 16, 24, 10)
    sd = synth_state_dict(cfg, seed=0, random_affine=True)
    s, e, c, p = _inputs(cfg, B, h, w, L)
    dev = backend.device
    outs, launches = {}, {}
    for fused in (True, False):
        monkeypatch.setattr(U, "FUSE_SHORTCUT", fused)
        m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
        m.load_state_dict(sd)
        m.to(dev)
        m._pack()
        blocks = [r for r in m._w.values() if isinstance(r, dict) and "conv1" in r and r["cin"] != r["cout"]]
        assert blocks and all(("conv2s" in r) == fused and ("conv2" in r) != fused and ("short" in r) != fused for r in blocks)
        count = {"n": 0}
        orig_gemm = ops.gemm

        def counting(*args, **kw):
            count["n"] += 1
            return orig_gemm(*args, **kw)
        monkeypatch.setattr(ops, "gemm", counting)
        out = m(s.to(dev), torch.tensor(981, device=dev), e.to(dev), class_labels=c.to(dev), my_pose_cond=p.to(dev), return_dict=False)[0]
        backend.sync()
        monkeypatch.setattr(ops, "gemm", orig_gemm)
        outs[fused], launches[fused] = out.float().cpu(), count["n"]
    assert launches[False] - launches[True] == len(blocks), (launches, len(blocks))
    ref = unet_forward(sd, cfg, s, torch.tensor(981), e, c, p)
    rel_f, _ = _check(outs[True], ref)
    rel_u, _ = _check(outs[False], ref)
    assert rel_f <= rel_u * 1.25 + 1e-4, (rel_f, rel_u)
    assert ((outs[True] - outs[False]).norm() / outs[False].norm()).item() < 1e-2


def test_phase_upsample_matches_the_gather_form(backend, monkeypatch):
    """Upsample2D as its phase decomposition (pcdms_amd/unet.py PHASE_UPSAMPLE: one 3x3 launch on the low-res tensor with N = 4 C over four taps
    per output-channel group + a pixel shuffle) against the gather form (nine taps on the upsampled grid): both inside the forward
    tolerance against the fp32 oracle, and within bf16 rounding of the summed weights of each other."""
    from pcdms_amd import unet as U
    cfg = UNetConfig.tiny()
    B, h, w, L = (2, 8, 8, 5) if backend.is_emu else (4, 16, 24, 10)
    sd = synth_state_dict(cfg, seed=0, random_affine=True)
    s, e, c, p = _inputs(cfg, B, h, w, L)
    dev = backend.device
    outs = {}
    for phase in (True, False):
        monkeypatch.setattr(U, "PHASE_UPSAMPLE", phase)
        m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
        m.load_state_dict(sd)
        m.to(dev)
        m._pack()
        assert any(k.endswith("conv4.") for k in m._w if isinstance(k, str)) == phase
        out = m(s.to(dev), torch.tensor(981, device=dev), e.to(dev), class_labels=c.to(dev), my_pose_cond=p.to(dev), return_dict=False)[0]
        backend.sync()
        outs[phase] = out.float().cpu()
    ref = unet_forward(sd, cfg, s, torch.tensor(981), e, c, p)
    rel_p, _ = _check(outs[True], ref)
    rel_g, _ = _check(outs[False], ref)
    assert rel_p <= rel_g * 1.25 + 1e-4, (rel_p, rel_g)
    assert ((outs[True] - outs[False]).norm() / outs[False].norm()).item() < 1e-2


@pytest.mark.gpu
def test_unet_latent_not_divisible_by_8(gpu_backend):
    """Latent 20x11 (like the stage-3 latent 64x44 of a 352-wide image): the stride-2 convs round up (11 -> 6 -> 3 -> 2) and
    the up path interpolates to each skip's size (ref :625-633 forward_upsample_size) instead of x2."""
    cfg = UNetConfig.tiny()
    m, out, ref, _ = _run(gpu_backend, cfg, 2, 20, 11, 7)
    _check(out, ref)


@pytest.mark.gpu
def test_unet_full_size_config1(gpu_backend):
    """Full 868.9 M-parameter topology at config 1's latent 32x64 (256x256 pair), UNet batch 2."""
    cfg = UNetConfig()
    m, out, ref, _ = _run(gpu_backend, cfg, 2, 32, 64, 258, random_affine=False)
    rel, mx = _check(out, ref)
    print(f"full-size forward: rel-L2 {rel:.4f} max/scale {mx:.4f}")


def test_unet_fp8_attention(backend):
    """SURVEY.md §8f N4 / BASELINE.json configs[4]: every attention with e4m3 operands on the MX-scaled fp8 MFMA.  Stated tolerance of
    this mode for one forward against the fp32 oracle: rel-L2 <= 6e-2 (bf16 attention: 2.5e-2), and it must stay close to the bf16
    path (rel-L2 <= 5e-2): attention is 21 % of the FLOPs but every block's output passes through it."""
    cfg = UNetConfig.tiny()
    sd = synth_state_dict(cfg, seed=7, random_affine=True)
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(sd)
    m.to(backend.device)
    B, h, w, L = (2, 8, 8, 4) if backend.is_emu else (4, 16, 24, 9)
    sample, ehs, cl, pose = _inputs(cfg, B, h, w, L, seed=3)
    dev = backend.device
    t = torch.tensor(400)
    ref = unet_forward(sd, cfg, sample, t, ehs, cl, pose)
    args = dict(encoder_hidden_states=ehs.to(dev), class_labels=cl.to(dev), my_pose_cond=pose.to(dev))
    out16 = m(sample.to(dev), t, **args).sample.float().cpu()
    m.set_attention_precision("fp8")
    out8 = m(sample.to(dev), t, **args).sample.float().cpu()
    backend.sync()
    r8 = ((out8 - ref).norm() / ref.norm()).item()
    r816 = ((out8 - out16).norm() / out16.norm()).item()
    assert torch.isfinite(out8).all() and r8 <= 6e-2 and r816 <= 5e-2, (r8, r816)
    assert r816 > 0   # (the fp8 path really ran)
    m.set_attention_precision("bf16")
    assert torch.equal(m(sample.to(dev), t, **args).sample.float().cpu(), out16)
    with pytest.raises(ValueError):
        m.set_attention_precision("fp4")
