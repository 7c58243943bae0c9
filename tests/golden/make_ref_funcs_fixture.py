#!/usr/bin/env python3
"""Generate tests/golden/ref_funcs.npz by EXECUTING the pure-torch arithmetic the reference itself holds.

    python tests/golden/make_ref_funcs_fixture.py        (needs /root/reference; dev container only)

Unlike the ``ref_wiring_*`` fixtures (reference wiring on top of oracle blocks), everything recorded here is computed
by reference code alone -- torch in, torch out, no oracle and no diffusers arithmetic underneath:

  * ``rescale_noise_cfg``                 src/pipelines/stage2_inpaint_pipeline.py:52-63 (and its twin PCDMs_pipeline.py:176-187)
  * ``FusedAttnProcessor2_0.__call__``    src/pipelines/PCDMs_pipeline.py:59-153 -- the reference's own copy of the Attention arithmetic
                                          of SURVEY.md Appendix A-7 (fused qkv / kv projections, head split, SDPA, to_out), run on a
                                          minimal ``attn`` namespace of plain ``torch.nn.Linear`` layers
  * ``ImageProjModel_p``                  stage2_batchtest_inpaint_model.py:48-64
  * ``ImageProjection``                   src/pipelines/PCDMs_pipeline.py:154-173
  * ``retrieve_timesteps``                src/pipelines/PCDMs_pipeline.py:190-231 (both branches + the error path)
  * ``split_list_into_chunks``            stage2_batchtest_inpaint_model.py:25-31

The two modules import names from ``diffusers`` / ``torchvision`` / ``skimage`` that are not installed; they are only *names* at
import time (base classes, type hints), so the generator registers import-permissive placeholder modules for them.  None of the
functions executed below touches those placeholders (asserted: the placeholders raise when called).

The fixture holds inputs, weights and expected outputs only.
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")


class _Placeholder:
    """Stands for any third-party NAME the reference modules import; using it in arithmetic is an error."""

    def __init__(self, *a, **k):
        raise RuntimeError("placeholder for an un-installed third-party class was instantiated")

    def __init_subclass__(cls, **k):   # reference classes may list it as a base (never instantiated here)
        pass


def _permissive_module(name: str) -> types.ModuleType:
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    if "__getattr__" not in m.__dict__:
        def _ga(attr, _n=name):
            if attr.startswith("__"):
                raise AttributeError(attr)
            if attr == "USE_PEFT_BACKEND":   # diffusers' "peft installed" flag: the projections are then plain torch.nn.Linear
                return True                  # (called without the LoRA `scale` argument, PCDMs_pipeline.py:110)
            return type(attr, (_Placeholder,), {})
        m.__getattr__ = _ga
    return m


def import_reference():
    from tests.golden import diffusers_stub
    diffusers_stub.install()
    import transformers  # noqa: F401  (real package; must probe for torchvision BEFORE the placeholder below exists)
    from transformers import CLIPImageProcessor, CLIPVisionModelWithProjection, Dinov2Model  # noqa: F401
    for name in ("diffusers", "diffusers.configuration_utils", "diffusers.image_processor", "diffusers.loaders", "diffusers.models",
                 "diffusers.models.attention_processor", "diffusers.models.lora", "diffusers.models.controlnet", "diffusers.schedulers",
                 "diffusers.utils", "diffusers.utils.torch_utils", "diffusers.pipelines", "diffusers.pipelines.pipeline_utils",
                 "diffusers.pipelines.stable_diffusion", "diffusers.pipelines.stable_diffusion.pipeline_output",
                 "diffusers.pipelines.stable_diffusion.safety_checker", "torchvision", "torchvision.transforms", "skimage",
                 "skimage.metrics", "cv2"):
        _permissive_module(name)
    sys.path.insert(0, str(REF))
    import src.pipelines.PCDMs_pipeline as pcdms
    import src.pipelines.stage2_inpaint_pipeline as s2p
    import stage2_batchtest_inpaint_model as drv
    return pcdms, s2p, drv


def _seeded_linear(lin: nn.Linear, g: torch.Generator, gain: float = 1.0):
    with torch.no_grad():
        bound = gain / (lin.in_features ** 0.5)
        lin.weight.copy_((torch.rand(lin.weight.shape, generator=g) * 2 - 1) * bound)
        if lin.bias is not None:
            lin.bias.copy_((torch.rand(lin.bias.shape, generator=g) * 2 - 1) * bound)


def main():
    pcdms, s2p, drv = import_reference()
    out = dict(torch_version=torch.__version__)
    g = torch.Generator().manual_seed(2024)

    # ---- rescale_noise_cfg (both copies must agree with each other)
    cfg_eps = torch.randn(3, 4, 8, 16, generator=g) * 1.3 + 0.05
    text_eps = torch.randn(3, 4, 8, 16, generator=g) * 0.9
    for gr in (0.0, 0.7, 1.0):
        a = s2p.rescale_noise_cfg(cfg_eps, text_eps, guidance_rescale=gr)
        b = pcdms.rescale_noise_cfg(cfg_eps, text_eps, guidance_rescale=gr)
        assert torch.equal(a, b)
        out[f"rescale_out_{int(gr * 10):02d}"] = a.numpy()
    out["rescale_cfg"], out["rescale_text"] = cfg_eps.numpy(), text_eps.numpy()

    # ---- FusedAttnProcessor2_0.__call__ on plain torch layers: self-attention (fused qkv), cross-attention (fused kv),
    #      and the 4-D input form (NCHW -> tokens -> NCHW)
    proc = pcdms.FusedAttnProcessor2_0()
    C, heads, ctx_dim = 128, 2, 64           # head_dim 64: the size the HIP attention kernel implements
    B, N, L = 2, 48, 7

    def make_attn(cross: bool):
        a = types.SimpleNamespace(spatial_norm=None, group_norm=None, norm_cross=False, residual_connection=False,
                                  rescale_output_factor=1.0, heads=heads)
        a.to_out = nn.ModuleList([nn.Linear(C, C), nn.Identity()])
        _seeded_linear(a.to_out[0], g)
        if cross:
            a.to_q, a.to_kv = nn.Linear(C, C, bias=False), nn.Linear(ctx_dim, 2 * C, bias=False)
            _seeded_linear(a.to_q, g, 2.0); _seeded_linear(a.to_kv, g, 2.0)
        else:
            a.to_qkv = nn.Linear(C, 3 * C, bias=False)
            _seeded_linear(a.to_qkv, g, 2.0)
        return a

    with torch.no_grad():
        a_self = make_attn(False)
        x = torch.randn(B, N, C, generator=g)
        out["attn_self_x"] = x.numpy()
        out["attn_self_wqkv"] = a_self.to_qkv.weight.numpy()
        out["attn_self_wo"], out["attn_self_bo"] = a_self.to_out[0].weight.numpy(), a_self.to_out[0].bias.numpy()
        out["attn_self_out"] = proc(a_self, x).numpy()
        x4 = torch.randn(B, C, 6, 8, generator=g)
        out["attn_self_x4"] = x4.numpy()
        out["attn_self_out4"] = proc(a_self, x4).numpy()

        a_cross = make_attn(True)
        xq = torch.randn(B, N, C, generator=g)
        ctx = torch.randn(B, L, ctx_dim, generator=g)
        out["attn_cross_x"], out["attn_cross_ctx"] = xq.numpy(), ctx.numpy()
        out["attn_cross_wq"], out["attn_cross_wkv"] = a_cross.to_q.weight.numpy(), a_cross.to_kv.weight.numpy()
        out["attn_cross_wo"], out["attn_cross_bo"] = a_cross.to_out[0].weight.numpy(), a_cross.to_out[0].bias.numpy()
        out["attn_cross_out"] = proc(a_cross, xq, encoder_hidden_states=ctx).numpy()
    out["attn_heads"] = heads

    # ---- ImageProjModel_p (driver) and ImageProjection (PCDMs_pipeline)
    with torch.no_grad():
        m = drv.ImageProjModel_p(in_dim=128, hidden_dim=64, out_dim=64).eval()
        for lin in (m.net[0], m.net[4]):
            _seeded_linear(lin, g)
        m.net[3].weight.copy_(1 + 0.2 * torch.randn(64, generator=g)); m.net[3].bias.copy_(0.1 * torch.randn(64, generator=g))
        xi = torch.randn(1, 9, 128, generator=g)
        out["ipm_x"], out["ipm_y"] = xi.numpy(), m(xi).numpy()
        for k, v in m.state_dict().items():
            out["ipm_sd." + k] = v.numpy()

        p = pcdms.ImageProjection(cross_attention_dim=64, clip_embeddings_dim=64, num_tokens=4).eval()
        for lin in (p.proj[0], p.proj[2]):
            _seeded_linear(lin, g)
        p.norm.weight.copy_(1 + 0.2 * torch.randn(64, generator=g)); p.norm.bias.copy_(0.1 * torch.randn(64, generator=g))
        xe = torch.randn(3, 64, generator=g)
        out["iproj_x"], out["iproj_y"] = xe.numpy(), p(xe).numpy()
        for k, v in p.state_dict().items():
            out["iproj_sd." + k] = v.numpy()

    # ---- retrieve_timesteps: delegates to scheduler.set_timesteps; custom timesteps need a `timesteps` parameter
    class SchedPlain:
        def set_timesteps(self, num_inference_steps, device=None):
            self.timesteps = torch.arange(num_inference_steps - 1, -1, -1) * 7 + 1
            self.called = ("n", num_inference_steps, device)

    class SchedCustom(SchedPlain):
        def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
            self.timesteps = torch.tensor(timesteps) if timesteps is not None else torch.arange(num_inference_steps - 1, -1, -1)
            self.called = ("t" if timesteps is not None else "n", num_inference_steps, device)

    s = SchedPlain()
    ts, n = pcdms.retrieve_timesteps(s, 5, "cpu")
    out["rt_plain_ts"], out["rt_plain_n"] = ts.numpy(), n
    assert s.called == ("n", 5, "cpu")
    s = SchedCustom()
    ts, n = pcdms.retrieve_timesteps(s, None, None, timesteps=[900, 500, 100])
    out["rt_custom_ts"], out["rt_custom_n"] = ts.numpy(), n
    try:
        pcdms.retrieve_timesteps(SchedPlain(), None, None, timesteps=[3, 2, 1])
        raise AssertionError("expected ValueError")
    except ValueError as e:
        out["rt_error_prefix"] = str(e)[:28]

    # ---- split_list_into_chunks (remainder folded into the LAST chunk)
    for total, parts in ((10, 3), (8, 8), (7, 2), (5, 1), (64, 8)):
        chunks = drv.split_list_into_chunks(list(range(total)), parts)
        out[f"chunks_{total}_{parts}"] = np.array([len(c) for c in chunks])
        assert sum(chunks, []) == list(range(total))

    np.savez_compressed(HERE / "ref_funcs.npz", **out)
    print("wrote", HERE / "ref_funcs.npz", "with", len(out), "entries")


if __name__ == "__main__":
    main()
