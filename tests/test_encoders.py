"""DINOv2 image encoder: pcdms_amd.Dinov2Model (HIP) vs ``transformers.Dinov2Model`` (fp32, CPU).

The oracle here is the real third-party implementation the reference calls
(/root/reference/stage2_batchtest_inpaint_model.py:96,165-166), instantiated from a config with seeded random weights
(no checkpoint offline).  Stated tolerance: bf16 tokens through L pre-norm blocks vs fp32: rel-L2 <= 3e-2 on
``last_hidden_state``.
"""
from __future__ import annotations

import pytest
import torch

from pcdms_amd import Dinov2Model

transformers = pytest.importorskip("transformers")


def _hf(cfg_kw, seed=0):
    from transformers import Dinov2Config
    from transformers import Dinov2Model as HFDinov2
    torch.manual_seed(seed)
    cfg = Dinov2Config(**cfg_kw)
    m = HFDinov2(cfg).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():   # HF init leaves LayerScale at 1 and biases at 0: randomise so that every term is exercised
        for k, p in m.named_parameters():
            if k.endswith("lambda1"):
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.25)
            elif k.endswith(".bias") and "norm" not in k:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif "norm" in k and k.endswith(".bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif "position_embeddings" in k or "cls_token" in k:
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    return cfg, m


def _rel(a, b):
    a, b = a.float().cpu(), b.float()
    return ((a - b).norm() / b.norm()).item()


TINY = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, mlp_ratio=4, use_swiglu_ffn=True, image_size=518,
            patch_size=14)


def test_dinov2_param_contract():
    cfg, hf = _hf(TINY)
    m = Dinov2Model(cfg)
    assert m.expected_shapes() == {k: tuple(v.shape) for k, v in hf.state_dict().items()}
    giant = Dinov2Model()   # defaults = DINOv2-giant
    n = sum(torch.Size(s).numel() for s in giant.expected_shapes().values())
    assert giant.ffn_dim == 4096 and 1.13e9 < n < 1.14e9
    with pytest.raises(NotImplementedError):
        Dinov2Model(hidden_size=96, num_attention_heads=2)


@pytest.mark.parametrize("swiglu", [True, False])
def test_dinov2_tiny_vs_transformers(backend, swiglu):
    cfg, hf = _hf(dict(TINY, use_swiglu_ffn=swiglu), seed=3)
    m = Dinov2Model(cfg)
    m.load_state_dict(hf.state_dict())
    m.to(backend.device)
    B, S = (1, 28) if backend.is_emu else (2, 224)
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(5))
    out = m(x.to(backend.device))
    backend.sync()
    with torch.no_grad():
        ref = hf(x)
    T = 1 + (S // 14) ** 2
    assert out.last_hidden_state.shape == ref.last_hidden_state.shape == (B, T, 128)
    assert _rel(out.last_hidden_state, ref.last_hidden_state) <= 3e-2, _rel(out.last_hidden_state, ref.last_hidden_state)
    assert _rel(out.pooler_output, ref.pooler_output) <= 3e-2
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 30, 28))


@pytest.mark.gpu
def test_dinov2_giant_shapes(gpu_backend):
    """The reference's encoder size (hidden 1536, 24 heads, SwiGLU 4096) at 224x224 -> [1, 257, 1536]
    (stage2_batchtest_inpaint_model.py:165-166); 8 of the 40 layers to bound the CPU oracle's memory / time."""
    cfg, hf = _hf(dict(hidden_size=1536, num_hidden_layers=8, num_attention_heads=24, mlp_ratio=4, use_swiglu_ffn=True,
                       image_size=518, patch_size=14), seed=7)
    m = Dinov2Model(cfg)
    m.load_state_dict(hf.state_dict())
    m.to(gpu_backend.device)
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(9))
    out = m(x.to(gpu_backend.device)).last_hidden_state
    with torch.no_grad():
        ref = hf(x).last_hidden_state
    assert out.shape == (1, 257, 1536) and _rel(out, ref) <= 3e-2, _rel(out, ref)


# ---------------------------------------------------------------------------------------------------------- CLIP vision tower
def _hf_clip(cfg_kw, seed=0):
    from transformers import CLIPVisionConfig
    from transformers import CLIPVisionModelWithProjection as HFClip
    torch.manual_seed(seed)
    cfg = CLIPVisionConfig(**cfg_kw)
    m = HFClip(cfg).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith(".bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif p.dim() == 2 and "position" not in k:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / p.shape[1] ** 0.5)   # HF's init std 0.02 would leave the blocks near-identity
            elif "class_embedding" in k or "position" in k:
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    return cfg, m


@pytest.mark.parametrize("head_dim", [80, 64])
def test_clip_vision_vs_transformers(backend, head_dim):
    """head_dim 80 (ViT-H/14: per-head GEMM attention) and 64 (flash kernel) against transformers' CLIP vision tower."""
    from pcdms_amd import CLIPVisionModelWithProjection
    S = 28 if backend.is_emu else 224
    cfg, hf = _hf_clip(dict(hidden_size=4 * head_dim, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4, image_size=S,
                            patch_size=14, hidden_act="gelu", projection_dim=64), seed=11)
    m = CLIPVisionModelWithProjection(cfg)
    assert m.expected_shapes() == {k: tuple(v.shape) for k, v in hf.state_dict().items() if not k.endswith("position_ids")}
    m.load_state_dict(hf.state_dict())
    m.to(backend.device)
    B = 1 if backend.is_emu else 2
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(12))
    out = m(x.to(backend.device))
    backend.sync()
    with torch.no_grad():
        ref = hf(x)
    assert out.image_embeds.shape == ref.image_embeds.shape == (B, 64) and out.image_embeds.dtype == torch.float32
    assert _rel(out.last_hidden_state, ref.last_hidden_state) <= 3e-2, _rel(out.last_hidden_state, ref.last_hidden_state)
    assert _rel(out.image_embeds, ref.image_embeds) <= 3e-2, _rel(out.image_embeds, ref.image_embeds)
    assert torch.equal(out["image_embeds"], out.image_embeds)   # the drivers index the output both ways
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 14, 14))


def test_clip_vision_param_contract():
    from pcdms_amd import CLIPVisionModelWithProjection
    m = CLIPVisionModelWithProjection()   # defaults = OpenCLIP ViT-H/14 vision tower + projection
    n = sum(torch.Size(s).numel() for s in m.expected_shapes().values())
    assert 6.30e8 < n < 6.34e8   # 632 M
    with pytest.raises(NotImplementedError):
        CLIPVisionModelWithProjection(hidden_act="quick_gelu")
