"""fp32 CPU restatement of the two small per-pair conditioning nets of the stage-2 driver (oracle; test infrastructure).

* ``pose_embedding`` -- diffusers 0.24.0 ``ControlNetConditioningEmbedding(320, 3, (16, 32, 96, 256))``
  (instantiated at /root/reference/stage2_batchtest_inpaint_model.py:101, called :173-174):
  conv_in 3->16, then per level (conv c->c, conv c->c_next stride 2), SiLU after every conv except the last,
  conv_out 256->320 (zero-initialised at construction, trained afterwards).  PARITY UNPINNED ([D-0.24] block).
* ``image_proj_p`` -- ``ImageProjModel_p`` (/root/reference/stage2_batchtest_inpaint_model.py:48-64): Linear -> GELU(erf) ->
  Dropout(0) -> LayerNorm -> Linear -> Dropout(0); state-dict keys ``net.{0,3,4}.*``.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterator, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def pose_param_shapes(out_ch: int = 320, cond_ch: int = 3, boc: Sequence[int] = (16, 32, 96, 256)) -> Iterator[Tuple[str, Tuple[int, ...]]]:
    yield "conv_in.weight", (boc[0], cond_ch, 3, 3)
    yield "conv_in.bias", (boc[0],)
    for i in range(len(boc) - 1):
        yield f"blocks.{2 * i}.weight", (boc[i], boc[i], 3, 3)
        yield f"blocks.{2 * i}.bias", (boc[i],)
        yield f"blocks.{2 * i + 1}.weight", (boc[i + 1], boc[i], 3, 3)
        yield f"blocks.{2 * i + 1}.bias", (boc[i + 1],)
    yield "conv_out.weight", (out_ch, boc[-1], 3, 3)
    yield "conv_out.bias", (out_ch,)


def image_proj_param_shapes(in_dim: int = 1536, hidden_dim: int = 768, out_dim: int = 1024) -> Iterator[Tuple[str, Tuple[int, ...]]]:
    yield "net.0.weight", (hidden_dim, in_dim)
    yield "net.0.bias", (hidden_dim,)
    yield "net.3.weight", (hidden_dim,)
    yield "net.3.bias", (hidden_dim,)
    yield "net.4.weight", (out_dim, hidden_dim)
    yield "net.4.bias", (out_dim,)


def synth(shapes: Iterator[Tuple[str, Tuple[int, ...]]], seed: int = 0, gain: float = 1.7) -> Dict[str, Tensor]:
    """Seeded synthetic weights: U(+-gain/sqrt(fan_in)) matrices (gain > 1 keeps the SiLU chain O(1)), random LN affine."""
    shapes = dict(shapes)
    sd = {}
    for key, shape in shapes.items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
        wshape = shapes[key[: key.rfind(".") + 1] + "weight"]
        if len(wshape) == 1:
            sd[key] = torch.rand(shape, generator=g) * 0.5 + 0.75 if key.endswith("weight") else (torch.rand(shape, generator=g) - 0.5) * 0.4
        else:
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * gain / math.sqrt(math.prod(wshape[1:]))
    return sd


def pose_embedding(sd: Dict[str, Tensor], conditioning: Tensor) -> Tensor:
    """[B,3,H,W] -> [B,320,H/8,W/8]."""
    h = F.silu(F.conv2d(conditioning, sd["conv_in.weight"], sd["conv_in.bias"], padding=1))
    nb = sum(1 for k in sd if k.startswith("blocks.") and k.endswith(".weight"))
    for i in range(nb):
        h = F.silu(F.conv2d(h, sd[f"blocks.{i}.weight"], sd[f"blocks.{i}.bias"], padding=1, stride=2 if i % 2 else 1))
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def image_proj_p(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """[B,L,in_dim] -> [B,L,out_dim]."""
    h = F.gelu(F.linear(x, sd["net.0.weight"], sd["net.0.bias"]))
    h = F.layer_norm(h, (h.shape[-1],), sd["net.3.weight"], sd["net.3.bias"], 1e-5)
    return F.linear(h, sd["net.4.weight"], sd["net.4.bias"])


def image_projection(sd: Dict[str, Tensor], id_embeds: Tensor, num_tokens: int) -> Tensor:
    """``ImageProjection`` (ref src/pipelines/PCDMs_pipeline.py:154-173): [B,E] -> [B,num_tokens,D]."""
    x = F.linear(F.gelu(F.linear(id_embeds, sd["proj.0.weight"], sd["proj.0.bias"])), sd["proj.2.weight"], sd["proj.2.bias"])
    D = sd["norm.weight"].shape[0]
    return F.layer_norm(x.reshape(-1, num_tokens, D), (D,), sd["norm.weight"], sd["norm.bias"], 1e-5)
