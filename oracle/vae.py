"""fp32 CPU restatement of diffusers 0.24.0 ``AutoencoderKL`` (oracle; test infrastructure).

Call sites in the reference: src/pipelines/stage2_inpaint_pipeline.py:443-444 (``vae.encode(x).latent_dist.sample()``
times ``scaling_factor``) and :528-532 (``vae.decode(latents / scaling_factor)`` + ``VaeImageProcessor.postprocess``).
PARITY UNPINNED like the rest of the [D-0.24] blocks (oracle/__init__.py): restated from the published
architecture (SD-2.1 ``vae/config.json``: block_out_channels [128,256,512,512], layers_per_block 2, 32 groups,
latent_channels 4, scaling_factor 0.18215).  Functional over a flat state dict with diffusers key names.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass
from typing import Dict, Iterator, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215

    @staticmethod
    def tiny(**kw) -> "VAEConfig":
        base = dict(block_out_channels=(64, 64, 128, 128))
        base.update(kw)
        return VAEConfig(**base)


def _res(p, cin, cout):
    yield p + "norm1.weight", (cin,)
    yield p + "norm1.bias", (cin,)
    yield p + "conv1.weight", (cout, cin, 3, 3)
    yield p + "conv1.bias", (cout,)
    yield p + "norm2.weight", (cout,)
    yield p + "norm2.bias", (cout,)
    yield p + "conv2.weight", (cout, cout, 3, 3)
    yield p + "conv2.bias", (cout,)
    if cin != cout:
        yield p + "conv_shortcut.weight", (cout, cin, 1, 1)
        yield p + "conv_shortcut.bias", (cout,)


def _mid(p, c):
    yield from _res(p + "resnets.0.", c, c)
    a = p + "attentions.0."
    yield a + "group_norm.weight", (c,)
    yield a + "group_norm.bias", (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        yield a + n + ".weight", (c, c)
        yield a + n + ".bias", (c,)
    yield from _res(p + "resnets.1.", c, c)


def param_shapes(cfg: VAEConfig) -> Iterator[Tuple[str, Tuple[int, ...]]]:
    boc, L, zc = cfg.block_out_channels, cfg.layers_per_block, cfg.latent_channels
    yield "encoder.conv_in.weight", (boc[0], cfg.in_channels, 3, 3)
    yield "encoder.conv_in.bias", (boc[0],)
    out = boc[0]
    for i in range(len(boc)):
        cin, out = out, boc[i]
        for j in range(L):
            yield from _res(f"encoder.down_blocks.{i}.resnets.{j}.", cin if j == 0 else out, out)
        if i != len(boc) - 1:
            yield f"encoder.down_blocks.{i}.downsamplers.0.conv.weight", (out, out, 3, 3)
            yield f"encoder.down_blocks.{i}.downsamplers.0.conv.bias", (out,)
    yield from _mid("encoder.mid_block.", boc[-1])
    yield "encoder.conv_norm_out.weight", (boc[-1],)
    yield "encoder.conv_norm_out.bias", (boc[-1],)
    yield "encoder.conv_out.weight", (2 * zc, boc[-1], 3, 3)
    yield "encoder.conv_out.bias", (2 * zc,)
    yield "quant_conv.weight", (2 * zc, 2 * zc, 1, 1)
    yield "quant_conv.bias", (2 * zc,)
    yield "post_quant_conv.weight", (zc, zc, 1, 1)
    yield "post_quant_conv.bias", (zc,)
    rev = list(reversed(boc))
    yield "decoder.conv_in.weight", (rev[0], zc, 3, 3)
    yield "decoder.conv_in.bias", (rev[0],)
    yield from _mid("decoder.mid_block.", rev[0])
    out = rev[0]
    for i in range(len(boc)):
        prev, out = out, rev[i]
        for j in range(L + 1):
            yield from _res(f"decoder.up_blocks.{i}.resnets.{j}.", prev if j == 0 else out, out)
        if i != len(boc) - 1:
            yield f"decoder.up_blocks.{i}.upsamplers.0.conv.weight", (out, out, 3, 3)
            yield f"decoder.up_blocks.{i}.upsamplers.0.conv.bias", (out,)
    yield "decoder.conv_norm_out.weight", (boc[0],)
    yield "decoder.conv_norm_out.bias", (boc[0],)
    yield "decoder.conv_out.weight", (cfg.out_channels, boc[0], 3, 3)
    yield "decoder.conv_out.bias", (cfg.out_channels,)


def synth_state_dict(cfg: VAEConfig, seed: int = 0) -> Dict[str, Tensor]:
    """Seeded synthetic weights: U(+-1/sqrt(fan_in)) (x0.5 on each resnet conv2 / to_out), random GN affine."""
    shapes = dict(param_shapes(cfg))
    sd = {}
    for key, shape in shapes.items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
        wshape = shapes[key[: key.rfind(".") + 1] + "weight"]
        if len(wshape) == 1:
            sd[key] = torch.rand(shape, generator=g) * 0.5 + 0.75 if key.endswith("weight") else (torch.rand(shape, generator=g) - 0.5) * 0.4
        else:
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(math.prod(wshape[1:]))
            if key.endswith("weight") and (".conv2." in key or ".to_out.0." in key):
                t = t * 0.5
            sd[key] = t
    return sd


def _resnet(sd, p, x, G):
    h = F.conv2d(F.silu(F.group_norm(x, G, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)),
                 sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.conv2d(F.silu(F.group_norm(h, G, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)),
                 sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def _attn(sd, p, x, G):
    """diffusers ``Attention(heads=1, dim_head=C, residual_connection=True, norm_num_groups=G, bias=True)``."""
    B, C, H, W = x.shape
    h = F.group_norm(x, G, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], 1e-6)
    t = h.view(B, C, H * W).transpose(1, 2)
    q = F.linear(t, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(t, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(t, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), dim=-1) @ v
    o = F.linear(a, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def _mid_block(sd, p, x, G):
    x = _resnet(sd, p + "resnets.0.", x, G)
    x = _attn(sd, p + "attentions.0.", x, G)
    return _resnet(sd, p + "resnets.1.", x, G)


def encode_moments(sd, cfg: VAEConfig, x: Tensor) -> Tensor:
    """``quant_conv(encoder(x))`` -> [B, 2*latent, h/8, w/8] (mean | logvar)."""
    G, boc, L = cfg.norm_num_groups, cfg.block_out_channels, cfg.layers_per_block
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(len(boc)):
        for j in range(L):
            h = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}.", h, G)
        if i != len(boc) - 1:   # Downsample2D(padding=0): pad right/bottom by one, stride-2 conv
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    h = _mid_block(sd, "encoder.mid_block.", h, G)
    h = F.silu(F.group_norm(h, G, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], 1e-6))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def sample_latents(moments: Tensor, noise: Tensor) -> Tensor:
    """``DiagonalGaussianDistribution.sample``: mean + exp(0.5*clamp(logvar,-30,20)) * noise."""
    mean, logvar = moments.chunk(2, dim=1)
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise


def decode(sd, cfg: VAEConfig, z: Tensor) -> Tensor:
    G, boc, L = cfg.norm_num_groups, cfg.block_out_channels, cfg.layers_per_block
    z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _mid_block(sd, "decoder.mid_block.", h, G)
    for i in range(len(boc)):
        for j in range(L + 1):
            h = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}.", h, G)
        if i != len(boc) - 1:
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"),
                         sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = F.silu(F.group_norm(h, G, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def postprocess_uint8(image: Tensor) -> Tensor:
    """``VaeImageProcessor.postprocess(..., "pil")`` up to the PIL wrapper:
    (img/2+0.5).clamp(0,1) -> NHWC -> (x*255).round() -> uint8."""
    x = (image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1)
    return (x * 255).round().to(torch.uint8)
