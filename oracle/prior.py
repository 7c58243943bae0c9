"""fp32 CPU restatement of the stage-1 prior (oracle; test infrastructure).

``prior_forward`` follows /root/reference/src/models/stage1_prior_transformer.py:197-297 (token order :264-274,
positional add :277, ``attention_mask=None`` at the only call site src/pipelines/stage1_prior_pipeline.py:458-465, so
no causal mask is applied); the pose encoders are the ``MLP`` of :18-36; each transformer block is diffusers 0.24.0
``BasicTransformerBlock(inner, heads, head_dim, activation_fn="gelu", attention_bias=True)`` without cross-attention:
``x += attn1(norm1(x)); x += ff(norm3(x))`` with FF = Linear -> GELU(erf) -> Linear.  ``stage1_sample`` follows the
pipeline loop :439-485 and ``post_process_latents`` (:299-301).  PARITY UNPINNED for the [D-0.24] blocks; the
reference's own wiring is pinned by tests/golden/ref_wiring_prior.npz.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .unet import timestep_embedding

Tensor = torch.Tensor
CLIP_MEAN, CLIP_STD = -0.016, 0.415      # stage1_prior_transformer.py:132-133


@dataclass
class PriorConfig:
    num_attention_heads: int = 32
    attention_head_dim: int = 64
    num_layers: int = 20
    embedding_dim: int = 1024        # stage1_batchtest_prior_model.py:56 (embedding_dim=1024, num_embeddings=2)
    num_embeddings: int = 2
    additional_embeddings: int = 4
    pose_dim: int = 36               # MLP(in_dim=36, hidden_dim=512, out_dim=1024), :97-98
    pose_hidden: int = 512

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @staticmethod
    def tiny(**kw) -> "PriorConfig":
        base = dict(num_attention_heads=2, num_layers=2)
        base.update(kw)
        return PriorConfig(**base)


def param_shapes(cfg: PriorConfig) -> Iterator[Tuple[str, Tuple[int, ...]]]:
    D, E = cfg.inner_dim, cfg.embedding_dim

    def lin(p, o, i):
        yield p + ".weight", (o, i)
        yield p + ".bias", (o,)

    def ln(p, c):
        yield p + ".weight", (c,)
        yield p + ".bias", (c,)
    for pe in ("pose_encoder", "pose_encoder1"):
        yield from lin(pe + ".net.0", cfg.pose_hidden, cfg.pose_dim)
        yield from ln(pe + ".net.3", cfg.pose_hidden)
        yield from lin(pe + ".net.4", E, cfg.pose_hidden)
        yield from ln(pe + ".net.6", E)
    yield from lin("time_embedding.linear_1", D, D)
    yield from lin("time_embedding.linear_2", D, D)
    yield from lin("proj_in", D, E)
    yield from lin("embedding_proj", D, E)
    yield from lin("encoder_hidden_states_proj", D, E)
    yield from lin("encoder_hidden_states_proj1", D, E)
    yield "positional_embedding", (1, cfg.num_embeddings + cfg.additional_embeddings, D)
    yield "prd_embedding", (1, 1, D)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}."
        yield from ln(p + "norm1", D)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            yield from lin(p + "attn1." + n, D, D)
        yield from ln(p + "norm3", D)
        yield from lin(p + "ff.net.0.proj", 4 * D, D)
        yield from lin(p + "ff.net.2", D, 4 * D)
    yield from ln("norm_out", D)
    yield from lin("proj_to_clip_embeddings", E, D)


def param_count(cfg: PriorConfig) -> int:
    return sum(math.prod(s) for _, s in param_shapes(cfg))


def synth_state_dict(cfg: PriorConfig, seed: int = 0) -> Dict[str, Tensor]:
    """Seeded synthetic weights: U(+-1/sqrt(fan_in)) (x0.5 on attn to_out / ff.net.2), random LN affine,
    N(0, 0.02) positional / prd embeddings."""
    shapes = dict(param_shapes(cfg))
    sd = {}
    for key, shape in shapes.items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
        if key in ("positional_embedding", "prd_embedding"):
            sd[key] = torch.randn(shape, generator=g) * 0.3
            continue
        wshape = shapes[key[: key.rfind(".") + 1] + "weight"]
        if len(wshape) == 1:
            sd[key] = torch.rand(shape, generator=g) * 0.5 + 0.75 if key.endswith("weight") else (torch.rand(shape, generator=g) - 0.5) * 0.4
        else:
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(wshape[1])
            if key.endswith("weight") and (".to_out.0." in key or ".ff.net.2." in key):
                t = t * 0.5
            sd[key] = t
    return sd


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def pose_mlp(sd, p, x):
    """``MLP`` (:18-36): Linear -> GELU -> LayerNorm -> Linear -> LayerNorm (dropouts are identity)."""
    h = _ln(sd, p + ".net.3", F.gelu(_lin(sd, p + ".net.0", x)))
    return _ln(sd, p + ".net.6", _lin(sd, p + ".net.4", h))


def transformer_block(sd, p, x, heads):
    B, T, D = x.shape
    h = _ln(sd, p + "norm1", x)
    q, k, v = (_lin(sd, p + "attn1." + n, h).view(B, T, heads, D // heads).transpose(1, 2) for n in ("to_q", "to_k", "to_v"))
    a = torch.softmax(q @ k.transpose(-1, -2) * (D // heads) ** -0.5, dim=-1) @ v
    x = x + _lin(sd, p + "attn1.to_out.0", a.transpose(1, 2).reshape(B, T, D))
    h = _ln(sd, p + "norm3", x)
    return x + _lin(sd, p + "ff.net.2", F.gelu(_lin(sd, p + "ff.net.0.proj", h)))


def prior_forward(sd: Dict[str, Tensor], cfg: PriorConfig, hidden_states: Tensor, timestep, proj_embedding: Tensor,
                  encoder_hidden_states: Tensor, encoder_hidden_states1: Tensor, taps: Optional[dict] = None) -> Tensor:
    """hidden_states [B,1,E] (x_t), proj_embedding [B,1,E] (source-image CLIP embed), poses [B,1,36] ->
    predicted_image_embedding [B,E]."""
    B = hidden_states.shape[0]
    t = torch.as_tensor(timestep).reshape(-1).to(torch.int64)
    t = t * torch.ones(B, dtype=t.dtype)
    temb = timestep_embedding(t, cfg.inner_dim, True, 0.0)
    temb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", temb)))
    pe = _lin(sd, "embedding_proj", proj_embedding)
    e0 = _lin(sd, "encoder_hidden_states_proj", pose_mlp(sd, "pose_encoder", encoder_hidden_states))
    e1 = _lin(sd, "encoder_hidden_states_proj1", pose_mlp(sd, "pose_encoder1", encoder_hidden_states1))
    x = _lin(sd, "proj_in", hidden_states)
    prd = sd["prd_embedding"].expand(B, -1, -1)
    x = torch.cat([e0, e1, pe, temb[:, None, :], x, prd], dim=1) + sd["positional_embedding"]
    if taps is not None:
        taps["tokens"] = x
    for i in range(cfg.num_layers):
        x = transformer_block(sd, f"transformer_blocks.{i}.", x, cfg.num_attention_heads)
        if taps is not None:
            taps[f"block{i}"] = x
    x = _ln(sd, "norm_out", x)[:, -1]
    return _lin(sd, "proj_to_clip_embeddings", x)


def stage1_sample(sd, cfg: PriorConfig, scheduler, *, s_embed: Tensor, s_pose: Tensor, t_pose: Tensor, latents: Tensor,
                  noises: Optional[List[Tensor]] = None, num_inference_steps: int = 20, guidance_scale: float = 0.0,
                  num_images_per_prompt: int = 1) -> Tensor:
    """ref stage1_prior_pipeline.py:421-485 for one pair.  With guidance_scale > 1 the reference concatenates a zero
    negative embedding in front of ``prompt_embeds`` (:341-346) but does not double the poses (they reach the model
    with batch 1 and ``torch.cat`` at stage1_prior_transformer.py:264 fails); the evident intent -- zero image
    embedding, same poses, for the unconditional rows -- is what is implemented.  ``noises[i]`` = variance noise of
    loop step i (the reference draws it from the global RNG, :478)."""
    N = num_images_per_prompt
    cfg_on = guidance_scale > 1.0
    emb = s_embed.repeat(N, 1, 1)
    sp, tp = s_pose.repeat(N, 1, 1), t_pose.repeat(N, 1, 1)
    if cfg_on:
        emb = torch.cat([torch.zeros_like(emb), emb])
        sp, tp = torch.cat([sp, sp]), torch.cat([tp, tp])
    scheduler.set_timesteps(num_inference_steps)
    ts = scheduler.timesteps
    latents = latents * scheduler.init_noise_sigma
    for i, t in enumerate(ts):
        x = torch.cat([latents] * 2) if cfg_on else latents
        pred = prior_forward(sd, cfg, x.unsqueeze(1), t, emb, sp, tp)
        if cfg_on:
            u, c = pred.chunk(2)
            pred = u + guidance_scale * (c - u)
        prev_t = None if i + 1 == len(ts) else ts[i + 1]
        latents = scheduler.step(pred, t, latents, prev_timestep=prev_t, variance_noise=None if noises is None else noises[i])
    return latents * CLIP_STD + CLIP_MEAN
