"""A minimal stand-in for the ``diffusers`` package, TEST / FIXTURE-GENERATION ONLY.

Purpose (SURVEY.md §8c e): ``diffusers==0.24.0`` is not installable here, so the reference's own
``Stage2_InapintUNet2DConditionModel.forward`` and ``Stage2_InpaintDiffusionPipeline.__call__`` cannot
be imported as is.  ``install()`` registers fake ``diffusers.*`` modules whose block classes carry
the diffusers parameter names and delegate their arithmetic to ``oracle.unet`` (fp32).  Running the
REFERENCE's code on top of them pins everything the reference itself contributes -- pose add (:742),
class-embedding add (:708), skip bookkeeping (:746-814), input cat order
(stage2_inpaint_pipeline.py:501), CFG (:511-512), repeat semantics (:449-452) -- into the committed
fixtures of tests/golden/.  It does NOT pin the diffusers block internals (oracle/__init__.py).
Used by make_reference_wiring_fixtures.py in the dev container only; nothing here travels into the
product and the GPU box never needs /root/reference.
"""
from __future__ import annotations

import inspect
import sys
import types
from dataclasses import dataclass, fields
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import prior as OP
from oracle import unet as O
from oracle.schedulers import DDIMOracle, UnCLIPOracle, UniPCOracle


def _sd(mod: nn.Module):
    return {k: v.float() for k, v in mod.state_dict().items()}


# ------------------------------------------------------------------ configuration_utils / utils
class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kw):
        d = dict(getattr(self, "_internal_dict", {}))
        d.update(kw)
        self._internal_dict = FrozenDict(d)


def register_to_config(init):
    sig = inspect.signature(init)

    def wrapper(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        cfg.setdefault("_diffusers_version", "0.24.0")
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **kwargs)
    return wrapper


class BaseOutput:
    def __getitem__(self, i):
        return tuple(getattr(self, f.name) for f in fields(self))[i]


class _Logger:
    def info(self, *a, **k):
        pass

    warning = warn = debug = error = info


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


# ------------------------------------------------------------------ embeddings
class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.a = (num_channels, flip_sin_to_cos, downscale_freq_shift)

    def forward(self, t):
        return O.timestep_embedding(t, *self.a)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x, condition=None):
        return O.timestep_mlp(_sd(self), "", x.float())


# ------------------------------------------------------------------ blocks (parameter names = diffusers')
class _Resnet(nn.Module):
    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.g, self.eps = groups, eps
        self.norm1 = nn.GroupNorm(groups, cin, eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x, temb):
        return O.resnet_block(_sd(self), "", x.float(), temb.float(), self.g, self.eps)


class _Attn(nn.Module):
    def __init__(self, c, ctx):
        super().__init__()
        self.to_q = nn.Linear(c, c, bias=False)
        self.to_k = nn.Linear(ctx, c, bias=False)
        self.to_v = nn.Linear(ctx, c, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])


class _GEGLU(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.proj = nn.Linear(c, 8 * c)


class _FF(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.net = nn.ModuleList([_GEGLU(c), nn.Dropout(0.0), nn.Linear(4 * c, c)])


class _BTB(nn.Module):
    def __init__(self, c, ctx):
        super().__init__()
        self.norm1, self.attn1 = nn.LayerNorm(c), _Attn(c, c)
        self.norm2, self.attn2 = nn.LayerNorm(c), _Attn(c, ctx)
        self.norm3, self.ff = nn.LayerNorm(c), _FF(c)


class _Transformer2D(nn.Module):
    def __init__(self, heads, c, ctx, groups):
        super().__init__()
        self.h, self.g = heads, groups
        self.norm = nn.GroupNorm(groups, c, 1e-6)
        self.proj_in = nn.Linear(c, c)
        self.transformer_blocks = nn.ModuleList([_BTB(c, ctx)])
        self.proj_out = nn.Linear(c, c)

    def forward(self, x, encoder_hidden_states=None, **kw):
        return (O.transformer_2d(_sd(self), "", x.float(), encoder_hidden_states.float(), self.h, self.g),)


class _Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class _Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x, size=None):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps,
                 resnet_groups, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(in_channels if i == 0 else out_channels, out_channels, temb_channels,
                                              resnet_groups, resnet_eps) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([_Down(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb=None, **kw):
        out = ()
        for r in self.resnets:
            hidden_states = r(hidden_states, temb)
            out += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            out += (hidden_states,)
        return hidden_states, out


class CrossAttnDownBlock2D(DownBlock2D):
    has_cross_attention = True

    def __init__(self, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps,
                 resnet_groups, cross_attention_dim, num_attention_heads, **kw):
        super().__init__(num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps, resnet_groups)
        self.attentions = nn.ModuleList([_Transformer2D(num_attention_heads, out_channels, cross_attention_dim,
                                                        resnet_groups) for _ in range(num_layers)])

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, **kw):
        out = ()
        for r, a in zip(self.resnets, self.attentions):
            hidden_states = a(r(hidden_states, temb), encoder_hidden_states=encoder_hidden_states)[0]
            out += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            out += (hidden_states,)
        return hidden_states, out


class UpBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample,
                 resnet_eps, resnet_groups, **kw):
        super().__init__()
        rs = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            rs.append(_Resnet(rin + skip, out_channels, temb_channels, resnet_groups, resnet_eps))
        self.resnets = nn.ModuleList(rs)
        self.upsamplers = nn.ModuleList([_Up(out_channels)]) if add_upsample else None

    def _attn(self, i, x, ehs):
        return x

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None,
                encoder_hidden_states=None, **kw):
        for i, r in enumerate(self.resnets):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = r(torch.cat([hidden_states, res], dim=1), temb)
            hidden_states = self._attn(i, hidden_states, encoder_hidden_states)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states, upsample_size)
        return hidden_states


class CrossAttnUpBlock2D(UpBlock2D):
    has_cross_attention = True

    def __init__(self, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample,
                 resnet_eps, resnet_groups, cross_attention_dim, num_attention_heads, **kw):
        super().__init__(num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample,
                         resnet_eps, resnet_groups)
        self.attentions = nn.ModuleList([_Transformer2D(num_attention_heads, out_channels, cross_attention_dim,
                                                        resnet_groups) for _ in range(num_layers)])

    def _attn(self, i, x, ehs):
        return self.attentions[i](x, encoder_hidden_states=ehs)[0]


class UNetMidBlock2DCrossAttn(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, resnet_eps, resnet_groups, cross_attention_dim,
                 num_attention_heads, **kw):
        super().__init__()
        self.attentions = nn.ModuleList([_Transformer2D(num_attention_heads, in_channels, cross_attention_dim, resnet_groups)])
        self.resnets = nn.ModuleList([_Resnet(in_channels, in_channels, temb_channels, resnet_groups, resnet_eps)
                                      for _ in range(2)])

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, **kw):
        x = self.resnets[0](hidden_states, temb)
        x = self.attentions[0](x, encoder_hidden_states=encoder_hidden_states)[0]
        return self.resnets[1](x, temb)


def get_down_block(down_block_type, **kw):
    return {"DownBlock2D": DownBlock2D, "CrossAttnDownBlock2D": CrossAttnDownBlock2D}[down_block_type](**kw)


def get_up_block(up_block_type, **kw):
    return {"UpBlock2D": UpBlock2D, "CrossAttnUpBlock2D": CrossAttnUpBlock2D}[up_block_type](**kw)


# ------------------------------------------------------------------ stage-1 prior blocks
class _AttnBias(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])


class _GELU(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.proj = nn.Linear(c, 4 * c)


class BasicTransformerBlock(nn.Module):
    """Self-attention-only block as stage1_prior_transformer.py:108-119 constructs it (parameter names = diffusers')."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, activation_fn="geglu", attention_bias=False, **kw):
        super().__init__()
        assert activation_fn == "gelu" and attention_bias and dim == num_attention_heads * attention_head_dim
        self.h = num_attention_heads
        self.norm1, self.attn1 = nn.LayerNorm(dim), _AttnBias(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = nn.Module()
        self.ff.net = nn.ModuleList([_GELU(dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, hidden_states, attention_mask=None, **kw):
        assert attention_mask is None
        return OP.transformer_block(_sd(self), "", hidden_states.float(), self.h)


class UnCLIPSchedulerStub:
    """diffusers-style wrapper over UnCLIPOracle; variance noise comes from ``noises`` (one per step), because the
    reference draws it from the global RNG (stage1_prior_pipeline.py:478)."""
    init_noise_sigma = 1.0

    def __init__(self, noises):
        self.impl, self.noises, self.i = UnCLIPOracle(), noises, 0

    def set_timesteps(self, n, device=None):
        self.impl.set_timesteps(n)
        self.timesteps = self.impl.timesteps
        self.i = 0

    def step(self, model_output, timestep, sample, prev_timestep=None, generator=None, return_dict=True):
        out = self.impl.step(model_output.float(), timestep, sample.float(), prev_timestep=prev_timestep,
                             variance_noise=self.noises[self.i])
        self.i += 1
        return types.SimpleNamespace(prev_sample=out)


# ------------------------------------------------------------------ pipeline-side stubs
class DiffusionPipeline:
    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def _execution_device(self):
        return torch.device("cpu")

    @property
    def device(self):
        return torch.device("cpu")

    def progress_bar(self, iterable=None, total=None):
        class _PB:
            def __enter__(s):
                return s

            def __exit__(s, *a):
                return False

            def update(s, *a):
                pass

            def __iter__(s):
                return iter(iterable)
        return _PB()


class VaeImageProcessor:
    def __init__(self, vae_scale_factor=8, **kw):
        pass

    def postprocess(self, image, output_type="pil", do_denormalize=None):
        return image


class _Dist:
    def __init__(self, x):
        self.x = x

    def sample(self, generator=None):
        return self.x


class FakeVAE:
    """encode() returns a preset tensor (the fixture's masked_latents / scaling_factor); decode() is identity."""

    def __init__(self, preset):
        self.preset = preset
        self.config = FrozenDict(block_out_channels=[128, 256, 512, 512], scaling_factor=0.18215)

    def encode(self, x):
        return types.SimpleNamespace(latent_dist=_Dist(self.preset))

    def decode(self, z, return_dict=False):
        return (z,)


class _SchedWrap:
    """diffusers-style wrapper over an oracle scheduler (config / set_timesteps / step(return_dict))."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, impl, with_eta):
        self.impl = impl
        self.config = FrozenDict(steps_offset=1, clip_sample=False)
        if with_eta:
            self.step = self._step_eta
        else:
            self.step = self._step_plain

    def set_timesteps(self, n, device=None):
        self.impl.set_timesteps(n)
        self.timesteps = self.impl.timesteps

    def scale_model_input(self, x, t):
        return x

    def _step_eta(self, model_output, timestep, sample, eta=0.0, generator=None, return_dict=True):
        return (self.impl.step(model_output.float(), timestep, sample.float(), eta=eta),)

    def _step_plain(self, model_output, timestep, sample, return_dict=True):
        return (self.impl.step(model_output.float(), timestep, sample.float()),)


def make_scheduler(kind: str):
    return _SchedWrap(DDIMOracle(), True) if kind == "ddim" else _SchedWrap(UniPCOracle(), False)


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    return torch.randn(shape, generator=generator, dtype=dtype)


def install():
    """Register the fake package tree in sys.modules (idempotent)."""
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_pcdm_stub", False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

    logging = types.SimpleNamespace(get_logger=lambda name=None: _Logger())
    root = mod("diffusers", DiffusionPipeline=DiffusionPipeline, _pcdm_stub=True)
    root.__path__ = []
    mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config, FrozenDict=FrozenDict)
    mod("diffusers.loaders", UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {}),
        LoraLoaderMixin=type("LoraLoaderMixin", (), {}))
    u = mod("diffusers.utils", BaseOutput=BaseOutput, logging=logging, deprecate=lambda *a, **k: None,
            is_accelerate_available=lambda: False, is_accelerate_version=lambda *a: False,
            replace_example_docstring=lambda doc: (lambda fn: fn))
    u.__path__ = []
    mod("diffusers.utils.torch_utils", randn_tensor=randn_tensor)
    m = mod("diffusers.models", AutoencoderKL=FakeVAE)
    m.__path__ = []
    mod("diffusers.models.activations", get_activation=lambda name: nn.SiLU())
    mod("diffusers.models.attention_processor", AttentionProcessor=_Any, AttnProcessor=_Any)
    mod("diffusers.models.attention", BasicTransformerBlock=BasicTransformerBlock)
    pl = mod("diffusers.pipelines")
    pl.__path__ = []
    mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline)
    mod("diffusers.models.embeddings", GaussianFourierProjection=_Any, TextImageProjection=_Any,
        TextImageTimeEmbedding=_Any, TextTimeEmbedding=_Any, TimestepEmbedding=TimestepEmbedding, Timesteps=Timesteps)
    mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    mod("diffusers.models.unet_2d_blocks", CrossAttnDownBlock2D=CrossAttnDownBlock2D, CrossAttnUpBlock2D=CrossAttnUpBlock2D,
        DownBlock2D=DownBlock2D, UNetMidBlock2DCrossAttn=UNetMidBlock2DCrossAttn,
        UNetMidBlock2DSimpleCrossAttn=_Any, UpBlock2D=UpBlock2D, get_down_block=get_down_block, get_up_block=get_up_block)
    mod("diffusers.image_processor", VaeImageProcessor=VaeImageProcessor)
    mod("diffusers.schedulers", KarrasDiffusionSchedulers=_Any, DDIMScheduler=_Any, DPMSolverMultistepScheduler=_Any,
        EulerAncestralDiscreteScheduler=_Any, EulerDiscreteScheduler=_Any, LMSDiscreteScheduler=_Any, PNDMScheduler=_Any, UnCLIPScheduler=_Any)
