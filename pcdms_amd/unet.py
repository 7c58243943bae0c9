"""MI355X-native stand-in for the reference's ``pipe.unet``.

Mirrors the interface of ``Stage2_InapintUNet2DConditionModel``
(/root/reference/src/models/stage2_inpaint_unet_2d_condition.py:61; ctor :66-448, forward
:579-825) as used by the drivers (stage2_batchtest_inpaint_model.py:125-133) and pipelines
(src/pipelines/stage2_inpaint_pipeline.py:504-506): ``from_pretrained`` / ``load_state_dict`` with
diffusers key names / ``.to`` / ``.config`` / ``forward(sample, timestep, encoder_hidden_states,
class_labels, ..., my_pose_cond, return_dict)``.

Nothing in here computes with PyTorch: the forward is a schedule of calls into libpcdm.so
(hand-written gfx950 HIP, include/pcdm.h) on the current stream.  Activations are NHWC bf16; all
scratch buffers are preallocated per input shape (static addresses => the whole step can be captured
in a hipGraph by the pipeline).  Step-invariant work (class embedding, pose layout change,
cross-attention K/V of the context) is computed by ``prepare_conditioning`` into a
``Conditioning`` object: the pipelines build one per sampling call and hand it to every step; bare
``unet(...)`` callers get it through a cache that holds REFERENCES to the three source tensors (so their
addresses cannot be recycled while the entry lives) and compares tensor identity + in-place version.
"""
from __future__ import annotations

import json
import math
import os
from pathlib import Path
from types import SimpleNamespace
from typing import Any, Dict, Iterator, List, Optional, Tuple, Union

import torch

from . import ops
from ._module import ModuleSurface
from .ops import BF16, PackedWeight

SHARE_CFG_PREFIX = os.environ.get("PCDM_SHARE_CFG_PREFIX", "1") != "0"   # A/B switch (tools/README.md)
TIME_TABLE = os.environ.get("PCDM_TIME_TABLE", "1") != "0"               # A/B switch: time embeddings of all steps with the conditioning
# ff.net.2 (+ residual) -> proj_out (+ residual) as ONE two-source GEMM (round 6).  Nothing non-linear sits between the two Linears
# (diffusers BasicTransformerBlock / Transformer2DModel as composed at stage2_inpaint_unet_2d_condition.py:321-361,407-430):
#     proj_out(ff2(g) + h2) + x  =  g (Wp W2)^T + h2 Wp^T + (Wp b2 + bp) + x
# so the block's tail is a GEMM over the channel concat [g | h2] (K = 4C + C) against [Wp W2 | Wp], composed once at load time in fp64 and
# rounded to bf16 like any weight: the same 2 M C 5C FLOPs, one launch instead of two, and the block's last residual state (h3: M x C bf16)
# is neither written nor read back.  PCDM_FUSE_FF_OUT=0: the two launches (A/B switch, tools/README.md).
FUSE_FF_OUT = os.environ.get("PCDM_FUSE_FF_OUT", "1") != "0"
# ResnetBlock2D: conv2(h) + conv_shortcut(x) as ONE contraction (round 6).  The 1x1 shortcut over the block's input is K-concatenated behind
# conv2's nine taps (ops.pack_conv3x3_shortcut; pcdm_gemm_params.a3, ABI 5): the 14 shortcut launches of a step and the read-back of their
# outputs as conv2's residual disappear.  PCDM_FUSE_SHORTCUT=0: the two launches (A/B switch, tools/README.md).
FUSE_SHORTCUT = os.environ.get("PCDM_FUSE_SHORTCUT", "1") != "0"
# Upsample2D: conv3x3(nearest-upsample x2 (x)) as its PHASE DECOMPOSITION (round 6).  Output pixel (2y + a, 2x + b) of the convolution on the
# upsampled tensor reads only the low-res pixels {y - 1 + a, y + a} x {x - 1 + b, x + b}: four 2x2 convolutions of x with summed taps -- one
# 3x3 launch on the LOW-RES tensor with N = 4 C (ops.pack_upsample_phases: each output-channel group = phase contracts over its four taps
# only, pcdm_gemm_params.tap_lut) and a pixel shuffle: 4/9 of the FLOPs of the three most expensive convolutions of the step.  Used when the
# target size is exactly (2 H, 2 W) (the reference's output_size path for odd skip sizes keeps the gather form).  PCDM_PHASE_UPSAMPLE=0: off.
PHASE_UPSAMPLE = os.environ.get("PCDM_PHASE_UPSAMPLE", "1") != "0"

_DEFAULT_CONFIG: Dict[str, Any] = dict(
    sample_size=None, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
    cross_attention_dim=1280, encoder_hid_dim=None, encoder_hid_dim_type=None, attention_head_dim=8,
    num_attention_heads=None, dual_cross_attention=False, use_linear_projection=False, class_embed_type=None,
    addition_embed_type=None, num_class_embeds=None, upcast_attention=False, resnet_time_scale_shift="default",
    resnet_skip_time_act=False, resnet_out_scale_factor=1.0, time_embedding_type="positional",
    time_embedding_dim=None, time_embedding_act_fn=None, timestep_post_act=None, time_cond_proj_dim=None,
    conv_in_kernel=3, conv_out_kernel=3, projection_class_embeddings_input_dim=None,
    class_embeddings_concat=False, mid_block_only_cross_attention=None, cross_attention_norm=None,
    addition_embed_type_num_heads=64,
)

# SD-2.1-base ``unet/config.json``; used when no config.json is on disk.
SD21_BASE_UNET_CONFIG: Dict[str, Any] = dict(
    sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True,
    norm_num_groups=32, norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0, act_fn="silu",
    _diffusers_version="0.24.0",
)

# values of the reference ctor this implementation supports (anything else -> NotImplementedError)
_REQUIRED = dict(center_input_sample=False, mid_block_type="UNetMidBlock2DCrossAttn", only_cross_attention=False,
                 act_fn="silu", encoder_hid_dim=None, encoder_hid_dim_type=None, dual_cross_attention=False,
                 addition_embed_type=None, num_class_embeds=None, resnet_time_scale_shift="default",
                 resnet_skip_time_act=False, time_embedding_type="positional", time_embedding_act_fn=None,
                 timestep_post_act=None, time_cond_proj_dim=None, conv_in_kernel=3, conv_out_kernel=3,
                 class_embeddings_concat=False, cross_attention_norm=None, downsample_padding=1)


class UNet2DConditionOutput:
    """``diffusers.models.unet_2d_condition.UNet2DConditionOutput`` stand-in (attribute + index access)."""

    def __init__(self, sample: torch.Tensor):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]

    def to_tuple(self):
        return (self.sample,)


class _Config(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, d=None):
        return getattr(self, k, d)

    def keys(self):
        return self.__dict__.keys()


def _as_tuple(v, n):
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * n


class Conditioning:
    """Step-invariant part of one sampling call (``prepare_conditioning``): views of the model's scratch buffers."""

    __slots__ = ("gen", "B", "h", "w", "L", "n0", "cls_emb", "pose_nhwc", "kv", "shared_halves", "temb_all", "temb_steps", "step_error")

    def __init__(self, gen: int, B: int, h: int, w: int, L: int):
        self.gen, self.B, self.h, self.w, self.L = gen, B, h, w, L
        self.n0 = 0                      # leading batch entries with an all-zero context (cross-attention skipped there)
        self.shared_halves = False       # batch entries b and b + B/2 have the same sample / mask / masked latents / pose (the CFG halves)
        self.temb_all: Optional[torch.Tensor] = None   # fp32 [steps, B, sum Cout]: every resnet's time_emb_proj(silu(emb)) for EVERY step
        self.temb_steps: Optional[torch.Tensor] = None  # the timestep table it was computed for
        self.step_error: Optional[torch.Tensor] = None  # device int32: a forward found the step counter outside that table (clamped on the device)
        self.cls_emb: Optional[torch.Tensor] = None
        self.pose_nhwc: Optional[torch.Tensor] = None
        self.kv: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}


class Stage2_InapintUNet2DConditionModel(ModuleSurface):
    """Drop-in for the reference class of the same (sic) name; inference only."""

    _pose_required = True   # the stage-2 forward adds my_pose_cond unconditionally (ref :742)

    def __init__(self, **kwargs):
        cfg = dict(_DEFAULT_CONFIG)
        unknown = [k for k in kwargs if k not in cfg and not k.startswith("_")]
        if unknown:
            raise TypeError(f"unexpected config keys: {unknown}")
        cfg.update(kwargs)
        for k, v in _REQUIRED.items():
            if cfg[k] != v and not (k == "only_cross_attention" and not any(_as_tuple(cfg[k], 1))):
                raise NotImplementedError(f"config {k}={cfg[k]!r} is outside the stage-2 hot path (supported: {v!r})")
        if cfg["class_embed_type"] not in (None, "projection"):
            raise NotImplementedError(f"class_embed_type={cfg['class_embed_type']!r}")
        if cfg["class_embed_type"] == "projection" and cfg["projection_class_embeddings_input_dim"] is None:
            raise ValueError("`class_embed_type`: 'projection' requires `projection_class_embeddings_input_dim` be set")
        n = len(cfg["down_block_types"])
        if len(cfg["up_block_types"]) != n or len(cfg["block_out_channels"]) != n:
            raise ValueError("Must provide the same number of down/up block types and block_out_channels")
        cfg.setdefault("_class_name", "Stage2_InapintUNet2DConditionModel")
        cfg.setdefault("_diffusers_version", "0.24.0")
        self.config = _Config(**cfg)
        heads = cfg["num_attention_heads"] or cfg["attention_head_dim"]  # ref :122-128 (mis-named head COUNT)
        self._heads = _as_tuple(heads, n)
        self._boc = tuple(cfg["block_out_channels"])
        self._layers = _as_tuple(cfg["layers_per_block"], n)
        if len(set(self._layers)) != 1:
            raise NotImplementedError("per-block layers_per_block")
        for c, h in zip(self._boc, self._heads):
            if c % h or c // h != 64:
                raise NotImplementedError(f"attention head size {c}/{h} != 64 (the HIP attention kernel is head_dim 64)")
            if c % 64:
                raise NotImplementedError("block_out_channels must be multiples of 64")
        if cfg["cross_attention_dim"] % 64 if isinstance(cfg["cross_attention_dim"], int) else True:
            raise NotImplementedError("cross_attention_dim must be an int multiple of 64")
        self.num_upsamplers = n - 1
        self.sample_size = cfg["sample_size"]
        self._device = torch.device("cpu")
        self._dtype = torch.float32
        self._sd: Optional[Dict[str, torch.Tensor]] = None   # fp32 CPU master copy (diffusers key names)
        self._w: Optional[Dict[str, Any]] = None             # packed device weights
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self._cache: Dict[str, Any] = {}
        self._cond_gen = 0   # bumped by every prepare_conditioning (the shared K/V, pose and class-embedding buffers are rewritten)
        self._attn_fp8 = False   # e4m3 K / V^T / Q / P attention on the MX-scaled fp8 MFMA (set_attention_precision)

    # ------------------------------------------------------------------ nn.Module-like surface
    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def modules(self):
        return iter(())

    def parameters(self):
        return iter((self._sd or {}).values())

    def eval(self):
        return self

    def requires_grad_(self, flag: bool = False):
        return self

    def set_use_memory_efficient_attention_xformers(self, valid: bool = True, attention_op=None):
        """No-op: the fused HIP attention kernel is always used (ref stage2_batchtest_inpaint_model.py:133)."""

    enable_xformers_memory_efficient_attention = set_use_memory_efficient_attention_xformers

    def set_attention_precision(self, precision: str = "bf16"):
        """``"bf16"`` (default; the reference's fp16 attention maps to it) or ``"fp8"``: every attention of the UNet (self and cross)
        runs with OCP e4m3 operands on the MX-scaled fp8 MFMA (BASELINE.json configs[4]; no reference counterpart).  K and V^T are
        quantised once per projection (``pcdm_quantize_fp8``), Q and P inside the kernel; softmax stays fp32.  Looser parity: see
        tests/test_unet.py::test_unet_fp8_attention for the stated tolerance.  Changing it invalidates captured graphs."""
        if precision not in ("bf16", "fp8"):
            raise ValueError("attention precision must be 'bf16' or 'fp8'")
        if (precision == "fp8") != self._attn_fp8:
            self._attn_fp8 = precision == "fp8"
            self._pack_gen = getattr(self, "_pack_gen", 0) + 1   # (pipelines re-capture their hipGraph)
            self._cache.clear()
            self._cond_gen += 1                                   # outstanding Conditioning objects lack / carry the e4m3 K, V^T
        return self

    def to(self, *args, **kwargs):
        device, dtype = kwargs.get("device"), kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif a is not None:
                device = torch.device(a)
        if dtype is not None:
            self._dtype = dtype  # I/O dtype only; arithmetic is bf16 x bf16 -> fp32 on MFMA
        if device is not None and torch.device(device) != self._device:
            self._device = torch.device(device)
            if self._device.type == "cuda" and self._device.index is None:
                self._device = torch.device("cuda", torch.cuda.current_device())
            self._w = None
            self._bufs.clear()
            self._cache.clear()
        return self

    def half(self):
        return self.to(torch.float16)

    # ------------------------------------------------------------------ weights
    def expected_shapes(self) -> Dict[str, Tuple[int, ...]]:
        return dict(_param_shapes(self))

    def state_dict(self) -> Dict[str, torch.Tensor]:
        if self._sd is None:
            raise RuntimeError("no weights loaded")
        return dict(self._sd)

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        exp = self.expected_shapes()
        missing = [k for k in exp if k not in state_dict]
        unexpected = [k for k in state_dict if k not in exp]
        bad = [f"{k}: {tuple(state_dict[k].shape)} vs {exp[k]}" for k in exp
               if k in state_dict and tuple(state_dict[k].shape) != tuple(exp[k])]
        if bad or (strict and (missing or unexpected)):
            raise RuntimeError("Error(s) in loading state_dict for Stage2_InapintUNet2DConditionModel:\n"
                               f"  Missing key(s): {missing[:8]}{'...' if len(missing) > 8 else ''}\n"
                               f"  Unexpected key(s): {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}\n"
                               f"  size mismatch: {bad[:8]}")
        base = self._sd or {}
        self._sd = {k: (state_dict[k].detach().to("cpu", torch.float32) if k in state_dict else base[k]) for k in exp}
        self._w = None
        self._cache.clear()
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def init_weights(self, seed: int = 0):
        """PyTorch-default init (what from_pretrained leaves in re-initialised tensors)."""
        g = torch.Generator().manual_seed(seed)
        exp = self.expected_shapes()
        sd = {}
        for k, shp in exp.items():
            wk = k[: k.rfind(".") + 1] + "weight"
            if len(exp[wk]) == 1:
                sd[k] = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
            else:
                bound = 1.0 / math.sqrt(math.prod(exp[wk][1:]))
                sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
        self._sd = sd
        self._w = None
        return self

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = dict(config.__dict__ if isinstance(config, SimpleNamespace) else config)
        cfg = {k: v for k, v in cfg.items() if k in _DEFAULT_CONFIG or k.startswith("_")}
        cfg.update(kwargs)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: Optional[str] = None, torch_dtype=None,
                        low_cpu_mem_usage: bool = False, ignore_mismatched_sizes: bool = False, **kwargs):
        """Reads ``{path}/{subfolder}/config.json`` + ``diffusion_pytorch_model.{safetensors,bin}``; ctor kwargs
        override the config; with ``ignore_mismatched_sizes`` shape-mismatched tensors (conv_in when
        in_channels 4 -> 9) keep their fresh init, as diffusers does (ref stage2_batchtest_inpaint_model.py:125-128).
        A path without config.json falls back to the SD-2.1-base UNet config."""
        root = Path(str(pretrained_model_name_or_path))
        d = root / subfolder if subfolder else root
        cfg = dict(SD21_BASE_UNET_CONFIG)
        cj = d / "config.json"
        if cj.exists():
            cfg = {k: v for k, v in json.loads(cj.read_text()).items() if k in _DEFAULT_CONFIG or k.startswith("_")}
        cfg.update(kwargs)
        model = cls(**cfg)
        model.init_weights(seed=0)
        sd = None
        if (d / "diffusion_pytorch_model.safetensors").exists():
            from safetensors.torch import load_file
            sd = load_file(str(d / "diffusion_pytorch_model.safetensors"))
        elif (d / "diffusion_pytorch_model.bin").exists():
            sd = torch.load(str(d / "diffusion_pytorch_model.bin"), map_location="cpu")
        if sd is not None:
            exp = model.expected_shapes()
            mism = [k for k in sd if k in exp and tuple(sd[k].shape) != tuple(exp[k])]
            if mism and not ignore_mismatched_sizes:
                raise ValueError(f"size mismatch for {mism}; pass ignore_mismatched_sizes=True")
            model.load_state_dict({k: v for k, v in sd.items() if k in exp and k not in mism}, strict=False)
        if torch_dtype is not None:
            model.to(torch_dtype)
        return model

    # ------------------------------------------------------------------ packing
    def _pack(self):
        if self._sd is None:
            raise RuntimeError("weights not loaded: call load_state_dict / from_pretrained first")
        if self._device.type != "cuda" and not _emu():
            raise RuntimeError("Stage2_InapintUNet2DConditionModel runs on the MI355X only: call .to('cuda') "
                               "(there is no CPU implementation)")
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1   # a captured hipGraph holds pointers into the packed weights
        sd, dev = self._sd, self._device
        w: Dict[str, Any] = {}

        def f32(k):
            return sd[k].to(dev, torch.float32).contiguous()

        def lin(p, bias=True):
            return ops.pack_linear(sd[p + "weight"], sd[p + "bias"] if bias and p + "bias" in sd else None, dev)

        def conv(p):
            return ops.pack_conv3x3(sd[p + "weight"], sd[p + "bias"], dev)

        def small(p):
            return sd[p + "weight"].to(BF16).to(dev).contiguous(), f32(p + "bias")

        w["conv_in"] = conv("conv_in.")
        w["time1"], w["time2"] = small("time_embedding.linear_1."), small("time_embedding.linear_2.")
        if self.config.class_embed_type == "projection":
            w["class1"], w["class2"] = small("class_embedding.linear_1."), small("class_embedding.linear_2.")
        tw, tb, off = [], [], 0
        for p, cin, cout, _ in _resnets(self):
            r: Dict[str, Any] = dict(cin=cin, cout=cout, toff=off)
            r["n1"] = (f32(p + "norm1.weight"), f32(p + "norm1.bias"))
            r["n2"] = (f32(p + "norm2.weight"), f32(p + "norm2.bias"))
            r["conv1"] = conv(p + "conv1.")
            if cin != cout and FUSE_SHORTCUT:   # (the composed rows replace both layers: neither is packed a second time)
                r["conv2s"] = ops.pack_conv3x3_shortcut(sd[p + "conv2.weight"], sd[p + "conv2.bias"], sd[p + "conv_shortcut.weight"],
                                                        sd[p + "conv_shortcut.bias"], dev)
            else:
                r["conv2"] = conv(p + "conv2.")
                if cin != cout:
                    r["short"] = lin(p + "conv_shortcut.")
            tw.append(sd[p + "time_emb_proj.weight"])
            tb.append(sd[p + "time_emb_proj.bias"])
            off += cout
            w[p] = r
        # all 22 time_emb_proj as ONE [sum Cout = 20160, 1280] linear: an MFMA GEMM with M = batch rows (the 52 MB of weights stream
        # once; the one-wave-per-output GEMV it replaces re-read the 8 activation rows for every output: 50 us against ~15)
        w["temb"] = ops.pack_linear(torch.cat(tw, 0), torch.cat(tb, 0), dev)
        w["temb_n"] = off
        for p, c, h in _transformers(self):
            a: Dict[str, Any] = dict(c=c, heads=h)
            a["norm"] = (f32(p + "norm.weight"), f32(p + "norm.bias"))
            a["proj_in"] = lin(p + "proj_in.")
            b = p + "transformer_blocks.0."
            for i in (1, 2, 3):
                a[f"ln{i}"] = (f32(b + f"norm{i}.weight"), f32(b + f"norm{i}.bias"))
            a["qkv"] = ops.pack_linear(torch.cat([sd[b + "attn1.to_q.weight"], sd[b + "attn1.to_k.weight"],
                                                  sd[b + "attn1.to_v.weight"]], 0), None, dev)
            a["o1"] = lin(b + "attn1.to_out.0.")
            a["q2"] = ops.pack_linear(sd[b + "attn2.to_q.weight"], None, dev)
            a["kv2"] = ops.pack_linear(torch.cat([sd[b + "attn2.to_k.weight"], sd[b + "attn2.to_v.weight"]], 0), None, dev)
            a["o2"] = lin(b + "attn2.to_out.0.")
            a["ff1"] = ops.pack_geglu(sd[b + "ff.net.0.proj.weight"], sd[b + "ff.net.0.proj.bias"], dev)
            if FUSE_FF_OUT:   # (the composed weight replaces both layers: neither is packed a second time)
                w2, b2 = sd[b + "ff.net.2.weight"].double(), sd[b + "ff.net.2.bias"].double()
                wp, bp = sd[p + "proj_out.weight"].double().reshape(c, c), sd[p + "proj_out.bias"].double()
                a["ffo"] = ops.pack_linear(torch.cat([wp @ w2, wp], 1).float(), (wp @ b2 + bp).float(), dev)
            else:
                a["ff2"], a["proj_out"] = lin(b + "ff.net.2."), lin(p + "proj_out.")
            if True:   # second copies of the three LayerNorm-fed linears with the LayerNorm FOLDED into the weights: K = 320 -> the
                       # A-in-registers kernel (rowgemm.hip); K = 640 / 1280 (round 5) -> the LNF instances of the tiled kernel (gemm.hip)
                ln = [(sd[b + f"norm{i}.weight"], sd[b + f"norm{i}.bias"]) for i in (1, 2, 3)]
                a["qkv_ln"] = ops.pack_linear_ln(torch.cat([sd[b + "attn1.to_q.weight"], sd[b + "attn1.to_k.weight"],
                                                            sd[b + "attn1.to_v.weight"]], 0), None, ln[0][0], ln[0][1], dev)
                a["q2_ln"] = ops.pack_linear_ln(sd[b + "attn2.to_q.weight"], None, ln[1][0], ln[1][1], dev)
                a["ff1_ln"] = ops.pack_geglu_ln(sd[b + "ff.net.0.proj.weight"], sd[b + "ff.net.0.proj.bias"], ln[2][0], ln[2][1], dev)
            w[p] = a
        for i in range(len(self._boc) - 1):
            w[f"down_blocks.{i}.downsamplers.0.conv."] = conv(f"down_blocks.{i}.downsamplers.0.conv.")
            w[f"up_blocks.{i}.upsamplers.0.conv."] = conv(f"up_blocks.{i}.upsamplers.0.conv.")
            if PHASE_UPSAMPLE:
                w[f"up_blocks.{i}.upsamplers.0.conv4."] = ops.pack_upsample_phases(sd[f"up_blocks.{i}.upsamplers.0.conv.weight"],
                                                                                   sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], dev)
        w["norm_out"] = (f32("conv_norm_out.weight"), f32("conv_norm_out.bias"))
        w["conv_out"] = conv("conv_out.")
        self._w = w

    # ------------------------------------------------------------------ scratch / caches
    def _buf(self, name: str, shape, dtype=BF16, zero: bool = False) -> torch.Tensor:
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self._device)
            self._bufs[key] = t
        return t

    def invalidate_caches(self):
        """Drop the bare-``forward`` conditioning cache (and the references it holds to the caller's tensors)."""
        self._cache.clear()

    # ------------------------------------------------------------------ step-invariant conditioning
    def prepare_conditioning(self, B: int, h: int, w: int, encoder_hidden_states: torch.Tensor,
                             class_labels: Optional[torch.Tensor], my_pose_cond: Optional[torch.Tensor],
                             zero_ctx_batches: Optional[int] = None, shared_cfg_input: bool = False,
                             timesteps: Optional[torch.Tensor] = None) -> "Conditioning":
        """Everything of one forward that does not depend on the timestep or the latents:
        class embedding (ref :688-708), NHWC pose feature (ref :742) and the cross-attention K / V^T of the context for all
        16 transformer blocks.  Results live in shape-keyed scratch buffers (static addresses: a captured hipGraph stays valid
        across calls) that every later ``prepare_conditioning`` of the same shape OVERWRITES -- the returned object is valid
        until then, which ``_forward_nhwc`` checks through the generation counter.

        ``zero_ctx_batches`` = n0: the first n0 batch entries of ``encoder_hidden_states`` are all-zero (the CFG
        unconditional half, ref stage2_inpaint_pipeline.py:457-458).  ``to_k`` / ``to_v`` have no bias, so K = V = 0 there and
        the attention output is exactly 0, i.e. ``attn2(x) == to_out.0.bias``: those rows skip
        LayerNorm-2, ``to_q``, the attention and the ``to_out`` contraction.  ``None`` = find out (one device reduction + sync).

        ``shared_cfg_input``: the caller guarantees that batch entries b and b + B/2 will always carry the SAME ``sample`` and the same
        pose feature -- the two classifier-free-guidance halves (ref stage2_inpaint_pipeline.py:499-501: ``torch.cat([latents] * 2)``,
        one mask / masked latents / pose for both).  Only ``class_labels`` and the context differ between them, and those enter behind
        ``conv_in``, the first GroupNorm and the contraction of the first ``conv1``: that prefix is then computed for B/2 entries and
        written for both halves (``pcdm_gemm_params.dup_rows``) -- the same arithmetic, executed once.

        ``timesteps`` (device int64 table of the whole schedule): the time / class embedding MLPs and all 22 ``time_emb_proj`` rows depend
        on the timestep and the class labels only, so they are computed HERE for every step (one GEMM with M = steps x B rows) instead of
        five launches per denoise step; a forward that is given the table's device step counter (``_forward_nhwc(..., step_dev)``) picks
        its block through ``pcdm_gemm_params.rowvec_step``.  Same per-row arithmetic as the per-step launches (bit-identical)."""
        if self._w is None:
            self._pack()
        W, cfg, dev = self._w, self.config, self._device
        boc = self._boc
        temb_dim = boc[0] * 4
        ehs = encoder_hidden_states
        if ehs.dim() != 3 or ehs.shape[0] != B or ehs.shape[2] != cfg.cross_attention_dim:
            raise ValueError(f"encoder_hidden_states must be [{B},L,{cfg.cross_attention_dim}], got {tuple(ehs.shape)}")
        L = ehs.shape[1]
        self._cond_gen += 1
        cond = Conditioning(gen=self._cond_gen, B=B, h=h, w=w, L=L)
        cond.shared_halves = bool(shared_cfg_input) and B % 2 == 0 and B >= 2 and SHARE_CFG_PREFIX
        if cfg.class_embed_type == "projection":
            if class_labels is None:
                raise ValueError("class_labels should be provided when num_class_embeds > 0")
            cl = class_labels.reshape(B, -1).to(dev, torch.float32).contiguous()
            c1 = ops.small_linear(cl, W["class1"][0], W["class1"][1], self._buf("cls1", (B, temb_dim), torch.float32),
                                  act_out=True)
            cond.cls_emb = ops.small_linear(c1, W["class2"][0], W["class2"][1], self._buf("cls2", (B, temb_dim), torch.float32))
        if my_pose_cond is not None:
            pose = my_pose_cond
            if pose.dim() != 4 or pose.shape[0] not in (1, B) or tuple(pose.shape[1:]) != (boc[0], h, w):
                raise ValueError(f"my_pose_cond must be [1|{B},{boc[0]},{h},{w}], got {tuple(pose.shape)}")
            # one NHWC copy per batch entry, also for a batch-1 pose feature (the reference doubles / repeats it, ref pipeline :457-459): the
            # conv_in epilogue then adds it as a plain residual (res_mod = M) on the 16-byte-per-lane path -- a broadcast residual
            # (res_mod = h w) takes the kernel's per-element epilogue, 59 against ~30 us for the launch
            buf = self._buf("pose", (B, h, w, boc[0]))
            pose = pose.to(dev)
            if pose.shape[0] == B:
                ops.nchw_to_nhwc_bf16(pose, buf)
            else:
                for b in range(B):
                    ops.nchw_to_nhwc_bf16(pose, buf[b:b + 1])
            cond.pose_nhwc = buf
        e32 = ehs.to(dev, torch.float32).contiguous()
        if zero_ctx_batches is None:   # leading all-zero batch entries (bare callers; the pipelines know and say so)
            nz = (e32.reshape(B, -1) != 0).any(dim=1).to(torch.int32)
            zero_ctx_batches = int(torch.cumsum(nz, 0).eq(0).sum().item())
        n0 = int(zero_ctx_batches)
        if not 0 <= n0 <= B:
            raise ValueError("zero_ctx_batches out of range")
        if n0 == B:
            n0 = B - 1 if B > 1 else 0   # keep the kernels' shapes non-empty (the last entry is then computed in full)
        cond.n0 = n0
        Bc = B - n0
        ctx = ops.f32_to_bf16(e32[n0:].reshape(Bc * L, -1), self._buf("ctx", (Bc * L, cfg.cross_attention_dim)))
        Lp = (L + 7) // 8 * 8
        for p, c, _ in _transformers(self):
            kbuf = self._buf(("k2", p), (Bc * L, c))
            vtbuf = self._buf(("vt2", p), (Bc, c, Lp), zero=True)
            ops.gemm(ctx, W[p]["kv2"], kbuf, rows_per_batch=L, epilogue=ops.EPI_SPLIT_VT, out2=vtbuf, vt_col0=c)
            if self._attn_fp8:   # e4m3 copies, once per call
                L16 = (L + 15) // 16 * 16
                k8 = ops.quantize_fp8(kbuf, self._buf(("k2_8", p), (Bc * L, c), ops.FP8))
                vt8 = ops.quantize_fp8(vtbuf.view(Bc * c, Lp), self._buf(("vt2_8", p), (Bc * c, L16), ops.FP8), cols=L)
                cond.kv[p] = (k8, vt8.view(Bc, c, L16))
            else:
                cond.kv[p] = (kbuf, vtbuf)
        if timesteps is not None and TIME_TABLE:
            self._time_table(cond, timesteps)
        return cond

    def _time_table(self, cond: "Conditioning", timesteps: torch.Tensor) -> None:
        """cond.temb_all[i, b, :] = concat over resnets of time_emb_proj(silu(time_embedding(t_i) + class_embedding(b))) (ref :661-708 and every
        ResnetBlock2D's ``time_emb_proj(act(temb))``), for all steps i of the table: 2 + 2 x ceil(n / 32) + 2 launches per sampling call."""
        W, cfg, dev, B = self._w, self.config, self._device, cond.B
        boc = self._boc
        temb_dim = boc[0] * 4
        ts = timesteps.reshape(-1)
        if ts.dtype != torch.int64 or ts.device != dev:
            ts = ts.to(dev, torch.int64)
        n = ts.numel()
        t_emb = ops.timestep_embedding_rows(ts, self._buf("tt_emb", (n, boc[0]), torch.float32), cfg.flip_sin_to_cos, float(cfg.freq_shift))
        e1 = self._buf("tt_e1", (n, temb_dim), torch.float32)
        emb_t = self._buf("tt_emb_t", (n, temb_dim), torch.float32)
        for r0 in range(0, n, 32):   # (pcdm_small_linear takes <= 32 rows; rows are independent)
            r1 = min(n, r0 + 32)
            ops.small_linear(t_emb[r0:r1], W["time1"][0], W["time1"][1], e1[r0:r1], act_out=True)
            ops.small_linear(e1[r0:r1], W["time2"][0], W["time2"][1], emb_t[r0:r1])
        emb_bf = ops.time_class_combine(emb_t, cond.cls_emb, self._buf("tt_emb_bf", (n * B, temb_dim)), B)
        # the SAME tile as the per-step launch (M = B rows) uses: every output element is then the same sequence of MFMAs -> identical bits
        tile, split = ops._TUNED.get((B, W["temb"].Npad, W["temb"].K, 0, 0, 0, ops.EPI_NCHW_F32, False, False), (8, 1))
        out = self._buf("tt_all", (n * B, W["temb_n"]), torch.float32)
        ops.gemm(emb_bf, W["temb"], out, rows_per_batch=1, epilogue=ops.EPI_NCHW_F32, tile=tile if split <= 1 else 8)
        cond.temb_all = out.view(n, B, W["temb_n"])
        cond.temb_steps = ts
        cond.step_error = self._buf("step_err", (4,), torch.int32, zero=True)[:1]
        cond.step_error.zero_()

    def step_overflow(self) -> bool:
        """Did a forward since the last ``prepare_conditioning(timesteps=...)`` read the per-call time table with a device step counter outside ``[0, n)``?  The kernels
        clamp such a counter into the table (no out-of-bounds read) and raise this flag (pcdm_gemm_params.rowvec_step_count / step_error).
        Synchronises."""
        return bool(int(self._buf("step_err", (4,), torch.int32, zero=True)[0].item()) != 0)

    def _conditioning_for(self, B, h, w, ehs, class_labels, pose, zero_ctx_batches: Optional[int] = None) -> "Conditioning":
        """Bare ``forward`` callers (the reference pipeline's own loop): reuse the last conditioning
        iff the caller passed the SAME tensor objects, unmodified (torch's in-place version counter), and nobody has
        overwritten the buffers since.  The entry keeps the three tensors alive, so a fresh tensor can never sit at a
        cached tensor's address (the round-1 bug: keys were (data_ptr, _version) of tensors that had been freed)."""
        srcs = (ehs, class_labels, pose)
        hit = self._cache.get("cond")
        if hit is not None:
            hsrcs, vers, shape, cond = hit
            if shape == (B, h, w) and cond.gen == self._cond_gen and all(a is b for a, b in zip(srcs, hsrcs)) and \
                    vers == tuple(None if t is None else t._version for t in srcs) and \
                    (zero_ctx_batches is None or min(zero_ctx_batches, max(B - 1, 0)) == cond.n0):
                return cond
        cond = self.prepare_conditioning(B, h, w, ehs, class_labels, pose, zero_ctx_batches=zero_ctx_batches)
        self._cache["cond"] = (srcs, tuple(None if t is None else t._version for t in srcs), (B, h, w), cond)
        return cond

    # ------------------------------------------------------------------ forward
    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor, class_labels: Optional[torch.Tensor] = None,
                timestep_cond: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                added_cond_kwargs: Optional[Dict[str, torch.Tensor]] = None,
                down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
                mid_block_additional_residual: Optional[torch.Tensor] = None,
                encoder_attention_mask: Optional[torch.Tensor] = None, my_pose_cond: Optional[torch.Tensor] = None,
                return_dict: bool = True, _step_dev: Optional[torch.Tensor] = None,
                _cond: Optional["Conditioning"] = None, zero_ctx_batches: Optional[int] = None):
        """``zero_ctx_batches`` (extension; bare callers only): how many LEADING batch entries of ``encoder_hidden_states`` are
        all-zero (the CFG unconditional half): their cross-attention is skipped.  ``None`` = find out with one device reduction and a
        host sync per conditioning-cache miss -- except while the stream is being captured into a hipGraph, where no sync is possible
        and 0 is assumed (every row computed in full; same result).  Callers that rebuild ``encoder_hidden_states`` every step
        should pass the number (or 0) to avoid the sync; see INTEGRATION.md §1."""
        for name, v in (("timestep_cond", timestep_cond), ("attention_mask", attention_mask),
                        ("added_cond_kwargs", added_cond_kwargs),
                        ("down_block_additional_residuals", down_block_additional_residuals),
                        ("mid_block_additional_residual", mid_block_additional_residual),
                        ("encoder_attention_mask", encoder_attention_mask)):
            if v is not None:
                raise NotImplementedError(f"{name} is not part of the stage-2 path")
        if cross_attention_kwargs:
            raise NotImplementedError("cross_attention_kwargs (LoRA scale etc.) is not part of the stage-2 path")
        if self.config.class_embed_type is not None and class_labels is None:
            raise ValueError("class_labels should be provided when num_class_embeds > 0")
        if my_pose_cond is None and self._pose_required:
            raise ValueError("my_pose_cond is required (ref stage2_inpaint_unet_2d_condition.py:742)")
        if sample.dim() != 4 or sample.shape[1] != self.config.in_channels:
            raise ValueError(f"sample must be [B,{self.config.in_channels},h,w], got {tuple(sample.shape)}")
        B, _, h, w = sample.shape
        # latent sizes that are not multiples of 2**num_upsamplers (e.g. the stage-3 latent 64x44): the reference forwards the
        # skip's size to Upsample2D (ref :625-633,796-799); here the upsampling conv gathers to the next skip's size
        if self._w is None:
            self._pack()
        if sample.device != self._device:
            raise RuntimeError(f"sample on {sample.device}, model on {self._device}")
        x_in = ops.nchw_to_nhwc_bf16(sample, self._buf("x_in", (B, h, w, self._w["conv_in"].cin)),
                                     cpad=self._w["conv_in"].cin)
        if zero_ctx_batches is None and sample.is_cuda and torch.cuda.is_current_stream_capturing():
            zero_ctx_batches = 0
        cond = _cond if _cond is not None else self._conditioning_for(B, h, w, encoder_hidden_states, class_labels, my_pose_cond,
                                                                     zero_ctx_batches)
        eps = self._forward_nhwc(x_in, B, h, w, timestep, cond, _step_dev)
        # a fresh tensor at the public boundary (nn.Module semantics: two calls never alias); the fused pipeline path uses
        # _forward_nhwc directly and reads the reused buffer in place
        out = eps.clone() if sample.dtype == torch.float32 else eps.to(sample.dtype)
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)

    # -- the schedule proper; x_in is NHWC bf16 [B,h,w,cin_pad]; returns fp32 NCHW eps (a reused buffer)
    def _forward_nhwc(self, x_in, B, h, w, timestep, cond: Conditioning, step_dev=None) -> torch.Tensor:
        taps = getattr(self, "_taps", None)   # test hook: a dict set by a parity test receives the residual stream [B*HW, C] bf16 at the block boundaries
        W, cfg, dev = self._w, self.config, self._device
        boc, G, eps = self._boc, cfg.norm_num_groups, cfg.norm_eps
        nlev = len(boc)
        temb_dim = boc[0] * 4
        if cond.gen != self._cond_gen or (cond.B, cond.h, cond.w) != (B, h, w):
            raise RuntimeError("stale Conditioning: prepare_conditioning() was called again (the shared K/V / pose / class-embedding "
                               "buffers were overwritten) or the batch / latent size changed")
        if (cfg.class_embed_type == "projection") != (cond.cls_emb is not None) or (cond.pose_nhwc is None and self._pose_required):
            raise ValueError("class_labels / my_pose_cond missing from the conditioning")

        # ---- 1. time / class embedding (ref :661-708)
        if torch.is_tensor(timestep):
            t_dev = timestep.reshape(-1)[:1] if step_dev is None else timestep.reshape(-1)
            if t_dev.dtype != torch.int64 or t_dev.device != dev:
                t_dev = t_dev.to(dev, torch.int64)
        else:
            t_dev = torch.tensor([int(timestep)], dtype=torch.int64, device=dev)
        cls_emb, pose_nhwc, kv, L, nzero = cond.cls_emb, cond.pose_nhwc, cond.kv, cond.L, cond.n0
        rv_step, rv_stride, rv_count, rv_err = None, 0, 0, None
        if cond.temb_all is not None and step_dev is not None and torch.is_tensor(timestep) and timestep.data_ptr() == cond.temb_steps.data_ptr() \
                and timestep.numel() == cond.temb_steps.numel():
            # the time-embedding projections of every step were computed with the conditioning (prepare_conditioning(timesteps=...)): this
            # step's block is picked inside the kernels by the device step counter -- nothing to launch here
            temb = cond.temb_all[0]
            rv_step, rv_stride, rv_count, rv_err = step_dev, cond.temb_all.stride(0), cond.temb_all.shape[0], cond.step_error
        else:
            t_emb = ops.timestep_embedding(t_dev, step_dev, self._buf("t_emb", (B, boc[0]), torch.float32),
                                           cfg.flip_sin_to_cos, float(cfg.freq_shift))
            e1 = ops.small_linear(t_emb, W["time1"][0], W["time1"][1], self._buf("e1", (B, temb_dim), torch.float32), act_out=True)
            # emb = time_emb + class_emb is only ever consumed as silu(emb) (ResnetBlock2D): apply it here, once
            emb_act = ops.small_linear(e1, W["time2"][0], W["time2"][1], self._buf("emb", (B, temb_dim), torch.float32),
                                       add=cls_emb, act_out=2 if cls_emb is not None else 1)
            # every ResnetBlock2D.time_emb_proj(silu(emb)) in one launch
            emb_bf = ops.f32_to_bf16(emb_act, self._buf("emb_bf", (B, temb_dim)))
            temb = ops.gemm(emb_bf, W["temb"], self._buf("temb", (B, W["temb_n"]), torch.float32), rows_per_batch=1,
                            epilogue=ops.EPI_NCHW_F32)                      # fp32 [B, 20160] (rows_per_batch = 1: "NCHW" == row-major)

        # Split-K convolutions (levels 1-3: M <= 11264 rows against K up to 23040) leave their fp32 partial slabs for the GroupNorm that
        # always follows -- conv1 -> norm2, conv2 / down- / upsampling conv -> the next block's first norm -- which reduces them while it
        # loads its slab (ops.DeferredGemm / pcdm_groupnorm_splitk): no reduce launch, no bf16 round trip in front of the norm.  The bf16
        # tensor itself is written by that norm where a residual, shortcut or skip connection reads it later (`gn_next`: the caller
        # knows whether the next consumer of the block's output is a GroupNorm).
        def resnet(p, x1, x2, HW_, hh, ww, name, gn_next=False, shared=False):
            r = W[p]
            cin, cout, M = r["cin"], r["cout"], B * HW_
            ws = self._buf("gnws", (int(ops._lib.lib().pcdm_groupnorm_ws_floats(B, 4096)),), torch.float32, zero=True)
            tv = temb[:, r["toff"]: r["toff"] + cout]
            if shared:   # the CFG halves still have the same x1 here: norm1 and conv1's contraction once, two epilogues (temb rows b / b + B/2)
                Bs, Ms = B // 2, (B // 2) * HW_
                n1 = ops.groupnorm(x1[:Ms], None, Bs, HW_, G, eps, r["n1"][0], r["n1"][1], True, self._buf("gn", (M, cin))[:Ms], ws)
                h1 = ops.gemm(n1, r["conv1"], self._buf("c1", (M, cout)), conv=dict(B=Bs, Hi=hh, Wi=ww, Ho=hh, Wo=ww), rowvec=tv,
                              rows_per_batch=HW_, dup_rows=Ms, rowvec_step=rv_step, rowvec_step_stride=rv_stride, rowvec_step_count=rv_count,
                              step_error=rv_err)
                cv = dict(B=B, Hi=hh, Wi=ww, Ho=hh, Wo=ww)
            else:
                n1 = ops.groupnorm(x1, x2, B, HW_, G, eps, r["n1"][0], r["n1"][1], True, self._buf("gn", (M, cin)), ws)
                x1 = ops.as_tensor(x1)   # (written by the norm above if it was deferred)
                cv = dict(B=B, Hi=hh, Wi=ww, Ho=hh, Wo=ww)
                h1 = ops.gemm(n1, r["conv1"], self._buf("c1", (M, cout)), conv=cv, rowvec=tv, rows_per_batch=HW_, defer_reduce=False,
                              rowvec_step=rv_step, rowvec_step_stride=rv_stride, rowvec_step_count=rv_count, step_error=rv_err)
            n2 = ops.groupnorm(h1, None, B, HW_, G, eps, r["n2"][0], r["n2"][1], True, self._buf("gn", (M, cout)), ws)
            if "conv2s" in r:   # conv2 + conv_shortcut as one contraction: the block's input enters through the extra K (FUSE_SHORTCUT above)
                return ops.gemm(n2, r["conv2s"], self._buf(name, (M, cout)), conv=cv, a2=x1, a3=x2, defer_reduce=True if gn_next else None,
                                rows_per_batch=HW_)
            if "short" in r:
                res = ops.gemm(x1, r["short"], self._buf("sc", (M, cout)), a2=x2)
            else:
                res = x1
            return ops.gemm(n2, r["conv2"], self._buf(name, (M, cout)), conv=cv, residual=res, res_mod=M,
                            defer_reduce=True if gn_next else None, rows_per_batch=HW_)

        def transformer(p, x, HW_, name):
            a = W[p]
            c, H, M = a["c"], a["heads"], B * HW_
            ws = self._buf("gnws", (int(ops._lib.lib().pcdm_groupnorm_ws_floats(B, 4096)),), torch.float32, zero=True)
            n0 = ops.groupnorm(x, None, B, HW_, G, 1e-6, a["norm"][0], a["norm"][1], False, self._buf("gn", (M, c)), ws)
            x = ops.as_tensor(x)
            # Row statistics travel with the rows (round 5, levels 1-3): the linear that writes the input of a LayerNorm -- proj_in, attn1.to_out +
            # residual, attn2.to_out + residual -- also writes the {sum, M2} of every 32-column run (`row_stats`), and the LayerNorm-folded
            # projection that reads the rows next merges them: no LayerNorm launch, no statistics work in that GEMM's loop (ops._gemm_ln)
            r0 = nzero * HW_
            rs = self._buf("rs", (M, c // 32, 2), torch.float32) if c % 32 == 0 else None
            want = lambda pw_, M_, epi: rs if (rs is not None and pw_ is not None and ops.ln_wants_row_stats(M_, pw_, epi)) else None  # noqa: E731
            t0 = ops.gemm(n0, a["proj_in"], self._buf("t0", (M, c)), row_stats=want(a.get("qkv_ln"), M, ops.EPI_SPLIT_VT))
            # self-attention
            # LayerNorm -> projection pairs go through ops.gemm(ln=...): at K = 320 (level 0) the A-in-registers kernel takes the weights
            # with the LayerNorm folded in (no LayerNorm launch, no normalised tensor anywhere); elsewhere LayerNorm runs first into "ln"
            qk = self._buf("qk", (M, 2 * c))
            vt = self._buf("vt", (B, c, (HW_ + 7) // 8 * 8), zero=True)
            ops.gemm(t0, a["qkv"], qk, rows_per_batch=HW_, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * c,
                     ln=(a["ln1"][0], a["ln1"][1], 1e-5), ln_buf=self._buf("ln", (M, c)), pw_ln=a.get("qkv_ln"), row_stats=rs)
            if self._attn_fp8:
                HW16 = (HW_ + 15) // 16 * 16
                k8 = ops.quantize_fp8(qk[:, c:], self._buf("k8", (M, c), ops.FP8))
                vt8 = ops.quantize_fp8(vt.view(B * c, vt.shape[-1]), self._buf("vt8", (B * c, HW16), ops.FP8), cols=HW_)
                at = ops.flash_attn_fp8(qk[:, :c], k8, vt8.view(B, c, HW16), self._buf("at", (M, c)), B, H, HW_, HW_)
            else:
                at = ops.flash_attn(qk[:, :c], qk[:, c:], vt, self._buf("at", (M, c)), B, H, HW_, HW_)
            t1 = ops.gemm(at, a["o1"], self._buf("t1", (M, c)), residual=t0, res_mod=M, row_stats=want(a.get("q2_ln"), M - r0, ops.EPI_STORE))
            # cross-attention over the context tokens; the first n0 batch entries have an all-zero context, for which
            # attn2(x) == to_out.0.bias exactly: their rows skip LN2 / to_q / attention and enter
            # the to_out GEMM as zero A rows (no main loop for tiles that lie entirely inside them)
            k2, vt2 = kv[p]
            at2 = self._buf("at", (M, c))
            q2 = ops.gemm(t1[r0:], a["q2"], self._buf("q2", (M, c))[r0:], ln=(a["ln2"][0], a["ln2"][1], 1e-5),
                          ln_buf=self._buf("ln", (M, c))[r0:], pw_ln=a.get("q2_ln"), row_stats=None if rs is None else rs[r0:])
            (ops.flash_attn_fp8 if self._attn_fp8 else ops.flash_attn)(q2, k2, vt2, at2[r0:], B - nzero, H, HW_, L)
            t2 = ops.gemm(at2, a["o2"], self._buf("t0", (M, c)), residual=t1, res_mod=M, zero_rows=r0,
                          row_stats=want(a.get("ff1_ln"), M, ops.EPI_GEGLU))
            # GEGLU feed-forward
            ff = ops.gemm(t2, a["ff1"], self._buf("ff", (M, 4 * c)), epilogue=ops.EPI_GEGLU, ln=(a["ln3"][0], a["ln3"][1], 1e-5),
                          ln_buf=self._buf("ln", (M, c)), pw_ln=a.get("ff1_ln"), row_stats=rs)
            if "ffo" in a:   # ff.net.2 + residual and proj_out + residual as one contraction over [ff | t2] (FUSE_FF_OUT above)
                return ops.gemm(ff, a["ffo"], self._buf(name, (M, c)), a2=t2, residual=x, res_mod=M)
            t3 = ops.gemm(ff, a["ff2"], self._buf("t1", (M, c)), residual=t2, res_mod=M)
            return ops.gemm(t3, a["proj_out"], self._buf(name, (M, c)), residual=x, res_mod=M)

        # ---- 2. conv_in + pose (ref :742)
        HW = h * w
        shared = cond.shared_halves   # the CFG halves share conv_in (+ pose), the first norm1 and the first conv1's contraction
        if shared:
            Mh = (B // 2) * HW
            x = ops.gemm(x_in[: B // 2], W["conv_in"], self._buf("skip0", (B * HW, boc[0])), conv=dict(B=B // 2, Hi=h, Wi=w, Ho=h, Wo=w),
                         residual=None if pose_nhwc is None else pose_nhwc.view(-1, boc[0])[:Mh], res_mod=0 if pose_nhwc is None else Mh,
                         dup_rows=Mh)
        else:
            x = ops.gemm(x_in, W["conv_in"], self._buf("skip0", (B * HW, boc[0])), conv=dict(B=B, Hi=h, Wi=w, Ho=h, Wo=w),
                         residual=None if pose_nhwc is None else pose_nhwc.view(-1, boc[0]),
                         res_mod=0 if pose_nhwc is None else pose_nhwc.shape[0] * HW)
        # ---- 3. down (ref :746-761)
        skips: List[Tuple[torch.Tensor, int, int]] = [(x, h, w)]
        hh, ww = h, w
        L_ = self._layers[0]

        def skip_of(t):   # (a deferred tensor's buffer: filled by the norm that follows, long before the up path reads it)
            return t.out if isinstance(t, ops.DeferredGemm) else t
        for i, typ in enumerate(cfg.down_block_types):
            for j in range(L_):
                nm = f"d{i}.{j}"
                first = shared and i == 0 and j == 0
                if typ == "CrossAttnDownBlock2D":
                    x = resnet(f"down_blocks.{i}.resnets.{j}.", x, None, hh * ww, hh, ww, "r", gn_next=True, shared=first)
                    x = transformer(f"down_blocks.{i}.attentions.{j}.", x, hh * ww, nm)
                else:   # the next consumer is a resnet's norm1 (same block, or the mid block) unless a downsampling conv follows
                    x = resnet(f"down_blocks.{i}.resnets.{j}.", x, None, hh * ww, hh, ww, nm, gn_next=j < L_ - 1 or i == nlev - 1, shared=first)
                skips.append((skip_of(x), hh, ww))
            if i != nlev - 1:
                ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
                x = ops.gemm(x.view(B, hh, ww, boc[i]), W[f"down_blocks.{i}.downsamplers.0.conv."],
                             self._buf(f"ds{i}", (B * ho * wo, boc[i])),
                             conv=dict(B=B, Hi=hh, Wi=ww, Ho=ho, Wo=wo, stride=2), defer_reduce=True)   # -> the next block's norm1
                hh, ww = ho, wo
                skips.append((skip_of(x), hh, ww))
        all_skips = list(skips) if taps is not None else None
        # ---- 4. mid (ref :775-783)
        x = resnet("mid_block.resnets.0.", x, None, hh * ww, hh, ww, "r", gn_next=True)
        x = transformer("mid_block.attentions.0.", x, hh * ww, "r2")
        x = resnet("mid_block.resnets.1.", x, None, hh * ww, hh, ww, "r", gn_next=True)
        # ---- 5. up (ref :789-814)
        rev = list(reversed(boc))
        for i, typ in enumerate(cfg.up_block_types):
            for j in range(L_ + 1):
                sk, sh, sw = skips.pop()
                assert (sh, sw) == (hh, ww)
                cross = typ == "CrossAttnUpBlock2D"
                # next consumer: the transformer's norm, the next resnet's norm1, conv_norm_out -- or the upsampling conv (no norm)
                x = resnet(f"up_blocks.{i}.resnets.{j}.", x, sk, hh * ww, hh, ww, "r" if (j + i) % 2 else "rb",
                           gn_next=cross or j < L_ or i == nlev - 1)
                if cross:
                    x = transformer(f"up_blocks.{i}.attentions.{j}.", x, hh * ww, "u" if j % 2 else "ub")
                    if taps is not None:
                        taps[f"up{i}.{j}"] = x.clone()
            if i != nlev - 1:
                ho, wo = skips[-1][1], skips[-1][2]      # = (2 hh, 2 ww) unless a down conv rounded an odd size up
                w4 = W.get(f"up_blocks.{i}.upsamplers.0.conv4.")
                if w4 is not None and (ho, wo) == (2 * hh, 2 * ww):   # the phase decomposition (PHASE_UPSAMPLE above): 4/9 of the FLOPs + a shuffle
                    ph = ops.gemm(x.view(B, hh, ww, rev[i]), w4, self._buf("usp", (B * hh * ww, 4 * rev[i])),
                                  conv=dict(B=B, Hi=hh, Wi=ww, Ho=hh, Wo=ww), tap_lut=ops.UPSAMPLE_TAP_LUT, tap_group_n=rev[i])
                    x = ops.pixel_shuffle2(ph, self._buf("us", (B * ho * wo, rev[i])), B, hh, ww, rev[i])
                    hh, ww = ho, wo
                    continue
                x = ops.gemm(x.view(B, hh, ww, rev[i]), W[f"up_blocks.{i}.upsamplers.0.conv."],
                             self._buf("us", (B * ho * wo, rev[i])),
                             conv=dict(B=B, Hi=hh, Wi=ww, Ho=ho, Wo=wo, upsample=1), defer_reduce=True)    # -> the next block's norm1
                hh, ww = ho, wo
        # ---- 6. post-process (ref :817-820)
        ws = self._buf("gnws", (int(ops._lib.lib().pcdm_groupnorm_ws_floats(B, 4096)),), torch.float32, zero=True)
        n = ops.groupnorm(x, None, B, HW, G, eps, W["norm_out"][0], W["norm_out"][1], True, self._buf("gn", (B * HW, boc[0])), ws)
        out = self._buf("eps", (B, cfg.out_channels, h, w), torch.float32)
        ops.gemm(n, W["conv_out"], out, conv=dict(B=B, Hi=h, Wi=w, Ho=h, Wo=w), rows_per_batch=HW,
                 epilogue=ops.EPI_NCHW_F32)
        if taps is not None:
            # the down-path skip tensors, read back AFTER the forward: a deferred split-K tensor's buffer is filled by the norm that
            # consumes it, and every skip buffer stays untouched until here.  Names as oracle.unet.unet_forward's taps.
            names = ["conv_in"]
            for i in range(nlev):
                names += [f"down{i}.{j}" for j in range(L_)] + ([f"ds{i}"] if i != nlev - 1 else [])
            for nm_, (t_, _, _) in zip(names, all_skips):
                taps[nm_] = t_.clone()
        return out


def _emu() -> bool:
    from . import _lib
    return _lib.is_emulator()


# ---------------------------------------------------------------------- topology walkers
def _resnets(m: Stage2_InapintUNet2DConditionModel) -> Iterator[Tuple[str, int, int, int]]:
    """(prefix, cin, cout, level) of every ResnetBlock2D, following ref ctor :314-431."""
    boc, L = m._boc, m._layers[0]
    out = boc[0]
    for i in range(len(boc)):
        cin, out = out, boc[i]
        for j in range(L):
            yield f"down_blocks.{i}.resnets.{j}.", (cin if j == 0 else out), out, i
    yield "mid_block.resnets.0.", boc[-1], boc[-1], len(boc) - 1
    yield "mid_block.resnets.1.", boc[-1], boc[-1], len(boc) - 1
    rev = list(reversed(boc))
    out = rev[0]
    for i in range(len(boc)):
        prev, out = out, rev[i]
        inc = rev[min(i + 1, len(boc) - 1)]
        for j in range(L + 1):
            skip = inc if j == L else out
            rin = prev if j == 0 else out
            yield f"up_blocks.{i}.resnets.{j}.", rin + skip, out, len(boc) - 1 - i


def _transformers(m: Stage2_InapintUNet2DConditionModel) -> Iterator[Tuple[str, int, int]]:
    boc, L = m._boc, m._layers[0]
    for i, typ in enumerate(m.config.down_block_types):
        if typ == "CrossAttnDownBlock2D":
            for j in range(L):
                yield f"down_blocks.{i}.attentions.{j}.", boc[i], m._heads[i]
        elif typ != "DownBlock2D":
            raise NotImplementedError(typ)
    yield "mid_block.attentions.0.", boc[-1], m._heads[-1]
    rev, rh = list(reversed(boc)), list(reversed(m._heads))
    for i, typ in enumerate(m.config.up_block_types):
        if typ == "CrossAttnUpBlock2D":
            for j in range(L + 1):
                yield f"up_blocks.{i}.attentions.{j}.", rev[i], rh[i]
        elif typ != "UpBlock2D":
            raise NotImplementedError(typ)


def _param_shapes(m: Stage2_InapintUNet2DConditionModel) -> Iterator[Tuple[str, Tuple[int, ...]]]:
    """diffusers state-dict names and shapes."""
    cfg, boc = m.config, m._boc
    temb, ctx = boc[0] * 4, cfg.cross_attention_dim
    yield "conv_in.weight", (boc[0], cfg.in_channels, 3, 3)
    yield "conv_in.bias", (boc[0],)
    for nm, din in (("time_embedding", boc[0]),) + ((("class_embedding", cfg.projection_class_embeddings_input_dim),)
                                                    if cfg.class_embed_type == "projection" else ()):
        yield f"{nm}.linear_1.weight", (temb, din)
        yield f"{nm}.linear_1.bias", (temb,)
        yield f"{nm}.linear_2.weight", (temb, temb)
        yield f"{nm}.linear_2.bias", (temb,)
    for p, cin, cout, _ in _resnets(m):
        yield p + "norm1.weight", (cin,)
        yield p + "norm1.bias", (cin,)
        yield p + "conv1.weight", (cout, cin, 3, 3)
        yield p + "conv1.bias", (cout,)
        yield p + "time_emb_proj.weight", (cout, temb)
        yield p + "time_emb_proj.bias", (cout,)
        yield p + "norm2.weight", (cout,)
        yield p + "norm2.bias", (cout,)
        yield p + "conv2.weight", (cout, cout, 3, 3)
        yield p + "conv2.bias", (cout,)
        if cin != cout:
            yield p + "conv_shortcut.weight", (cout, cin, 1, 1)
            yield p + "conv_shortcut.bias", (cout,)
    lin = (lambda o, i: (o, i)) if cfg.use_linear_projection else (lambda o, i: (o, i, 1, 1))
    for p, c, _ in _transformers(m):
        yield p + "norm.weight", (c,)
        yield p + "norm.bias", (c,)
        yield p + "proj_in.weight", lin(c, c)
        yield p + "proj_in.bias", (c,)
        b = p + "transformer_blocks.0."
        for i, kd in ((1, c), (2, ctx)):
            yield b + f"norm{i}.weight", (c,)
            yield b + f"norm{i}.bias", (c,)
            yield b + f"attn{i}.to_q.weight", (c, c)
            yield b + f"attn{i}.to_k.weight", (c, kd)
            yield b + f"attn{i}.to_v.weight", (c, kd)
            yield b + f"attn{i}.to_out.0.weight", (c, c)
            yield b + f"attn{i}.to_out.0.bias", (c,)
        yield b + "norm3.weight", (c,)
        yield b + "norm3.bias", (c,)
        yield b + "ff.net.0.proj.weight", (8 * c, c)
        yield b + "ff.net.0.proj.bias", (8 * c,)
        yield b + "ff.net.2.weight", (c, 4 * c)
        yield b + "ff.net.2.bias", (c,)
        yield p + "proj_out.weight", lin(c, c)
        yield p + "proj_out.bias", (c,)
    for i in range(len(boc) - 1):
        yield f"down_blocks.{i}.downsamplers.0.conv.weight", (boc[i], boc[i], 3, 3)
        yield f"down_blocks.{i}.downsamplers.0.conv.bias", (boc[i],)
    rev = list(reversed(boc))
    for i in range(len(boc) - 1):
        yield f"up_blocks.{i}.upsamplers.0.conv.weight", (rev[i], rev[i], 3, 3)
        yield f"up_blocks.{i}.upsamplers.0.conv.bias", (rev[i],)
    yield "conv_norm_out.weight", (boc[0],)
    yield "conv_norm_out.bias", (boc[0],)
    yield "conv_out.weight", (cfg.out_channels, boc[0], 3, 3)
    yield "conv_out.bias", (cfg.out_channels,)


# friendlier alias
Stage2InpaintUNet = Stage2_InapintUNet2DConditionModel


class UNet2DConditionModel(Stage2_InapintUNet2DConditionModel):
    """The stock diffusers ``UNet2DConditionModel`` topology on the same kernels (no pose feature): the stage-3
    refinement UNet of the reference (``in_channels=8``, stage3_batchtest_refined_model.py:121-126)."""

    _pose_required = False
