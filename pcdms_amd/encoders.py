"""Conditioning image encoder on the HIP kernels: ``Dinov2Model``.

The stage-2 driver runs ``image_encoder_p = Dinov2Model.from_pretrained(path)`` (transformers, DINOv2-giant) once per pair:
``s_img_f = image_encoder_p(clip_processor_s_img).last_hidden_state`` -> [1, 257, 1536]
(/root/reference/stage2_batchtest_inpaint_model.py:96,165-166), which ``ImageProjModel_p`` then maps to the UNet's
context tokens.  This class mirrors that surface -- ``from_pretrained`` (HF ``config.json`` + ``model.safetensors`` /
``pytorch_model.bin``), ``load_state_dict`` with transformers' key names, ``.to``, ``__call__(pixel_values)`` returning an
object with ``last_hidden_state`` / ``pooler_output`` -- as a schedule of libpcdm.so calls:

* patch embedding = one GEMM over the 14x14 patches (K = 588 zero-padded to 640), its epilogue adds the (interpolated)
  position embedding as a row-broadcast residual and writes straight into the token buffer;
* per layer: LayerNorm -> fused QKV GEMM (V written transposed) -> d = 64 flash attention -> output GEMM (+residual) ->
  LayerNorm -> SwiGLU GEMM (gate activation in the epilogue; plain GELU MLP for the non-giant variants) -> GEMM (+residual);
  the two LayerScale vectors are folded into the output / down-projection weights at pack time (exact);
* final LayerNorm.

Parity oracle: ``transformers.Dinov2Model`` itself (fp32, CPU), which IS the library the reference calls
(tests/test_encoders.py) -- this row's parity is pinned to the real third-party implementation, unlike the diffusers blocks.
The position-embedding interpolation (a load-time, weights-only transform) follows the installed transformers
(``size=`` bicubic); transformers 4.32.1, which the reference pins, used ``scale_factor=(h + 0.1) / sqrt(N)``: select it with
``pos_interp="scale_factor"``.
"""
from __future__ import annotations

import json
import math
from pathlib import Path
from types import SimpleNamespace
from typing import Any, Dict, Optional, Tuple

import torch

from . import ops
from .cond import _HipModule
from .ops import BF16

DINOV2_GIANT_CONFIG = dict(hidden_size=1536, num_hidden_layers=40, num_attention_heads=24, mlp_ratio=4, image_size=518,
                           patch_size=14, num_channels=3, layer_norm_eps=1e-6, qkv_bias=True, use_swiglu_ffn=True,
                           layerscale_value=1.0, hidden_act="gelu")


class BaseModelOutputWithPooling:
    def __init__(self, last_hidden_state, pooler_output):
        self.last_hidden_state, self.pooler_output = last_hidden_state, pooler_output

    def __getitem__(self, i):
        return (self.last_hidden_state, self.pooler_output)[i]


class Dinov2Model(_HipModule):
    _name = "Dinov2Model"

    def __init__(self, config: Optional[Any] = None, pos_interp: str = "size", **kwargs):
        super().__init__()
        cfg = dict(DINOV2_GIANT_CONFIG)
        src = {} if config is None else (config if isinstance(config, dict) else
                                         (config.to_dict() if hasattr(config, "to_dict") else dict(vars(config))))
        cfg.update({k: v for k, v in {**src, **kwargs}.items() if k in cfg})
        self.config = SimpleNamespace(**cfg)
        c = self.config
        if c.hidden_size % c.num_attention_heads or c.hidden_size // c.num_attention_heads != 64:
            raise NotImplementedError("the attention kernel is specialised for head_dim 64 (all DINOv2 sizes)")
        if c.hidden_size % 64 or not c.qkv_bias or (not c.use_swiglu_ffn and c.hidden_act != "gelu"):
            raise NotImplementedError("unsupported DINOv2 variant")
        if pos_interp not in ("size", "scale_factor"):
            raise ValueError("pos_interp must be 'size' or 'scale_factor'")
        self.pos_interp = pos_interp
        self._pos_cache: Dict[Tuple[int, int], torch.Tensor] = {}

    @property
    def ffn_dim(self) -> int:
        c = self.config
        hidden = int(c.hidden_size * c.mlp_ratio)
        return (int(hidden * 2 / 3) + 7) // 8 * 8 if c.use_swiglu_ffn else hidden

    def expected_shapes(self) -> Dict[str, Tuple[int, ...]]:
        c, D, F = self.config, self.config.hidden_size, self.ffn_dim
        npos = (c.image_size // c.patch_size) ** 2 + 1
        exp: Dict[str, Tuple[int, ...]] = {
            "embeddings.cls_token": (1, 1, D), "embeddings.mask_token": (1, D), "embeddings.position_embeddings": (1, npos, D),
            "embeddings.patch_embeddings.projection.weight": (D, c.num_channels, c.patch_size, c.patch_size),
            "embeddings.patch_embeddings.projection.bias": (D,)}
        for i in range(c.num_hidden_layers):
            p = f"encoder.layer.{i}."
            for n in ("norm1", "norm2"):
                exp[p + n + ".weight"], exp[p + n + ".bias"] = (D,), (D,)
            for n in ("query", "key", "value"):
                exp[p + f"attention.attention.{n}.weight"], exp[p + f"attention.attention.{n}.bias"] = (D, D), (D,)
            exp[p + "attention.output.dense.weight"], exp[p + "attention.output.dense.bias"] = (D, D), (D,)
            exp[p + "layer_scale1.lambda1"], exp[p + "layer_scale2.lambda1"] = (D,), (D,)
            if c.use_swiglu_ffn:
                exp[p + "mlp.weights_in.weight"], exp[p + "mlp.weights_in.bias"] = (2 * F, D), (2 * F,)
                exp[p + "mlp.weights_out.weight"], exp[p + "mlp.weights_out.bias"] = (D, F), (D,)
            else:
                exp[p + "mlp.fc1.weight"], exp[p + "mlp.fc1.bias"] = (F, D), (F,)
                exp[p + "mlp.fc2.weight"], exp[p + "mlp.fc2.bias"] = (D, F), (D,)
        exp["layernorm.weight"], exp["layernorm.bias"] = (D,), (D,)
        return exp

    def load_state_dict(self, state_dict, strict: bool = True):
        self._pos_cache.clear()
        return super().load_state_dict(state_dict, strict)

    def to(self, *args, **kwargs):
        self._pos_cache.clear()
        return super().to(*args, **kwargs)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, **kwargs):
        d = Path(str(pretrained_model_name_or_path))
        cfg = {}
        if (d / "config.json").exists():
            cfg = json.loads((d / "config.json").read_text())
        m = cls(cfg, **kwargs)
        sd = None
        if (d / "model.safetensors").exists():
            from safetensors.torch import load_file
            sd = load_file(str(d / "model.safetensors"))
        elif (d / "pytorch_model.bin").exists():
            sd = torch.load(str(d / "pytorch_model.bin"), map_location="cpu")
        if sd is None:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {d}")
        m.load_state_dict({k: v for k, v in sd.items() if k in m.expected_shapes()})
        return m

    # ---------------------------------------------------------------- packing
    def _pack(self):
        self._ready()
        sd, dev, c, D, F = self._sd, self._device, self.config, self.config.hidden_size, self.ffn_dim

        def f32(k):
            return sd[k].to(dev, torch.float32).contiguous()
        kp = c.num_channels * c.patch_size ** 2
        wpe = torch.zeros(D, ops._round_up(kp, 64))
        wpe[:, :kp] = sd["embeddings.patch_embeddings.projection.weight"].reshape(D, kp)
        w: Dict[str, Any] = dict(patch=ops.pack_linear(wpe, sd["embeddings.patch_embeddings.projection.bias"], dev), kp=kp,
                                 norm=(f32("layernorm.weight"), f32("layernorm.bias")), layers=[])
        Fp = ops._round_up(F, 64)
        for i in range(c.num_hidden_layers):
            p = f"encoder.layer.{i}."
            a = p + "attention.attention."
            g1, g2 = sd[p + "layer_scale1.lambda1"], sd[p + "layer_scale2.lambda1"]
            L = dict(n1=(f32(p + "norm1.weight"), f32(p + "norm1.bias")), n2=(f32(p + "norm2.weight"), f32(p + "norm2.bias")),
                     qkv=ops.pack_linear(torch.cat([sd[a + "query.weight"], sd[a + "key.weight"], sd[a + "value.weight"]], 0),
                                         torch.cat([sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"]], 0), dev),
                     # LayerScale folded in: gamma * (W x + b) == (diag(gamma) W) x + gamma * b
                     o=ops.pack_linear(g1[:, None] * sd[p + "attention.output.dense.weight"], g1 * sd[p + "attention.output.dense.bias"], dev))
            if c.use_swiglu_ffn:   # hidden = silu(x1) * x2 with (x1, x2) = weights_in(x).chunk(2): gate = x1, linear half = x2
                wi, bi = sd[p + "mlp.weights_in.weight"], sd[p + "mlp.weights_in.bias"]
                L["glu"] = ops.pack_geglu(torch.cat([wi[F:], wi[:F]], 0), torch.cat([bi[F:], bi[:F]], 0), dev)
                wo = torch.zeros(D, Fp)
                wo[:, :F] = g2[:, None] * sd[p + "mlp.weights_out.weight"]
                L["down"] = ops.pack_linear(wo, g2 * sd[p + "mlp.weights_out.bias"], dev)
            else:
                L["fc1"] = ops.pack_linear(sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], dev)
                L["down"] = ops.pack_linear(g2[:, None] * sd[p + "mlp.fc2.weight"], g2 * sd[p + "mlp.fc2.bias"], dev)
            w["layers"].append(L)
        self._w = w

    def _pos(self, gh: int, gw: int) -> torch.Tensor:
        """bf16 [1 + gh*gw, D] on the device: cls_token + pos[0] in row 0, interpolated patch position embeddings below.
        Load-time, weights-only transform (transformers ``Dinov2Embeddings.interpolate_pos_encoding``)."""
        key = (gh, gw)
        if key not in self._pos_cache:
            pos = self._sd["embeddings.position_embeddings"][0]
            D = pos.shape[-1]
            n = pos.shape[0] - 1
            s = int(round(math.sqrt(n)))
            patch = pos[1:]
            if (gh, gw) != (s, s):
                grid = patch.reshape(1, s, s, D).permute(0, 3, 1, 2).float()
                if self.pos_interp == "size":
                    grid = torch.nn.functional.interpolate(grid, size=(gh, gw), mode="bicubic", align_corners=False)
                else:   # transformers 4.32.1
                    grid = torch.nn.functional.interpolate(grid, scale_factor=((gh + 0.1) / s, (gw + 0.1) / s), mode="bicubic",
                                                           align_corners=False)
                    assert grid.shape[-2:] == (gh, gw)
                patch = grid.permute(0, 2, 3, 1).reshape(gh * gw, D)
            row0 = self._sd["embeddings.cls_token"][0, 0] + pos[0]
            self._pos_cache[key] = torch.cat([row0[None], patch], 0).to(BF16).to(self._device).contiguous()
        return self._pos_cache[key]

    def _buf(self, name, shape, dtype=BF16, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self._device)
        return t

    # ---------------------------------------------------------------- forward
    @torch.no_grad()
    def __call__(self, pixel_values: torch.Tensor, bool_masked_pos=None, head_mask=None, output_attentions=None,
                 output_hidden_states=None, return_dict=True):
        if bool_masked_pos is not None or head_mask is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("only the plain forward the reference uses (stage2_batchtest_inpaint_model.py:166)")
        if self._w is None:
            self._pack()
        w, c, D = self._w, self.config, self.config.hidden_size
        B, Cc, Hh, Ww = pixel_values.shape
        P = c.patch_size
        if Cc != c.num_channels or Hh % P or Ww % P:
            raise ValueError(f"pixel_values must be [B,{c.num_channels},H,W] with H, W multiples of {P}")
        gh, gw = Hh // P, Ww // P
        T = 1 + gh * gw
        M = B * T
        pos = self._pos(gh, gw)
        x = self._buf("tok_a", (M, D))
        # patches -> rows (pure data movement), zero-padded K, bf16
        kp, Kp = w["kp"], w["patch"].K
        pv = pixel_values.to(self._device, torch.float32)
        cols = pv.unfold(2, P, P).unfold(3, P, P).permute(0, 2, 3, 1, 4, 5).reshape(B, gh * gw, kp)
        colp = self._buf("cols32", (B, gh * gw, Kp), torch.float32, zero=True)
        colp[:, :, :kp] = cols
        colb = ops.f32_to_bf16(colp, self._buf("cols", (B, gh * gw, Kp)))
        xv = x.view(B, T, D)
        for b in range(B):   # epilogue: + bias + position embedding row (m % (gh*gw)), written below the CLS row
            ops.gemm(colb[b], w["patch"], xv[b, 1:], residual=pos[1:], res_mod=gh * gw)
            xv[b, 0].copy_(pos[0])
        H = c.num_attention_heads
        Tp = (T + 7) // 8 * 8
        Fp = w["layers"][0]["down"].K
        for L in w["layers"]:
            n = ops.layernorm(x, L["n1"][0], L["n1"][1], c.layer_norm_eps, self._buf("ln", (M, D)))
            qk = self._buf("qk", (M, 2 * D))
            vt = self._buf("vt", (B, D, Tp), zero=True)
            ops.gemm(n, L["qkv"], qk, rows_per_batch=T, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * D)
            at = ops.flash_attn(qk[:, :D], qk[:, D:], vt, self._buf("at", (M, D)), B, H, T, T)
            x1 = ops.gemm(at, L["o"], self._buf("tok_b", (M, D)), residual=x, res_mod=M)
            n = ops.layernorm(x1, L["n2"][0], L["n2"][1], c.layer_norm_eps, self._buf("ln", (M, D)))
            if "glu" in L:
                f = ops.gemm(n, L["glu"], self._buf("ff", (M, Fp), zero=True), epilogue=ops.EPI_GEGLU, act=ops.ACT_SILU)
            else:
                f = ops.gemm(n, L["fc1"], self._buf("ff", (M, Fp)), act=ops.ACT_GELU)
            x = ops.gemm(f, L["down"], self._buf("tok_a", (M, D)), residual=x1, res_mod=M)
        out = ops.layernorm(x, w["norm"][0], w["norm"][1], c.layer_norm_eps, self._buf("out", (M, D)))
        hs = out.view(B, T, D).to(pixel_values.dtype if pixel_values.dtype.is_floating_point else torch.float32)
        res = BaseModelOutputWithPooling(hs, hs[:, 0])
        return res if return_dict else (res.last_hidden_state, res.pooler_output)

    forward = __call__


# ======================================================================================================================
CLIP_VIT_H14_CONFIG = dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16,
                           image_size=224, patch_size=14, num_channels=3, hidden_act="gelu", layer_norm_eps=1e-5,
                           projection_dim=1024)


class CLIPVisionModelOutput:
    def __init__(self, image_embeds, last_hidden_state):
        self.image_embeds, self.last_hidden_state = image_embeds, last_hidden_state

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else (self.image_embeds, self.last_hidden_state)[k]


class CLIPVisionModelWithProjection(_HipModule):
    """``image_encoder_g`` / ``image_encoder`` of the drivers (OpenCLIP ViT-H/14 through transformers):
    ``image_encoder(pixel_values).image_embeds`` -> [B, 1024] (/root/reference/stage1_batchtest_prior_model.py:62,96-98;
    stage2_batchtest_inpaint_model.py:95,178).  Same kernels as ``Dinov2Model``; the head dimension is 80, which the d = 64
    flash kernel does not cover, so attention runs as per-head MFMA GEMMs around one row-softmax launch per layer
    (K Q^T in fp32 -> ``pcdm_softmax_rows`` -> P V), with q / k emitted in a 128-wide zero-padded per-head layout by the
    fused QKV GEMM (the padding lives in the packed weights).  Parity oracle: ``transformers.CLIPVisionModelWithProjection``."""

    _name = "CLIPVisionModelWithProjection"

    def __init__(self, config: Optional[Any] = None, **kwargs):
        super().__init__()
        cfg = dict(CLIP_VIT_H14_CONFIG)
        src = {} if config is None else (config if isinstance(config, dict) else
                                         (config.to_dict() if hasattr(config, "to_dict") else dict(vars(config))))
        src = dict(src.get("vision_config", src))
        cfg.update({k: v for k, v in {**src, **kwargs}.items() if k in cfg})
        self.config = SimpleNamespace(**cfg)
        c = self.config
        if c.hidden_size % 64 or c.intermediate_size % 64 or c.hidden_size % c.num_attention_heads or c.projection_dim % 4:
            raise NotImplementedError("hidden / intermediate sizes must be multiples of 64")
        if (c.hidden_size // c.num_attention_heads) % 8:
            raise NotImplementedError("head_dim must be a multiple of 8")
        if c.hidden_act != "gelu":
            raise NotImplementedError("hidden_act: only 'gelu' (OpenCLIP ViT-H/14); quick_gelu is not implemented")
        if c.image_size % c.patch_size:
            raise ValueError("image_size must be a multiple of patch_size")

    def expected_shapes(self) -> Dict[str, Tuple[int, ...]]:
        c, D, F = self.config, self.config.hidden_size, self.config.intermediate_size
        T = (c.image_size // c.patch_size) ** 2 + 1
        v = "vision_model."
        exp: Dict[str, Tuple[int, ...]] = {
            v + "embeddings.class_embedding": (D,), v + "embeddings.patch_embedding.weight": (D, c.num_channels, c.patch_size, c.patch_size),
            v + "embeddings.position_embedding.weight": (T, D), v + "pre_layrnorm.weight": (D,), v + "pre_layrnorm.bias": (D,)}
        for i in range(c.num_hidden_layers):
            p = v + f"encoder.layers.{i}."
            for n in ("layer_norm1", "layer_norm2"):
                exp[p + n + ".weight"], exp[p + n + ".bias"] = (D,), (D,)
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                exp[p + f"self_attn.{n}.weight"], exp[p + f"self_attn.{n}.bias"] = (D, D), (D,)
            exp[p + "mlp.fc1.weight"], exp[p + "mlp.fc1.bias"] = (F, D), (F,)
            exp[p + "mlp.fc2.weight"], exp[p + "mlp.fc2.bias"] = (D, F), (D,)
        exp[v + "post_layernorm.weight"], exp[v + "post_layernorm.bias"] = (D,), (D,)
        exp["visual_projection.weight"] = (c.projection_dim, D)
        return exp

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}   # buffer in older checkpoints
        return super().load_state_dict(sd, strict)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, **kwargs):
        d = Path(str(pretrained_model_name_or_path))
        cfg = json.loads((d / "config.json").read_text()) if (d / "config.json").exists() else {}
        m = cls(cfg, **kwargs)
        sd = None
        if (d / "model.safetensors").exists():
            from safetensors.torch import load_file
            sd = load_file(str(d / "model.safetensors"))
        elif (d / "pytorch_model.bin").exists():
            sd = torch.load(str(d / "pytorch_model.bin"), map_location="cpu")
        if sd is None:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {d}")
        exp = m.expected_shapes()
        m.load_state_dict({k: v for k, v in sd.items() if k in exp})   # a full CLIPModel checkpoint also holds the text tower
        return m

    def _pack(self):
        self._ready()
        sd, dev, c, D = self._sd, self._device, self.config, self.config.hidden_size
        H = c.num_attention_heads
        dh = D // H
        dhp = ops._round_up(dh, 64)
        v = "vision_model."

        def f32(k):
            return sd[k].to(dev, torch.float32).contiguous()

        def ln(p):
            return f32(p + ".weight"), f32(p + ".bias")

        def lin(p):
            return ops.pack_linear(sd[p + ".weight"], sd[p + ".bias"], dev)

        def pad_heads(wt, bs):   # [H*dh, D] -> [H*dhp, D] with zero rows after each head's dh rows
            wp, bp = torch.zeros(H, dhp, D), torch.zeros(H, dhp)
            wp[:, :dh], bp[:, :dh] = wt.reshape(H, dh, D), bs.reshape(H, dh)
            return wp.reshape(H * dhp, D), bp.reshape(H * dhp)
        kp = c.num_channels * c.patch_size ** 2
        wpe = torch.zeros(D, ops._round_up(kp, 64))
        wpe[:, :kp] = sd[v + "embeddings.patch_embedding.weight"].reshape(D, kp)
        pos = sd[v + "embeddings.position_embedding.weight"]
        w: Dict[str, Any] = dict(patch=ops.pack_linear(wpe, None, dev), kp=kp, dh=dh, dhp=dhp,
                                 pos=torch.cat([(sd[v + "embeddings.class_embedding"] + pos[0])[None], pos[1:]], 0).to(BF16).to(dev).contiguous(),
                                 pre=ln(v + "pre_layrnorm"), post=ln(v + "post_layernorm"),
                                 proj=ops.pack_linear(sd["visual_projection.weight"], None, dev), layers=[])
        for i in range(c.num_hidden_layers):
            p = v + f"encoder.layers.{i}."
            a = p + "self_attn."
            if dh == 64:
                wq, bq, wk, bk = sd[a + "q_proj.weight"], sd[a + "q_proj.bias"], sd[a + "k_proj.weight"], sd[a + "k_proj.bias"]
            else:
                wq, bq = pad_heads(sd[a + "q_proj.weight"], sd[a + "q_proj.bias"])
                wk, bk = pad_heads(sd[a + "k_proj.weight"], sd[a + "k_proj.bias"])
            w["layers"].append(dict(n1=ln(p + "layer_norm1"), n2=ln(p + "layer_norm2"),
                                    qkv=ops.pack_linear(torch.cat([wq, wk, sd[a + "v_proj.weight"]], 0),
                                                        torch.cat([bq, bk, sd[a + "v_proj.bias"]], 0), dev),
                                    o=lin(a + "out_proj"), fc1=lin(p + "mlp.fc1"), fc2=lin(p + "mlp.fc2")))
        self._w = w

    def _buf(self, name, shape, dtype=BF16, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self._device)
        return t

    def _attention(self, n, L, B, T):
        """softmax(q k^T / sqrt(dh)) v for every (image, head); returns [B*T, D]."""
        w, c, D = self._w, self.config, self.config.hidden_size
        H, dh, dhp = c.num_attention_heads, w["dh"], w["dhp"]
        M = B * T
        QW = H * dhp
        qk = self._buf("qk", (M + 64, 2 * QW), zero=True)[:M + 4]       # zero tail rows: read (never used) by the padded N-tiles
        at = self._buf("at", (M, D))
        if dh == 64:
            vt = self._buf("vt", (B, D, (T + 7) // 8 * 8), zero=True)
            ops.gemm(n, L["qkv"], qk[:M], rows_per_batch=T, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * QW)
            return ops.flash_attn(qk[:M, :QW], qk[:M, QW:], vt, at, B, H, T, T)
        Tk = ops._round_up(T, 64)                 # key axis padded to the GEMM's K granularity (zero columns)
        Tq = ops._round_up(T, 4)                  # query axis padded to the GEMM's N granularity (rows >= T are never read back)
        flat = self._buf("vt", (B * D * Tk + 64 * Tk,), zero=True)      # tail: the last head's N-tile reads past its dh rows
        vt = flat[:B * D * Tk].view(B, D, Tk)
        ops.gemm(n, L["qkv"], qk[:M], rows_per_batch=T, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * QW)
        S = self._buf("S", (B * H, Tq, T), torch.float32)
        P = self._buf("P", (B * H, Tq, Tk), zero=True)                  # columns >= T stay zero
        for b in range(B):
            rows = slice(b * T, (b + 1) * T)
            for h in range(H):   # S[q, k] = sum_d K[k, d] Q[q, d]: "A" = K_h, "weights" = Q_h (row stride = qk row), stored [q][k] fp32
                k = qk[rows, QW + h * dhp:QW + (h + 1) * dhp]
                q = qk[b * T:b * T + Tq, h * dhp:(h + 1) * dhp]         # Tq - T rows of the next image / of the zero tail
                wq = ops.PackedWeight(q, None, Tq, dhp, ops._round_up(Tq, 64), alg_nk=T * dh)
                ops.gemm(k, wq, S[b * H + h], rows_per_batch=T, epilogue=ops.EPI_NCHW_F32, w_ld=2 * QW)
        ops.softmax_rows(S.view(B * H * Tq, T), P.view(B * H * Tq, Tk), dh ** -0.5)
        for b in range(B):
            for h in range(H):
                wv = ops.PackedWeight(vt[b, h * dh:], None, dh, Tk, ops._round_up(dh, 64), alg_nk=dh * T)   # V_h^T [dh, Tk]
                ops.gemm(P[b * H + h, :T], wv, at[b * T:(b + 1) * T, h * dh:(h + 1) * dh])
        return at

    @torch.no_grad()
    def __call__(self, pixel_values: torch.Tensor, output_attentions=None, output_hidden_states=None, return_dict=True):
        if output_attentions or output_hidden_states:
            raise NotImplementedError("only the plain forward the reference uses")
        if self._w is None:
            self._pack()
        w, c, D = self._w, self.config, self.config.hidden_size
        B, Cc, Hh, Ww = pixel_values.shape
        P_ = c.patch_size
        if (Cc, Hh, Ww) != (c.num_channels, c.image_size, c.image_size):
            raise ValueError(f"pixel_values must be [B,{c.num_channels},{c.image_size},{c.image_size}]")
        g = Hh // P_
        T = 1 + g * g
        M = B * T
        kp, Kp = w["kp"], w["patch"].K
        pv = pixel_values.to(self._device, torch.float32)
        cols = pv.unfold(2, P_, P_).unfold(3, P_, P_).permute(0, 2, 3, 1, 4, 5).reshape(B, g * g, kp)   # data movement only
        colp = self._buf("cols32", (B, g * g, Kp), torch.float32, zero=True)
        colp[:, :, :kp] = cols
        colb = ops.f32_to_bf16(colp, self._buf("cols", (B, g * g, Kp)))
        e = self._buf("emb", (M, D))
        ev = e.view(B, T, D)
        for b in range(B):
            ops.gemm(colb[b], w["patch"], ev[b, 1:], residual=w["pos"][1:], res_mod=g * g)
            ev[b, 0].copy_(w["pos"][0])
        x = ops.layernorm(e, w["pre"][0], w["pre"][1], c.layer_norm_eps, self._buf("tok_a", (M, D)))
        for L in w["layers"]:
            n = ops.layernorm(x, L["n1"][0], L["n1"][1], c.layer_norm_eps, self._buf("ln", (M, D)))
            at = self._attention(n, L, B, T)
            x1 = ops.gemm(at, L["o"], self._buf("tok_b", (M, D)), residual=x, res_mod=M)
            n = ops.layernorm(x1, L["n2"][0], L["n2"][1], c.layer_norm_eps, self._buf("ln", (M, D)))
            f = ops.gemm(n, L["fc1"], self._buf("ff", (M, c.intermediate_size)), act=ops.ACT_GELU)
            x = ops.gemm(f, L["fc2"], self._buf("tok_a", (M, D)), residual=x1, res_mod=M)
        cls_rows = self._buf("cls", (B, D))
        cls_rows.copy_(x.view(B, T, D)[:, 0])                                       # gather of the CLS rows (copy)
        pooled = ops.layernorm(cls_rows, w["post"][0], w["post"][1], c.layer_norm_eps, self._buf("pooled", (B, D)))
        emb = torch.empty(B, c.projection_dim, dtype=torch.float32, device=self._device)
        ops.gemm(pooled, w["proj"], emb, rows_per_batch=1, epilogue=ops.EPI_NCHW_F32)   # fp32 [B, projection_dim]
        od = pixel_values.dtype if pixel_values.dtype.is_floating_point else torch.float32
        out = CLIPVisionModelOutput(emb.to(od), x.view(B, T, D).to(od))
        return out if return_dict else (out.image_embeds, out.last_hidden_state)

    forward = __call__
