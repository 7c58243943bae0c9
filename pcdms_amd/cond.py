"""The two small per-pair conditioning nets of the stage-2 driver, on the HIP kernels.

``ControlNetConditioningEmbedding`` replaces diffusers' class of the same name as the driver uses it
(/root/reference/stage2_batchtest_inpaint_model.py:101 ``ControlNetConditioningEmbedding(320, 3, (16, 32, 96, 256))``,
``load_state_dict(pose_proj_dict)`` :118, ``st_pose_f = pose_proj(cond_st_pose)`` :173-174): eight 3x3 convolutions
(three of them stride 2) with SiLU fused into the implicit-GEMM epilogue; channels are zero-padded to the 64-wide MFMA
tile at pack time (3/16/32 -> 64, 96 -> 128), the last conv writes fp32 NCHW, the layout ``st_pose_f`` has at the
pipeline boundary.  ``ImageProjModel_p`` replaces the module defined at :48-64 (``s_img_proj_f = image_proj_model_p(s_img_f)``
:167): Linear+GELU (one GEMM), LayerNorm, Linear.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Dict, Optional, Sequence, Tuple

import torch

from . import _lib, ops
from ._module import ModuleSurface
from .ops import BF16


class _HipModule(ModuleSurface):
    """Minimal nn.Module-like surface (to / eval / load_state_dict / state_dict) shared by the two nets."""

    _name = "module"

    def __init__(self):
        self._device = torch.device("cpu")
        self._sd: Optional[Dict[str, torch.Tensor]] = None
        self._w: Optional[Dict[str, Any]] = None
        self._bufs: Dict[Tuple, torch.Tensor] = {}

    def expected_shapes(self) -> Dict[str, Tuple[int, ...]]:
        raise NotImplementedError

    @property
    def device(self):
        return self._device

    def to(self, *args, **kwargs):
        device = kwargs.get("device")
        for a in args:
            if not isinstance(a, torch.dtype) and a is not None:
                device = torch.device(a)
        if device is not None and torch.device(device) != self._device:
            self._device = torch.device(device)
            if self._device.type == "cuda" and self._device.index is None:
                self._device = torch.device("cuda", torch.cuda.current_device())
            self._w = None
            self._bufs.clear()
        return self

    def state_dict(self):
        return dict(self._sd or {})

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        exp = self.expected_shapes()
        missing = [k for k in exp if k not in state_dict]
        unexpected = [k for k in state_dict if k not in exp]
        bad = [k for k in exp if k in state_dict and tuple(state_dict[k].shape) != tuple(exp[k])]
        if bad or (strict and (missing or unexpected)):
            raise RuntimeError(f"Error(s) in loading state_dict for {self._name}: missing {missing[:6]} unexpected "
                               f"{unexpected[:6]} size mismatch {bad[:6]}")
        self._sd = {k: state_dict[k].detach().to("cpu", torch.float32) for k in exp if k in state_dict}
        self._w = None
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def _ready(self):
        if self._sd is None:
            raise RuntimeError("weights not loaded")
        if self._device.type != "cuda" and not _lib.is_emulator():
            raise RuntimeError(f"{self._name} runs on the MI355X only: call .to('cuda')")

    def _buf(self, name, shape, dtype=BF16):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.empty(tuple(shape), dtype=dtype, device=self._device)
        return t


class ControlNetConditioningEmbedding(_HipModule):
    _name = "ControlNetConditioningEmbedding"

    def __init__(self, conditioning_embedding_channels: int = 320, conditioning_channels: int = 3,
                 block_out_channels: Sequence[int] = (16, 32, 96, 256)):
        super().__init__()
        self.out_channels, self.cond_channels, self.boc = conditioning_embedding_channels, conditioning_channels, tuple(block_out_channels)
        if self.out_channels % 4:
            raise NotImplementedError("conditioning_embedding_channels must be a multiple of 4")

    def expected_shapes(self):
        boc, exp = self.boc, {}
        exp["conv_in.weight"], exp["conv_in.bias"] = (boc[0], self.cond_channels, 3, 3), (boc[0],)
        for i in range(len(boc) - 1):
            exp[f"blocks.{2 * i}.weight"], exp[f"blocks.{2 * i}.bias"] = (boc[i], boc[i], 3, 3), (boc[i],)
            exp[f"blocks.{2 * i + 1}.weight"], exp[f"blocks.{2 * i + 1}.bias"] = (boc[i + 1], boc[i], 3, 3), (boc[i + 1],)
        exp["conv_out.weight"], exp["conv_out.bias"] = (self.out_channels, boc[-1], 3, 3), (self.out_channels,)
        return exp

    def _pack(self):
        self._ready()
        sd, dev = self._sd, self._device

        def conv(name, pad_out=True):
            w, b = sd[name + ".weight"], sd[name + ".bias"]
            if pad_out:    # output channels up to the 64-wide tile: the padding channels come out as silu(0) = 0
                n, npad = w.shape[0], ops._round_up(w.shape[0], 64)
                w = torch.cat([w, w.new_zeros(npad - n, *w.shape[1:])])
                b = torch.cat([b, b.new_zeros(npad - n)])
            return ops.pack_conv3x3(w, b, dev)
        nb = 2 * (len(self.boc) - 1)
        self._w = dict(conv_in=conv("conv_in"), blocks=[conv(f"blocks.{i}") for i in range(nb)], conv_out=conv("conv_out", False))

    @torch.no_grad()
    def __call__(self, conditioning: torch.Tensor) -> torch.Tensor:
        """conditioning [B, 3, H, W] (any float dtype) -> fp32 [B, C_out, H/8, W/8] on the device."""
        if self._w is None:
            self._pack()
        w = self._w
        B, Cc, H, W = conditioning.shape
        down = 2 ** (len(self.boc) - 1)
        if Cc != self.cond_channels or H % down or W % down:
            raise ValueError(f"conditioning must be [B,{self.cond_channels},H,W] with H, W multiples of {down}")
        x = ops.nchw_to_nhwc_bf16(conditioning.to(self._device), self._buf("in", (B, H, W, 64)), cpad=64)
        h = ops.gemm(x, w["conv_in"], self._buf("x0", (B * H * W, w["conv_in"].Npad)), conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W),
                     act=ops.ACT_SILU)
        for i, pw in enumerate(w["blocks"]):
            s = 2 if i % 2 else 1
            Ho, Wo = H // s, W // s
            h = ops.gemm(h, pw, self._buf(f"b{i}", (B * Ho * Wo, pw.Npad)), conv=dict(B=B, Hi=H, Wi=W, Ho=Ho, Wo=Wo, stride=s),
                         act=ops.ACT_SILU)
            H, W = Ho, Wo
        out = torch.empty(B, self.out_channels, H, W, dtype=torch.float32, device=self._device)
        ops.gemm(h, w["conv_out"], out, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), rows_per_batch=H * W, epilogue=ops.EPI_NCHW_F32)
        return out

    forward = __call__


class ImageProjModel_p(_HipModule):
    _name = "ImageProjModel_p"

    def __init__(self, in_dim: int = 1536, hidden_dim: int = 768, out_dim: int = 1024, dropout: float = 0.0):
        super().__init__()
        if in_dim % 64 or hidden_dim % 64 or out_dim % 4:
            raise NotImplementedError("in_dim / hidden_dim must be multiples of 64")
        self.in_dim, self.hidden_dim, self.out_dim = in_dim, hidden_dim, out_dim

    def expected_shapes(self):
        return {"net.0.weight": (self.hidden_dim, self.in_dim), "net.0.bias": (self.hidden_dim,),
                "net.3.weight": (self.hidden_dim,), "net.3.bias": (self.hidden_dim,),
                "net.4.weight": (self.out_dim, self.hidden_dim), "net.4.bias": (self.out_dim,)}

    def _pack(self):
        self._ready()
        sd, dev = self._sd, self._device
        self._w = dict(fc1=ops.pack_linear(sd["net.0.weight"], sd["net.0.bias"], dev),
                       ln=(sd["net.3.weight"].to(dev).contiguous(), sd["net.3.bias"].to(dev).contiguous()),
                       fc2=ops.pack_linear(sd["net.4.weight"], sd["net.4.bias"], dev))

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, L, in_dim] -> [B, L, out_dim] (bf16 compute, returned in x's dtype)."""
        if self._w is None:
            self._pack()
        w = self._w
        B, L, D = x.shape
        if D != self.in_dim:
            raise ValueError(f"expected last dim {self.in_dim}, got {D}")
        M = B * L
        xb = ops.f32_to_bf16(x.to(self._device, torch.float32).contiguous().view(M, D), self._buf("x", (M, D)))
        h = ops.gemm(xb, w["fc1"], self._buf("h", (M, self.hidden_dim)), act=ops.ACT_GELU)
        n = ops.layernorm(h, w["ln"][0], w["ln"][1], 1e-5, self._buf("n", (M, self.hidden_dim)))
        y = ops.gemm(n, w["fc2"], torch.empty(M, self.out_dim, dtype=BF16, device=self._device))
        return y.view(B, L, self.out_dim).to(x.dtype if x.dtype.is_floating_point else torch.float32)

    forward = __call__


class ImageProjection(_HipModule):
    """``ImageProjection`` of the notebook pipeline (ref src/pipelines/PCDMs_pipeline.py:154-173): Linear(E -> 2E) -> GELU ->
    Linear(2E -> num_tokens * D) -> reshape [-1, num_tokens, D] -> LayerNorm(D).  The same three kernels as ``ImageProjModel_p``."""
    _name = "ImageProjection"

    def __init__(self, cross_attention_dim: int = 768, clip_embeddings_dim: int = 512, num_tokens: int = 4):
        super().__init__()
        if clip_embeddings_dim % 64 or cross_attention_dim % 8:
            raise NotImplementedError("clip_embeddings_dim must be a multiple of 64, cross_attention_dim of 8")
        self.cross_attention_dim, self.clip_embeddings_dim, self.num_tokens = cross_attention_dim, clip_embeddings_dim, num_tokens

    def expected_shapes(self):
        E, D, T = self.clip_embeddings_dim, self.cross_attention_dim, self.num_tokens
        return {"proj.0.weight": (2 * E, E), "proj.0.bias": (2 * E,), "proj.2.weight": (D * T, 2 * E), "proj.2.bias": (D * T,),
                "norm.weight": (D,), "norm.bias": (D,)}

    def _pack(self):
        self._ready()
        sd, dev = self._sd, self._device
        self._w = dict(fc1=ops.pack_linear(sd["proj.0.weight"], sd["proj.0.bias"], dev),
                       fc2=ops.pack_linear(sd["proj.2.weight"], sd["proj.2.bias"], dev),
                       ln=(sd["norm.weight"].to(dev).contiguous(), sd["norm.bias"].to(dev).contiguous()))

    @torch.no_grad()
    def __call__(self, id_embeds: torch.Tensor) -> torch.Tensor:
        """id_embeds [B, E] -> [B, num_tokens, D] (bf16 compute, returned in the input's dtype)."""
        if self._w is None:
            self._pack()
        w = self._w
        E, D, T = self.clip_embeddings_dim, self.cross_attention_dim, self.num_tokens
        x = id_embeds.reshape(-1, id_embeds.shape[-1])
        if x.shape[1] != E:
            raise ValueError(f"expected last dim {E}, got {x.shape[1]}")
        M = x.shape[0]
        xb = ops.f32_to_bf16(x.to(self._device, torch.float32).contiguous(), self._buf("x", (M, E)))
        h = ops.gemm(xb, w["fc1"], self._buf("h", (M, 2 * E)), act=ops.ACT_GELU)
        y = ops.gemm(h, w["fc2"], self._buf("y", (M, D * T)))
        n = ops.layernorm(y.view(M * T, D), w["ln"][0], w["ln"][1], 1e-5, torch.empty(M * T, D, dtype=BF16, device=self._device))
        return n.view(M, T, D).to(id_embeds.dtype if id_embeds.dtype.is_floating_point else torch.float32)

    forward = __call__
