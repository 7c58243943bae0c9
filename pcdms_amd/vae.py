"""MI355X-native stand-in for the reference's ``pipe.vae``.

The reference uses diffusers' ``AutoencoderKL`` at two call sites of the stage-2 pipeline
(/root/reference/src/pipelines/stage2_inpaint_pipeline.py): ``self.vae.encode(vae_image).latent_dist.sample(generator)``
times ``self.vae.config.scaling_factor`` (:443-444) and ``self.vae.decode(latents / scaling_factor, return_dict=False)[0]``
followed by ``VaeImageProcessor.postprocess`` (:528-532); ``len(self.vae.config.block_out_channels)`` gives the 8x scale
factor (:138).  This class mirrors that surface (``from_pretrained`` / ``load_state_dict`` with diffusers key names /
``encode`` / ``decode`` / ``config``) and runs on the same HIP kernels as the UNet: implicit-GEMM 3x3 convolutions
(stride-2 with bottom/right padding for the encoder, nearest-x2 folded in for the decoder), GroupNorm+SiLU, and the
mid-block single-head d=C attention as two MFMA GEMMs around a row-softmax kernel.  ``quant_conv`` (1x1 after a 3x3) is
folded exactly into ``encoder.conv_out`` at pack time.
"""
from __future__ import annotations

import json
import math
from pathlib import Path
from types import SimpleNamespace
from typing import Any, Dict, Iterator, Optional, Tuple

import torch

from . import _lib, ops
from ._module import ModuleSurface
from .ops import BF16

SD21_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, sample_size=768, act_fn="silu",
                       down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4)


class _Config(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, d=None):
        return getattr(self, k, d)


class DiagonalGaussianDistribution:
    """``moments`` fp32 [B, 2*zc, h, w] on the device; ``sample`` is one HIP kernel."""

    def __init__(self, moments: torch.Tensor):
        self.parameters = moments
        self.mean, self.logvar = moments.chunk(2, dim=1)

    def sample(self, generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, c2, h, w = self.parameters.shape
        if noise is None:
            gdev = generator.device if isinstance(generator, torch.Generator) else self.parameters.device
            noise = torch.randn((B, c2 // 2, h, w), generator=generator, device=gdev, dtype=torch.float32)
        noise = noise.to(self.parameters.device, torch.float32).contiguous()
        out = torch.empty_like(noise)
        ops._chk(_lib.lib().pcdm_gaussian_sample(self.parameters.data_ptr(), noise.data_ptr(), out.data_ptr(), B, c2 // 2,
                                                 h * w, 1.0, ops._stream(out)), "pcdm_gaussian_sample")
        return out

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


class AutoencoderKL(ModuleSurface):
    def __init__(self, **kwargs):
        cfg = dict(SD21_VAE_CONFIG)
        cfg.update({k: v for k, v in kwargs.items() if k in cfg or k.startswith("_")})
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        self.config = _Config(**cfg)
        for c in cfg["block_out_channels"]:
            if c % 64:
                raise NotImplementedError("block_out_channels must be multiples of 64")
        self._device = torch.device("cpu")
        self._dtype = torch.float32
        self._sd: Optional[Dict[str, torch.Tensor]] = None
        self._w: Optional[Dict[str, Any]] = None
        self._bufs: Dict[Tuple, torch.Tensor] = {}

    # ---------------------------------------------------------------- module-like surface
    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return self._dtype

    def eval(self):
        return self

    def to(self, *args, **kwargs):
        device, dtype = kwargs.get("device"), kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif a is not None:
                device = torch.device(a)
        if dtype is not None:
            self._dtype = dtype
        if device is not None and torch.device(device) != self._device:
            self._device = torch.device(device)
            if self._device.type == "cuda" and self._device.index is None:
                self._device = torch.device("cuda", torch.cuda.current_device())
            self._w = None
            self._bufs.clear()
        return self

    def expected_shapes(self) -> Dict[str, Tuple[int, ...]]:
        return dict(_param_shapes(self.config))

    def state_dict(self):
        return dict(self._sd or {})

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        exp = self.expected_shapes()
        sd = dict(state_dict)
        # diffusers < 0.18 names of the mid-block attention
        for old, new in (("query", "to_q"), ("key", "to_k"), ("value", "to_v"), ("proj_attn", "to_out.0")):
            for k in list(sd):
                if f".attentions.0.{old}." in k:
                    sd[k.replace(f".attentions.0.{old}.", f".attentions.0.{new}.")] = sd.pop(k)
        missing = [k for k in exp if k not in sd]
        unexpected = [k for k in sd if k not in exp]
        bad = [k for k in exp if k in sd and tuple(sd[k].shape) != tuple(exp[k]) and sd[k].numel() != math.prod(exp[k])]
        if bad or (strict and (missing or unexpected)):
            raise RuntimeError(f"Error(s) in loading state_dict for AutoencoderKL: missing {missing[:6]} unexpected "
                               f"{unexpected[:6]} size mismatch {bad[:6]}")
        self._sd = {k: sd[k].detach().to("cpu", torch.float32).reshape(exp[k]) for k in exp if k in sd}
        self._w = None
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: Optional[str] = None, torch_dtype=None, **kwargs):
        d = Path(str(pretrained_model_name_or_path))
        d = d / subfolder if subfolder else d
        cfg = dict(SD21_VAE_CONFIG)
        if (d / "config.json").exists():
            cfg.update({k: v for k, v in json.loads((d / "config.json").read_text()).items() if k in cfg or k.startswith("_")})
        cfg.update(kwargs)
        m = cls(**cfg)
        sd = None
        if (d / "diffusion_pytorch_model.safetensors").exists():
            from safetensors.torch import load_file
            sd = load_file(str(d / "diffusion_pytorch_model.safetensors"))
        elif (d / "diffusion_pytorch_model.bin").exists():
            sd = torch.load(str(d / "diffusion_pytorch_model.bin"), map_location="cpu")
        if sd is None:
            raise FileNotFoundError(f"no diffusion_pytorch_model.{{safetensors,bin}} under {d}")
        m.load_state_dict(sd)
        if torch_dtype is not None:
            m.to(torch_dtype)
        return m

    # ---------------------------------------------------------------- packing
    def _pack(self):
        if self._sd is None:
            raise RuntimeError("weights not loaded")
        if self._device.type != "cuda" and not _lib.is_emulator():
            raise RuntimeError("AutoencoderKL runs on the MI355X only: call .to('cuda')")
        sd, dev = self._sd, self._device
        w: Dict[str, Any] = {}

        def f32(k):
            return sd[k].to(dev, torch.float32).contiguous()

        def res(p):
            r = dict(n1=(f32(p + "norm1.weight"), f32(p + "norm1.bias")), n2=(f32(p + "norm2.weight"), f32(p + "norm2.bias")),
                     conv1=ops.pack_conv3x3(sd[p + "conv1.weight"], sd[p + "conv1.bias"], dev),
                     conv2=ops.pack_conv3x3(sd[p + "conv2.weight"], sd[p + "conv2.bias"], dev))
            if p + "conv_shortcut.weight" in sd:
                r["short"] = ops.pack_linear(sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"], dev)
            return r

        def mid(p):
            a = p + "attentions.0."
            c = sd[a + "to_q.weight"].shape[0]
            return dict(r0=res(p + "resnets.0."), r1=res(p + "resnets.1."), c=c,
                        gn=(f32(a + "group_norm.weight"), f32(a + "group_norm.bias")),
                        qkv=ops.pack_linear(torch.cat([sd[a + "to_q.weight"], sd[a + "to_k.weight"], sd[a + "to_v.weight"]], 0),
                                            torch.cat([sd[a + "to_q.bias"], sd[a + "to_k.bias"], sd[a + "to_v.bias"]], 0), dev),
                        out=ops.pack_linear(sd[a + "to_out.0.weight"], sd[a + "to_out.0.bias"], dev))

        boc, L = self.config.block_out_channels, self.config.layers_per_block
        zc = self.config.latent_channels
        w["e.conv_in"] = ops.pack_conv3x3(sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], dev)
        for i in range(len(boc)):
            for j in range(L):
                w[f"e.d{i}.r{j}"] = res(f"encoder.down_blocks.{i}.resnets.{j}.")
            if i != len(boc) - 1:
                p = f"encoder.down_blocks.{i}.downsamplers.0.conv."
                w[f"e.d{i}.ds"] = ops.pack_conv3x3(sd[p + "weight"], sd[p + "bias"], dev)
        w["e.mid"] = mid("encoder.mid_block.")
        w["e.norm_out"] = (f32("encoder.conv_norm_out.weight"), f32("encoder.conv_norm_out.bias"))
        # quant_conv (1x1, after the 3x3 conv_out) folded in exactly: Wq (Wo * x + bo) + bq
        wq = sd["quant_conv.weight"].reshape(2 * zc, 2 * zc).double()
        wo, bo = sd["encoder.conv_out.weight"].double(), sd["encoder.conv_out.bias"].double()
        w["e.conv_out"] = ops.pack_conv3x3(torch.einsum("ab,bcij->acij", wq, wo).float(),
                                           (wq @ bo + sd["quant_conv.bias"].double()).float(), dev)
        w["d.post_quant"] = ops.pack_linear(torch.nn.functional.pad(sd["post_quant_conv.weight"].reshape(zc, zc), (0, 64 - zc)),
                                            sd["post_quant_conv.bias"], dev)
        w["d.conv_in"] = ops.pack_conv3x3(sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], dev)
        w["d.mid"] = mid("decoder.mid_block.")
        for i in range(len(boc)):
            for j in range(L + 1):
                w[f"d.u{i}.r{j}"] = res(f"decoder.up_blocks.{i}.resnets.{j}.")
            if i != len(boc) - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv."
                from .unet import PHASE_UPSAMPLE
                if PHASE_UPSAMPLE:   # Upsample2D as its phase decomposition (pcdms_amd/unet.py PHASE_UPSAMPLE: 4/9 of the FLOPs; the decoder's sizes are always x2)
                    w[f"d.u{i}.us4"] = ops.pack_upsample_phases(sd[p + "weight"], sd[p + "bias"], dev)
                else:
                    w[f"d.u{i}.us"] = ops.pack_conv3x3(sd[p + "weight"], sd[p + "bias"], dev)
        w["d.norm_out"] = (f32("decoder.conv_norm_out.weight"), f32("decoder.conv_norm_out.bias"))
        oc = self.config.out_channels
        wout = torch.zeros(4, *sd["decoder.conv_out.weight"].shape[1:])   # N must be a multiple of 4: pad 3 -> 4
        wout[:oc] = sd["decoder.conv_out.weight"]
        bout = torch.zeros(4)
        bout[:oc] = sd["decoder.conv_out.bias"]
        w["d.conv_out"] = ops.pack_conv3x3(wout, bout, dev)
        self._w = w

    def _buf(self, name, shape, dtype=BF16, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self._device)
            self._bufs[key] = t
        return t

    # ---------------------------------------------------------------- blocks
    def _gn(self, x, B, HW, gb, silu, name):
        C = x.shape[-1]
        ws = self._buf("gnws", (int(_lib.lib().pcdm_groupnorm_ws_floats(B, C)),), torch.float32, zero=True)
        return ops.groupnorm(x, None, B, HW, self.config.norm_num_groups, 1e-6, gb[0], gb[1], silu,
                             self._buf(name, (B * HW, C)), ws)

    def _resnet(self, r, x, B, H, W, name):
        HW, M = H * W, B * H * W
        cout = r["conv1"].N
        cv = dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W)
        n1 = self._gn(x, B, HW, r["n1"], True, "gn")
        h1 = ops.gemm(n1, r["conv1"], self._buf("c1", (M, cout)), conv=cv)
        n2 = self._gn(h1, B, HW, r["n2"], True, "gn2")
        resid = ops.gemm(x, r["short"], self._buf("sc", (M, cout))) if "short" in r else x
        return ops.gemm(n2, r["conv2"], self._buf(name, (M, cout)), conv=cv, residual=resid, res_mod=M)

    def _mid(self, m, x, B, H, W):
        """resnet -> single-head attention (d = C) with residual -> resnet."""
        HW, M, C = H * W, B * H * W, m["c"]
        if HW % 64:
            raise NotImplementedError("VAE mid-block attention needs (h/8)*(w/8) to be a multiple of 64")
        x = self._resnet(m["r0"], x, B, H, W, "ma")
        n = self._gn(x, B, HW, m["gn"], False, "gn")
        qk = self._buf("qk", (M, 2 * C))
        vt = self._buf("vt", (B, C, HW), zero=True)
        ops.gemm(n, m["qkv"], qk, rows_per_batch=HW, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * C)
        at = self._buf("at", (M, C))
        S = self._buf("S", (HW, HW), torch.float32)
        P = self._buf("P", (HW, HW))
        for b in range(B):   # S[q, k] = K Q^T stored transposed in fp32, softmax rows, O = P V
            q, k = qk[b * HW:(b + 1) * HW, :C], qk[b * HW:(b + 1) * HW, C:]
            wq = ops.PackedWeight(q, None, HW, C, HW, alg_nk=HW * C)           # "weights" = Q [HW, C] (row stride 2C)
            ops.gemm(k, wq, S, rows_per_batch=HW, epilogue=ops.EPI_NCHW_F32, w_ld=2 * C)
            ops.softmax_rows(S, P, C ** -0.5)
            wv = ops.PackedWeight(vt[b], None, C, HW, C, alg_nk=C * HW)        # "weights" = V^T [C, HW]
            ops.gemm(P, wv, at[b * HW:(b + 1) * HW])
        y = ops.gemm(at, m["out"], self._buf("mb", (M, C)), residual=x, res_mod=M)
        return self._resnet(m["r1"], y, B, H, W, "mc")

    # ---------------------------------------------------------------- encode / decode
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x [B,3,H,W] in [-1,1] -> latent_dist over [B,4,H/8,W/8]."""
        if self._w is None:
            self._pack()
        W_, cfg = self._w, self.config
        B, _, H, W = x.shape
        if H % 8 or W % 8:
            raise ValueError("image height/width must be multiples of 8")
        boc, L = cfg.block_out_channels, cfg.layers_per_block
        xin = ops.nchw_to_nhwc_bf16(x.to(self._device), self._buf("e.in", (B, H, W, 64)), cpad=64)
        h = ops.gemm(xin, W_["e.conv_in"], self._buf("e.x0", (B * H * W, boc[0])), conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W))
        for i in range(len(boc)):
            for j in range(L):
                h = self._resnet(W_[f"e.d{i}.r{j}"], h, B, H, W, "ra" if j % 2 == 0 else "rb")
            if i != len(boc) - 1:   # Downsample2D(padding=0): zero pad bottom/right only
                h = ops.gemm(h, W_[f"e.d{i}.ds"], self._buf("ds", (B * (H // 2) * (W // 2), boc[i])),
                             conv=dict(B=B, Hi=H, Wi=W, Ho=H // 2, Wo=W // 2, stride=2, no_pad_lo=1))
                H, W = H // 2, W // 2
        h = self._mid(W_["e.mid"], h, B, H, W)
        n = self._gn(h, B, H * W, W_["e.norm_out"], True, "gn")
        mom = torch.empty(B, 2 * cfg.latent_channels, H, W, dtype=torch.float32, device=self._device)
        ops.gemm(n, W_["e.conv_out"], mom, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), rows_per_batch=H * W,
                 epilogue=ops.EPI_NCHW_F32)
        dist = DiagonalGaussianDistribution(mom)
        return AutoencoderKLOutput(dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """z [B,4,h,w] (already divided by scaling_factor) -> image [B,3,8h,8w] fp32 (a fresh tensor: two calls never alias)."""
        img = self._decode_buf(z)[:, : self.config.out_channels].clone()
        return DecoderOutput(img) if return_dict else (img,)

    def _decode_buf(self, z: torch.Tensor) -> torch.Tensor:
        """The decoder proper; returns the reused fp32 [B,4,H,W] buffer (3 image channels + 1 padding channel)."""
        if self._w is None:
            self._pack()
        W_, cfg = self._w, self.config
        B, zc, H, W = z.shape
        boc, L = cfg.block_out_channels, cfg.layers_per_block
        rev = list(reversed(boc))
        zin = ops.nchw_to_nhwc_bf16(z.to(self._device), self._buf("d.in", (B, H, W, 64)), cpad=64)
        zq = ops.gemm(zin.view(B * H * W, 64), W_["d.post_quant"], self._buf("d.zq", (B * H * W, 64), zero=True))
        h = ops.gemm(zq, W_["d.conv_in"], self._buf("d.x0", (B * H * W, rev[0])), conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W))
        h = self._mid(W_["d.mid"], h, B, H, W)
        for i in range(len(boc)):
            for j in range(L + 1):
                h = self._resnet(W_[f"d.u{i}.r{j}"], h, B, H, W, "ra" if j % 2 == 0 else "rb")
            if i != len(boc) - 1:
                if f"d.u{i}.us4" in W_:   # Upsample2D as four 2x2 phase kernels on the low-res tensor (one launch, N = 4 C) + a pixel shuffle
                    ph = ops.gemm(h, W_[f"d.u{i}.us4"], self._buf("usp", (B * H * W, 4 * rev[i])), conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W),
                                  tap_lut=ops.UPSAMPLE_TAP_LUT, tap_group_n=rev[i])
                    h = ops.pixel_shuffle2(ph, self._buf("us", (B * 4 * H * W, rev[i])), B, H, W, rev[i])
                else:               # nearest x2 folded into the conv's gather
                    h = ops.gemm(h, W_[f"d.u{i}.us"], self._buf("us", (B * 4 * H * W, rev[i])),
                                 conv=dict(B=B, Hi=H, Wi=W, Ho=2 * H, Wo=2 * W, upsample=1))
                H, W = 2 * H, 2 * W
        n = self._gn(h, B, H * W, W_["d.norm_out"], True, "gn")
        img4 = self._buf("img4", (B, 4, H, W), torch.float32)      # 3 channels + 1 padding channel
        ops.gemm(n, W_["d.conv_out"], img4, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), rows_per_batch=H * W,
                 epilogue=ops.EPI_NCHW_F32)
        return img4

    @torch.no_grad()
    def decode_to_uint8(self, z: torch.Tensor) -> torch.Tensor:
        """decode + ``VaeImageProcessor.postprocess`` (ref :528-532) -> uint8 [B, H, W, 3] on the device."""
        base = self._decode_buf(z)
        B, _, H, W = base.shape
        out = torch.empty(B, H, W, 3, dtype=torch.uint8, device=self._device)
        ops._chk(_lib.lib().pcdm_image_to_uint8(base.data_ptr(), out.data_ptr(), B, base.shape[1], H * W, ops._stream(out)),
                 "pcdm_image_to_uint8")
        return out


def _param_shapes(cfg) -> Iterator[Tuple[str, Tuple[int, ...]]]:
    boc, L, zc = tuple(cfg.block_out_channels), cfg.layers_per_block, cfg.latent_channels

    def res(p, cin, cout):
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

    def mid(p, c):
        yield from res(p + "resnets.0.", c, c)
        a = p + "attentions.0."
        yield a + "group_norm.weight", (c,)
        yield a + "group_norm.bias", (c,)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            yield a + n + ".weight", (c, c)
            yield a + n + ".bias", (c,)
        yield from res(p + "resnets.1.", c, c)

    yield "encoder.conv_in.weight", (boc[0], cfg.in_channels, 3, 3)
    yield "encoder.conv_in.bias", (boc[0],)
    out = boc[0]
    for i in range(len(boc)):
        cin, out = out, boc[i]
        for j in range(L):
            yield from res(f"encoder.down_blocks.{i}.resnets.{j}.", cin if j == 0 else out, out)
        if i != len(boc) - 1:
            yield f"encoder.down_blocks.{i}.downsamplers.0.conv.weight", (out, out, 3, 3)
            yield f"encoder.down_blocks.{i}.downsamplers.0.conv.bias", (out,)
    yield from mid("encoder.mid_block.", boc[-1])
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
    yield from mid("decoder.mid_block.", rev[0])
    out = rev[0]
    for i in range(len(boc)):
        prev, out = out, rev[i]
        for j in range(L + 1):
            yield from res(f"decoder.up_blocks.{i}.resnets.{j}.", prev if j == 0 else out, out)
        if i != len(boc) - 1:
            yield f"decoder.up_blocks.{i}.upsamplers.0.conv.weight", (out, out, 3, 3)
            yield f"decoder.up_blocks.{i}.upsamplers.0.conv.bias", (out,)
    yield "decoder.conv_norm_out.weight", (boc[0],)
    yield "decoder.conv_norm_out.bias", (boc[0],)
    yield "decoder.conv_out.weight", (cfg.out_channels, boc[0], 3, 3)
    yield "decoder.conv_out.bias", (cfg.out_channels,)
