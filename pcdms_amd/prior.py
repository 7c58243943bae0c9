"""Stage-1 prior on the HIP kernels: ``Stage1_PriorTransformer`` + ``Stage1_PriorPipeline``.

Mirrors the surface the reference's stage-1 driver uses (/root/reference/stage1_batchtest_prior_model.py:55-59,105-113):
``Stage1_PriorTransformer.from_pretrained(path, subfolder="prior", num_embeddings=2, embedding_dim=1024, ...)``,
``load_state_dict`` with the reference's key names, ``forward(hidden_states, timestep, proj_embedding,
encoder_hidden_states, encoder_hidden_states1, attention_mask=None)`` (src/models/stage1_prior_transformer.py:197-297),
``post_process_latents`` (:299-301), and ``Stage1_PriorPipeline(...)(s_embed=, s_pose=, t_pose=, num_images_per_prompt=,
num_inference_steps=, generator=, guidance_scale=)`` (src/pipelines/stage1_prior_pipeline.py:355-504).

The model is six tokens wide (source pose, target pose, source-image CLIP embedding, time, x_t, learned read-out token;
:264-274) and 20 blocks deep at width 2048: every step streams ~2 GB of bf16 weights through thin GEMMs (M = 6 x batch),
i.e. it is HBM-bound on the weights, not MFMA-bound.  Everything runs on libpcdm.so: the token assembly is done by the
GEMM epilogue (strided output rows + the positional embedding as a broadcast residual), attention by the d=64 flash
kernel (32 heads, 6 keys), Linear+GELU / Linear+SiLU by the activation epilogue, the read-out writes fp32.
"""
from __future__ import annotations

import json
import math
from pathlib import Path
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _lib, ops
from .cond import _HipModule
from .ops import BF16
from .schedulers import UnCLIPScheduler

KANDINSKY22_PRIOR_CONFIG = dict(num_attention_heads=32, attention_head_dim=64, num_layers=20, embedding_dim=1280,
                                num_embeddings=77, additional_embeddings=4, dropout=0.0)
CLIP_MEAN, CLIP_STD = -0.016, 0.415     # stage1_prior_transformer.py:132-133
_POSE_DIM, _POSE_HIDDEN, _POSE_OUT = 36, 512, 1024   # MLP(in_dim=36, hidden_dim=512, out_dim=1024), :97-98


class PriorTransformerOutput:
    def __init__(self, predicted_image_embedding):
        self.predicted_image_embedding = predicted_image_embedding

    def __getitem__(self, i):
        return (self.predicted_image_embedding,)[i]


class _Config(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, d=None):
        return getattr(self, k, d)


class Stage1_PriorTransformer(_HipModule):
    _name = "Stage1_PriorTransformer"

    def __init__(self, num_attention_heads: int = 32, attention_head_dim: int = 64, num_layers: int = 20,
                 embedding_dim: int = 768, num_embeddings=77, additional_embeddings=4, dropout: float = 0.0, **kwargs):
        super().__init__()
        if attention_head_dim != 64:
            raise NotImplementedError("the attention kernel is specialised for head_dim 64")
        if embedding_dim != _POSE_OUT:
            raise NotImplementedError("embedding_dim must be 1024: the reference's pose MLP emits 1024 features "
                                      "(stage1_prior_transformer.py:97-98) into Linear(embedding_dim, inner_dim)")
        self.config = _Config(num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                              num_layers=num_layers, embedding_dim=embedding_dim, num_embeddings=num_embeddings,
                              additional_embeddings=additional_embeddings, dropout=dropout,
                              **{k: v for k, v in kwargs.items() if k.startswith("_")})
        self.clip_mean, self.clip_std = torch.tensor(CLIP_MEAN), torch.tensor(CLIP_STD)
        self._dtype = torch.float32
        self._static: Optional[Tuple] = None

    # ---------------------------------------------------------------- module-like surface
    @property
    def dtype(self):
        return self._dtype

    @property
    def inner_dim(self) -> int:
        return self.config.num_attention_heads * self.config.attention_head_dim

    @property
    def num_tokens(self) -> int:
        return self.config.num_embeddings + self.config.additional_embeddings

    def to(self, *args, **kwargs):
        for a in list(args) + [kwargs.get("dtype")]:
            if isinstance(a, torch.dtype):
                self._dtype = a
        self._static = None
        return super().to(*args, **kwargs)

    def half(self):
        return self.to(torch.float16)

    def expected_shapes(self) -> Dict[str, Tuple[int, ...]]:
        c, D, E = self.config, self.inner_dim, self.config.embedding_dim
        exp: Dict[str, Tuple[int, ...]] = {}

        def lin(p, o, i):
            exp[p + ".weight"], exp[p + ".bias"] = (o, i), (o,)

        def ln(p, n):
            exp[p + ".weight"], exp[p + ".bias"] = (n,), (n,)
        for pe in ("pose_encoder", "pose_encoder1"):
            lin(pe + ".net.0", _POSE_HIDDEN, _POSE_DIM); ln(pe + ".net.3", _POSE_HIDDEN)
            lin(pe + ".net.4", _POSE_OUT, _POSE_HIDDEN); ln(pe + ".net.6", _POSE_OUT)
        lin("time_embedding.linear_1", D, D); lin("time_embedding.linear_2", D, D)
        lin("proj_in", D, E); lin("embedding_proj", D, E)
        lin("encoder_hidden_states_proj", D, E); lin("encoder_hidden_states_proj1", D, E)
        exp["positional_embedding"] = (1, self.num_tokens, D)
        exp["prd_embedding"] = (1, 1, D)
        for i in range(c.num_layers):
            p = f"transformer_blocks.{i}."
            ln(p + "norm1", D)
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(p + "attn1." + n, D, D)
            ln(p + "norm3", D)
            lin(p + "ff.net.0.proj", 4 * D, D); lin(p + "ff.net.2", D, 4 * D)
        ln("norm_out", D); lin("proj_to_clip_embeddings", E, D)
        return exp

    def load_state_dict(self, state_dict, strict: bool = True):
        self._static = None
        return super().load_state_dict(state_dict, strict)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: Optional[str] = None, torch_dtype=None,
                        low_cpu_mem_usage: bool = False, ignore_mismatched_sizes: bool = False, **kwargs):
        """``{path}/{subfolder}/config.json`` + ``diffusion_pytorch_model.{safetensors,bin}``; kwargs override the config
        (stage1_batchtest_prior_model.py:56: num_embeddings=2, embedding_dim=1024); with ``ignore_mismatched_sizes`` the
        tensors whose shapes differ (and the reference-only pose encoders) keep a fresh PyTorch-default init, as diffusers
        does -- the driver then overwrites everything with ``load_state_dict`` (:58-59)."""
        d = Path(str(pretrained_model_name_or_path))
        d = d / subfolder if subfolder else d
        cfg = dict(KANDINSKY22_PRIOR_CONFIG)
        if (d / "config.json").exists():
            cfg.update({k: v for k, v in json.loads((d / "config.json").read_text()).items() if k in cfg or k.startswith("_")})
        cfg.update(kwargs)
        m = cls(**cfg)
        exp = m.expected_shapes()
        g = torch.Generator().manual_seed(0)
        sd = {}
        for k, shp in exp.items():   # fresh init
            wk = k[: k.rfind(".") + 1] + "weight"
            if k in ("positional_embedding", "prd_embedding"):
                sd[k] = torch.zeros(shp)
            elif len(exp[wk]) == 1:
                sd[k] = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
            else:
                sd[k] = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(exp[wk][1])
        loaded = None
        if (d / "diffusion_pytorch_model.safetensors").exists():
            from safetensors.torch import load_file
            loaded = load_file(str(d / "diffusion_pytorch_model.safetensors"))
        elif (d / "diffusion_pytorch_model.bin").exists():
            loaded = torch.load(str(d / "diffusion_pytorch_model.bin"), map_location="cpu")
        if loaded is not None:
            for k, v in loaded.items():
                if k in exp and tuple(v.shape) == tuple(exp[k]):
                    sd[k] = v
                elif k in exp and not ignore_mismatched_sizes:
                    raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {exp[k]}")
        m.load_state_dict(sd)
        if torch_dtype is not None:
            m.to(torch_dtype)
        return m

    def set_use_memory_efficient_attention_xformers(self, valid: bool = True, attention_op=None):
        return None   # the fused attention kernel is always on (pipe.enable_xformers_memory_efficient_attention(), :60)

    def post_process_latents(self, prior_latents):
        return prior_latents * CLIP_STD + CLIP_MEAN

    # ---------------------------------------------------------------- packing
    def _pack(self):
        self._ready()
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1   # (a captured hipGraph holds pointers into the packed weights)
        sd, dev, D = self._sd, self._device, self.inner_dim

        def f32(k):
            return sd[k].to(dev, torch.float32).contiguous()

        def lin(p):
            return ops.pack_linear(sd[p + ".weight"], sd[p + ".bias"], dev)

        def pose(p):
            w0 = torch.zeros(_POSE_HIDDEN, 64)
            w0[:, :_POSE_DIM] = sd[p + ".net.0.weight"]         # K 36 -> 64 (the input is zero-padded to match)
            return dict(l0=ops.pack_linear(w0, sd[p + ".net.0.bias"], dev), n3=(f32(p + ".net.3.weight"), f32(p + ".net.3.bias")),
                        l4=lin(p + ".net.4"), n6=(f32(p + ".net.6.weight"), f32(p + ".net.6.bias")))
        w: Dict[str, Any] = dict(pose=pose("pose_encoder"), pose1=pose("pose_encoder1"), t1=lin("time_embedding.linear_1"),
                                 t2=lin("time_embedding.linear_2"), proj_in=lin("proj_in"), emb=lin("embedding_proj"),
                                 ehsp=lin("encoder_hidden_states_proj"), ehsp1=lin("encoder_hidden_states_proj1"),
                                 norm_out=(f32("norm_out.weight"), f32("norm_out.bias")), out=lin("proj_to_clip_embeddings"))
        pos = sd["positional_embedding"][0]
        w["pos"] = pos.to(BF16).to(dev).contiguous()                                       # [T, D]: residual rows of the token GEMMs
        w["prd"] = (sd["prd_embedding"][0, 0] + pos[-1]).to(BF16).to(dev).contiguous()     # weights-only: folded at load
        blocks = []
        for i in range(self.config.num_layers):
            p = f"transformer_blocks.{i}."
            a = p + "attn1."
            blocks.append(dict(
                n1=(f32(p + "norm1.weight"), f32(p + "norm1.bias")), n3=(f32(p + "norm3.weight"), f32(p + "norm3.bias")),
                qkv=ops.pack_linear(torch.cat([sd[a + "to_q.weight"], sd[a + "to_k.weight"], sd[a + "to_v.weight"]], 0),
                                    torch.cat([sd[a + "to_q.bias"], sd[a + "to_k.bias"], sd[a + "to_v.bias"]], 0), dev),
                o=lin(a + "to_out.0"), ff1=lin(p + "ff.net.0.proj"), ff2=lin(p + "ff.net.2")))
        w["blocks"] = blocks
        self._w = w

    # ---------------------------------------------------------------- forward
    def _token(self, tokens: torch.Tensor, j: int) -> torch.Tensor:
        """[B, D] view of token j of every batch row inside the [B*T, D] token buffer (row stride T*D)."""
        T, D = self.num_tokens, self.inner_dim
        return tokens.view(-1, T * D)[:, j * D:(j + 1) * D]

    def _in_bf16(self, name: str, x: torch.Tensor, B: int, width: int) -> torch.Tensor:
        """[B, (1,) k] float input -> bf16 [B, width] on the device (zero-padded on the right)."""
        x = x.reshape(B, -1).to(self._device, torch.float32)
        if x.shape[1] != width:
            x = torch.nn.functional.pad(x, (0, width - x.shape[1]))
        return ops.f32_to_bf16(x.contiguous(), self._buf(name, (B, width)))

    def _static_tokens(self, tokens, proj_embedding, ehs, ehs1, B):
        """tokens 0 (source pose), 1 (target pose), 2 (image embedding) and T-1 (read-out): step-invariant."""
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (proj_embedding, ehs, ehs1)) + (B,)
        if self._static is not None and self._static[0] == key:
            return
        w, E = self._w, self.config.embedding_dim
        for j, (x, pw, proj) in enumerate(((ehs, w["pose"], w["ehsp"]), (ehs1, w["pose1"], w["ehsp1"]))):
            h = ops.gemm(self._in_bf16("pose_in", x, B, 64), pw["l0"], self._buf("pose_h", (B, _POSE_HIDDEN)), act=ops.ACT_GELU)
            h = ops.layernorm(h, pw["n3"][0], pw["n3"][1], 1e-5, self._buf("pose_hn", (B, _POSE_HIDDEN)))
            h = ops.gemm(h, pw["l4"], self._buf("pose_o", (B, _POSE_OUT)))
            h = ops.layernorm(h, pw["n6"][0], pw["n6"][1], 1e-5, self._buf("pose_on", (B, _POSE_OUT)))
            ops.gemm(h, proj, self._token(tokens, j), residual=w["pos"][j:j + 1], res_mod=1)
        ops.gemm(self._in_bf16("emb_in", proj_embedding, B, E), w["emb"], self._token(tokens, 2), residual=w["pos"][2:3], res_mod=1)
        self._token(tokens, self.num_tokens - 1).copy_(w["prd"])    # device-to-device copy of a load-time constant
        self._static = (key, proj_embedding, ehs, ehs1)             # (keeps the keyed tensors alive)

    @torch.no_grad()
    def forward(self, hidden_states, timestep: Union[torch.Tensor, float, int], proj_embedding, encoder_hidden_states,
                encoder_hidden_states1, attention_mask=None, return_dict: bool = True,
                do_classifier_free_guidance: bool = False, test_flag: bool = False, _step_dev: Optional[torch.Tensor] = None):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask: the reference's only call site passes None (stage1_prior_pipeline.py:464)")
        if test_flag:
            raise NotImplementedError("test_flag")
        if self.num_tokens != 6:
            raise NotImplementedError("token layout is 2 pose + embedding + time + x_t + read-out (num_embeddings=2, "
                                      "additional_embeddings=4; stage1_batchtest_prior_model.py:56)")
        if self._w is None:
            self._pack()
        w, c, D, E, T = self._w, self.config, self.inner_dim, self.config.embedding_dim, self.num_tokens
        B = hidden_states.shape[0]
        M = B * T
        if _step_dev is not None:   # captured loop: ``timestep`` is the device table of all timesteps, indexed by the device step counter
            t_dev = timestep
        else:
            if torch.is_tensor(timestep):
                tv = timestep.reshape(-1)
                if tv.numel() > 1 and not bool((tv == tv[0]).all()):
                    raise NotImplementedError("per-row timesteps")
                timestep = tv[0].item()
            t_dev = torch.tensor([int(timestep)], dtype=torch.int64, device=self._device)
        tokens = self._buf("tokens", (M, D))
        self._static_tokens(tokens, proj_embedding, encoder_hidden_states, encoder_hidden_states1, B)
        # time token: sinusoid -> Linear+SiLU -> Linear (+ positional row 3)
        sin = ops.timestep_embedding(t_dev, _step_dev, self._buf("sin", (B, D), torch.float32), flip=True, shift=0.0)
        th = ops.gemm(ops.f32_to_bf16(sin, self._buf("sinb", (B, D))), w["t1"], self._buf("th", (B, D)), act=ops.ACT_SILU)
        ops.gemm(th, w["t2"], self._token(tokens, 3), residual=w["pos"][3:4], res_mod=1)
        # x_t token
        ops.gemm(self._in_bf16("x_in", hidden_states, B, E), w["proj_in"], self._token(tokens, 4), residual=w["pos"][4:5], res_mod=1)
        x = tokens
        H = c.num_attention_heads
        for i, blk in enumerate(w["blocks"]):
            n = ops.layernorm(x, blk["n1"][0], blk["n1"][1], 1e-5, self._buf("ln", (M, D)))
            qk = self._buf("qk", (M, 2 * D))
            vt = self._buf("vt", (B, D, 8), zero=True)
            ops.gemm(n, blk["qkv"], qk, rows_per_batch=T, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * D)
            at = ops.flash_attn(qk[:, :D], qk[:, D:], vt, self._buf("at", (M, D)), B, H, T, T)
            x1 = ops.gemm(at, blk["o"], self._buf("xa", (M, D)), residual=x, res_mod=M)
            n = ops.layernorm(x1, blk["n3"][0], blk["n3"][1], 1e-5, self._buf("ln", (M, D)))
            f = ops.gemm(n, blk["ff1"], self._buf("ff", (M, 4 * D)), act=ops.ACT_GELU)
            x = ops.gemm(f, blk["ff2"], self._buf("xb", (M, D)), residual=x1, res_mod=M)
        n = ops.layernorm(x, w["norm_out"][0], w["norm_out"][1], 1e-5, self._buf("ln", (M, D)))
        o = self._buf("out", (B, E, T), torch.float32)
        ops.gemm(n, w["out"], o, rows_per_batch=T, epilogue=ops.EPI_NCHW_F32)     # fp32 [B, E, T]
        pred = o[:, :, T - 1].contiguous()                                          # hidden_states[:, -1] (:291)
        return PriorTransformerOutput(pred) if return_dict else (pred,)

    __call__ = forward

    def _buf(self, name, shape, dtype=BF16, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self._device)
        return t


class KandinskyPriorPipelineOutput:
    def __init__(self, image_embeds, negative_image_embeds):
        self.image_embeds, self.negative_image_embeds = image_embeds, negative_image_embeds

    def __getitem__(self, i):
        return (self.image_embeds, self.negative_image_embeds)[i]


class Stage1_PriorPipeline:
    """src/pipelines/stage1_prior_pipeline.py:355-504 for one (source, target) pair.

    ``image_encoder`` (CLIP-H vision tower) is only needed for ``negative_image_embeds`` =
    encoder(zero image) (:281-288); without one that output is ``None``.  Deviations, both stated in DESIGN.md: with
    ``guidance_scale > 1`` the reference doubles the image embedding but not the poses and fails in ``torch.cat``
    (stage1_prior_transformer.py:264); here the poses are doubled too (zero embedding + same poses = unconditional
    rows).  The reference draws the scheduler's variance noise from the global RNG (:478 passes no generator); here the
    pipeline's ``generator`` is used, or ``variance_noises`` (one [N, E] tensor per step) when given."""

    def __init__(self, prior: Optional[Stage1_PriorTransformer], scheduler: Optional[UnCLIPScheduler] = None, image_encoder=None,
                 image_processor=None):
        self.prior, self.image_encoder, self.image_processor = prior, image_encoder, image_processor
        self.scheduler = scheduler or UnCLIPScheduler(**UnCLIPScheduler.KANDINSKY22_PRIOR)
        self._to_device = prior.device if prior is not None else torch.device("cpu")

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, **kwargs):
        """``Stage1_PriorPipeline.from_pretrained(kandinsky_2_2_prior_dir)`` as the stage-1 driver calls it
        (/root/reference/stage1_batchtest_prior_model.py:55): takes the scheduler from ``scheduler/scheduler_config.json``.
        The directory's stock prior (embedding_dim 1280, 81 tokens) is not this model and is not loaded -- the driver assigns
        ``pipe.prior`` on the next line (:56); the CLIP text / image towers of the directory are not used by this call."""
        d = Path(str(pretrained_model_name_or_path)) / "scheduler" / "scheduler_config.json"
        scfg = json.loads(d.read_text()) if d.exists() else dict(UnCLIPScheduler.KANDINSKY22_PRIOR)
        return cls(None, UnCLIPScheduler.from_config(scfg))

    def to(self, device):
        self._to_device = torch.device(device)
        for m in (self.prior, self.image_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(device)
        return self

    @property
    def device(self):
        return self.prior.device if self.prior is not None else self._to_device

    @property
    def _device(self):
        return self.device

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        if self.prior is not None:
            self.prior.set_use_memory_efficient_attention_xformers(True, attention_op)

    def prepare_latents(self, shape, dtype, device, generator, latents, scheduler):
        if latents is None:
            gdev = generator.device if isinstance(generator, torch.Generator) else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32)
        elif tuple(latents.shape) != tuple(shape):
            raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
        return (latents.to(device, torch.float32) * scheduler.init_noise_sigma).contiguous()

    def _run_graph(self, x, emb, sp, tp, ts, rows, noises, cfg_on, g):
        """The denoise loop (ref :453-483) as ONE captured step replayed ``len(ts)`` times: prior forward (timestep from a device
        table at the device step counter) -> ``pcdm_unclip_step_dev`` (CFG + scheduler step, coefficients and noise from device
        tables) -> ``pcdm_advance_step``.  ~150 launches of a few microseconds each per step otherwise pay the host's launch rate."""
        dev, prior = x.device, self.prior
        n = len(ts)
        key = (tuple(x.shape), tuple(emb.shape), n, cfg_on, g, id(prior), getattr(prior, "_pack_gen", 0))
        st = getattr(self, "_gst", None)
        if st is None or st["key"] != key:
            st = dict(key=key, x=torch.empty_like(x), emb=torch.empty_like(emb), sp=torch.empty_like(sp), tp=torch.empty_like(tp),
                      ts=torch.empty(n, dtype=torch.int64, device=dev), coef=torch.empty(n, 8, dtype=torch.float32, device=dev),
                      noise=torch.empty(n, x.numel(), dtype=torch.float32, device=dev),
                      step=torch.zeros(1, dtype=torch.int32, device=dev), graph=None)
            self._gst = st
        st["x"].copy_(x); st["emb"].copy_(emb); st["sp"].copy_(sp); st["tp"].copy_(tp)
        st["ts"].copy_(torch.tensor(ts, dtype=torch.int64))
        st["coef"].copy_(torch.tensor(rows, dtype=torch.float32))
        st["noise"].copy_(torch.stack([z.reshape(-1) for z in noises]))
        st["step"].zero_()

        def step():
            xin = torch.cat([st["x"], st["x"]]) if cfg_on else st["x"]
            pred = prior(xin.unsqueeze(1), timestep=st["ts"], proj_embedding=st["emb"], encoder_hidden_states=st["sp"],
                         encoder_hidden_states1=st["tp"], attention_mask=None, _step_dev=st["step"]).predicted_image_embedding
            ops.unclip_step_dev(pred, cfg_on, g, st["x"], st["noise"], st["coef"], st["step"])
            ops.advance_step(st["step"])
        # the step-invariant tokens (poses, image embedding) are refreshed eagerly from this call's tensors; the captured step then
        # finds them cached (same static tensors, same versions) and contains only the per-step kernels
        prior._static = None
        if st["graph"] is None or st.get("ws_gen") != ops.workspace_generation(dev):
            x0 = st["x"].clone()
            step()                                    # warm-up: allocates scratch, autotunes unseen GEMM shapes
            torch.cuda.synchronize()
            st["x"].copy_(x0); st["step"].zero_()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                step()
            st["x"].copy_(x0); st["step"].zero_()
            st["graph"], st["ws_gen"] = graph, ops.workspace_generation(dev)
        else:
            step_tokens = prior._static_tokens        # (eager refresh of tokens 0-2 for this call's conditioning)
            B = (2 if cfg_on else 1) * x.shape[0]
            step_tokens(prior._buf("tokens", (B * prior.num_tokens, prior.inner_dim)), st["emb"], st["sp"], st["tp"], B)
        for _ in range(n):
            st["graph"].replay()
        return st["x"].clone()

    def get_zero_embed(self, batch_size=1, device=None):
        if self.image_encoder is None:
            return None
        size = self.image_encoder.config.image_size
        z = self.image_encoder(torch.zeros(1, 3, size, size, device=device or self._device))["image_embeds"]
        return z.repeat(batch_size, 1)

    @torch.no_grad()
    def __call__(self, s_embed, s_pose, t_pose, negative_prompt=None, num_images_per_prompt: int = 1,
                 num_inference_steps: int = 25, generator=None, latents=None, guidance_scale: float = 4.0,
                 output_type: Optional[str] = "pt", return_dict: bool = True,
                 variance_noises: Optional[Sequence[torch.Tensor]] = None, use_graph: bool = True):
        if output_type not in ("pt", "np"):
            raise ValueError(f"Only the output types `pt` and `np` are supported not output_type={output_type}")
        if s_embed.shape[0] != 1:
            raise NotImplementedError("one (source, target) pair per call")
        dev, N = self._device, num_images_per_prompt
        cfg_on = guidance_scale > 1.0
        emb = s_embed.to(dev, torch.float32).repeat(N, 1, 1)
        sp = s_pose.to(dev, torch.float32).reshape(1, 1, -1).repeat(N, 1, 1)
        tp = t_pose.to(dev, torch.float32).reshape(1, 1, -1).repeat(N, 1, 1)
        if cfg_on:
            emb = torch.cat([torch.zeros_like(emb), emb])
            sp, tp = torch.cat([sp, sp]), torch.cat([tp, tp])
        self.scheduler.set_timesteps(num_inference_steps, device=None)
        ts = [int(t) for t in self.scheduler.timesteps.tolist()]
        E = self.prior.config.embedding_dim
        x = self.prepare_latents((N, E), torch.float32, dev, generator, latents, self.scheduler)
        n = len(ts)
        # per-step scalars and noise, all known before the loop (the noise is drawn in the same order the reference's loop would)
        rows, noises = [], []
        for i, t in enumerate(ts):
            last = i + 1 == n
            co = self.scheduler.step_coefficients(t, None if last else ts[i + 1])
            rows.append((*co, CLIP_STD if last else 1.0, CLIP_MEAN if last else 0.0))   # post_process_latents (:485) folded into the last step
            if co[5] > 0:
                if variance_noises is not None:
                    noises.append(variance_noises[i].to(dev, torch.float32).reshape(N, E))
                else:
                    gdev = generator.device if isinstance(generator, torch.Generator) else dev
                    noises.append(torch.randn((N, E), generator=generator, device=gdev, dtype=torch.float32).to(dev))
            else:
                noises.append(torch.zeros(N, E, device=dev))
        if use_graph and dev.type == "cuda":
            x = self._run_graph(x, emb, sp, tp, ts, rows, noises, cfg_on, float(guidance_scale))
        else:
            for i, t in enumerate(ts):
                xin = torch.cat([x, x]) if cfg_on else x
                pred = self.prior(xin.unsqueeze(1), timestep=t, proj_embedding=emb, encoder_hidden_states=sp,
                                  encoder_hidden_states1=tp, attention_mask=None).predicted_image_embedding
                # CFG combine (:467-471) + UnCLIPScheduler.step (:478-483)
                x = ops.unclip_step(pred, cfg_on, float(guidance_scale), x, noises[i].contiguous() if rows[i][5] > 0 else None,
                                    torch.empty_like(x), rows[i])
        image_embeddings = x
        if negative_prompt is None:
            zero_embeds = self.get_zero_embed(x.shape[0], device=dev)
        else:
            image_embeddings, zero_embeds = image_embeddings.chunk(2)
        if output_type == "np":
            image_embeddings = image_embeddings.cpu().numpy()
            zero_embeds = None if zero_embeds is None else zero_embeds.cpu().numpy()
        if not return_dict:
            return (image_embeddings, zero_embeds)
        return KandinskyPriorPipelineOutput(image_embeds=image_embeddings, negative_image_embeds=zero_embeds)
