"""Python face of the C-side UNet schedule (``pcdm_unet_*`` in include/pcdm.h, pcdms_amd/csrc/unet_ctx.hip).

``UNetContext(unet)`` registers an already packed ``Stage2_InapintUNet2DConditionModel`` / ``UNet2DConditionModel`` with the C
context -- the same packed device tensors, under their diffusers module paths -- so that ONE C call runs the forward the Python
schedule of ``unet._forward_nhwc`` runs as ~380 ctypes calls.  What a non-Python host does with ``pcdm_unet_create`` /
``pcdm_unet_set_weight`` / ``pcdm_unet_forward`` is exactly this file minus torch: tests/test_unet_ctx.py holds the C schedule to the
Python one bit for bit (same kernels, same tile choices, same buffers' roles).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, ops
from .unet import Stage2_InapintUNet2DConditionModel, _resnets, _transformers


class UNetContext:
    def __init__(self, unet: Stage2_InapintUNet2DConditionModel):
        if unet._w is None:
            unet._pack()
        self.unet = unet
        self._pack_gen = unet._pack_gen
        lib = _lib.lib()
        cfg = _lib.UNetConfig()
        c = unet.config
        boc = unet._boc
        cfg.out_channels, cfg.n_levels = c.out_channels, len(boc)
        for i, v in enumerate(boc):
            cfg.block_out_channels[i] = v
            cfg.heads[i] = unet._heads[i]
            cfg.cross_attn[i] = int(c.down_block_types[i] == "CrossAttnDownBlock2D")
        cfg.layers_per_block, cfg.cross_attention_dim = unet._layers[0], c.cross_attention_dim
        cfg.norm_groups, cfg.norm_eps = c.norm_num_groups, float(c.norm_eps)
        cfg.class_embed = int(c.class_embed_type == "projection")
        cfg.flip_sin_to_cos, cfg.freq_shift = int(c.flip_sin_to_cos), float(c.freq_shift)
        self._h = lib.pcdm_unet_create(C.byref(cfg))
        if not self._h:
            raise RuntimeError("pcdm_unet_create rejected the topology")
        self._attn_fp8 = bool(unet._attn_fp8)
        self._chk(lib.pcdm_unet_set_attention_fp8(self._h, int(self._attn_fp8)), "pcdm_unet_set_attention_fp8")
        self._keep = []       # tensors the context points into
        self._ws = {}
        self._register()
        self.sync_tiles()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().pcdm_unet_destroy(h)

    # ------------------------------------------------------------------ registration
    def _chk(self, rc, what):
        if rc != 0:
            err = _lib.lib().pcdm_unet_last_error(self._h)
            raise RuntimeError(f"{what} failed with code {rc}: {err.decode() if err else ''}")

    def _set_w(self, name: str, pw: ops.PackedWeight):
        self._keep.append(pw)
        self._chk(_lib.lib().pcdm_unet_set_weight(self._h, name.encode(), pw.w.data_ptr(), None if pw.bias is None else pw.bias.data_ptr(),
                                                  None if pw.wsum is None else pw.wsum.data_ptr(), pw.N, pw.K, pw.Npad, pw.cin), name)

    def _set_small(self, name: str, wb):
        w, b = wb
        self._keep.append(wb)
        self._chk(_lib.lib().pcdm_unet_set_weight(self._h, name.encode(), w.data_ptr(), b.data_ptr(), None, w.shape[0], w.shape[1], w.shape[0], 0), name)

    def _set_v(self, name: str, gb):
        for sfx, t in zip(("weight", "bias"), gb):
            self._keep.append(t)
            self._chk(_lib.lib().pcdm_unet_set_vector(self._h, f"{name}.{sfx}".encode(), t.data_ptr(), t.numel()), name)

    def _register(self):
        W, u = self.unet._w, self.unet
        self._set_w("conv_in", W["conv_in"])
        self._set_w("conv_out", W["conv_out"])
        self._set_w("time_emb_proj", W["temb"])
        self._set_small("time_embedding.linear_1", W["time1"])
        self._set_small("time_embedding.linear_2", W["time2"])
        if "class1" in W:
            self._set_small("class_embedding.linear_1", W["class1"])
            self._set_small("class_embedding.linear_2", W["class2"])
        for p, _, _, _ in _resnets(u):
            r = W[p]
            self._set_w(p + "conv1", r["conv1"])
            if "conv2" in r:
                self._set_w(p + "conv2", r["conv2"])
            if "short" in r:
                self._set_w(p + "conv_shortcut", r["short"])
            if "conv2s" in r:
                self._set_w(p + "conv2s", r["conv2s"])
            self._set_v(p + "norm1", r["n1"])
            self._set_v(p + "norm2", r["n2"])
        for p, _, _ in _transformers(u):
            a = W[p]
            for k in ("proj_in", "proj_out", "qkv", "o1", "q2", "kv2", "o2", "ff1", "ff2", "ffo", "qkv_ln", "q2_ln", "ff1_ln"):
                if k in a:
                    self._set_w(p + k, a[k])
            self._set_v(p + "norm", a["norm"])
            for i in (1, 2, 3):
                self._set_v(p + f"transformer_blocks.0.norm{i}", a[f"ln{i}"])
        for i in range(len(u._boc) - 1):
            self._set_w(f"down_blocks.{i}.downsamplers.0.conv", W[f"down_blocks.{i}.downsamplers.0.conv."])
            self._set_w(f"up_blocks.{i}.upsamplers.0.conv", W[f"up_blocks.{i}.upsamplers.0.conv."])
            if f"up_blocks.{i}.upsamplers.0.conv4." in W:
                self._set_w(f"up_blocks.{i}.upsamplers.0.conv4", W[f"up_blocks.{i}.upsamplers.0.conv4."])
        self._set_v("conv_norm_out", W["norm_out"])

    def sync_tiles(self) -> int:
        """Hand the (tile, split-K) choices of the Python tuner (``ops._TUNED``: committed table + online tuning) to the context."""
        n = 0
        for key, (tile, split) in ops._TUNED.items():
            if key[0] == "ln":
                _, M, Npad, K, epi = key
                args = (1, M, Npad, K, 0, 0, 0, epi, 0, 0, 0)
            else:
                M, Npad, K, conv, stride, ups, epi, two, res = key[:9]
                args = (0, M, Npad, K, int(conv), int(stride), int(ups), epi, int(two), int(res), int(key[9]) if len(key) > 9 else 0)   # (1: zero_rows, 2: dup_rows)
            self._chk(_lib.lib().pcdm_unet_set_tile(self._h, *args, int(tile), int(split)), "pcdm_unet_set_tile")
            n += 1
        return n

    # ------------------------------------------------------------------ workspace
    def workspace(self, B: int, h: int, w: int, L: int) -> torch.Tensor:
        key = (B, h, w, L)
        ws = self._ws.get(key)
        if ws is None:
            n = _lib.lib().pcdm_unet_workspace_bytes(self._h, B, h, w, L)
            if n <= 0:
                raise RuntimeError("pcdm_unet_workspace_bytes failed")
            ws = torch.empty(n, dtype=torch.uint8, device=self.unet.device)
            self._chk(_lib.lib().pcdm_unet_workspace_init(self._h, B, h, w, L, ws.data_ptr(), ops._stream(ws)), "pcdm_unet_workspace_init")
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ run
    @torch.no_grad()
    def prepare_conditioning(self, B: int, h: int, w: int, encoder_hidden_states: torch.Tensor, class_labels: Optional[torch.Tensor],
                             my_pose_cond: Optional[torch.Tensor], zero_ctx_batches: int = 0, shared_cfg_input: bool = False) -> int:
        """Returns ``pose_b`` (0 / 1 / B), which ``forward`` wants back."""
        dev = self.unet.device
        ehs = encoder_hidden_states.to(dev, torch.float32).contiguous()
        L = ehs.shape[1]
        cl = None if class_labels is None else class_labels.reshape(B, -1).to(dev, torch.float32).contiguous()
        pose = None if my_pose_cond is None else my_pose_cond.to(dev, torch.float32).contiguous()
        pose_b = 0 if pose is None else pose.shape[0]
        ws = self.workspace(B, h, w, L)
        self._chk(_lib.lib().pcdm_unet_prepare_conditioning(self._h, B, h, w, L, ehs.data_ptr(), None if cl is None else cl.data_ptr(),
                                                            None if pose is None else pose.data_ptr(), pose_b, int(zero_ctx_batches),
                                                            ws.data_ptr(), ops._stream(ws)), "pcdm_unet_prepare_conditioning")
        self._chk(_lib.lib().pcdm_unet_set_shared_cfg_input(self._h, ws.data_ptr(), int(bool(shared_cfg_input))), "pcdm_unet_set_shared_cfg_input")
        self._L = L
        return pose_b

    @torch.no_grad()
    def prepare_timesteps(self, t_dev: torch.Tensor, B: int, h: int, w: int) -> None:
        """After ``prepare_conditioning``: the time / class embeddings and every ``time_emb_proj`` for ALL steps of the device timestep table
        ``t_dev`` (``pcdm_unet_prepare_timesteps``); ``forward`` calls that pass the same ``t_dev`` with a step counter then launch nothing for them."""
        assert t_dev.dtype == torch.int64 and t_dev.is_contiguous() and t_dev.device == self.unet.device
        n = t_dev.numel()
        nbytes = _lib.lib().pcdm_unet_time_table_bytes(self._h, n, B)
        key = ("ttab", n, B)
        tab = self._ws.get(key)
        if tab is None:
            tab = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.unet.device)
        ws = self.workspace(B, h, w, self._L)
        self._chk(_lib.lib().pcdm_unet_prepare_timesteps(self._h, t_dev.data_ptr(), n, tab.data_ptr(), ws.data_ptr(), ops._stream(ws)),
                  "pcdm_unet_prepare_timesteps")

    def step_overflow(self, B: int, h: int, w: int) -> bool:
        """``pcdm_unet_step_overflow``: did a forward on this shape's workspace find the device step counter outside the prepared time table?
        (clamped on the device, never an out-of-bounds read).  Synchronises."""
        import ctypes as C
        ws = self.workspace(B, h, w, self._L)
        flag = C.c_int(0)
        self._chk(_lib.lib().pcdm_unet_step_overflow(self._h, ws.data_ptr(), C.byref(flag), ops._stream(ws)), "pcdm_unet_step_overflow")
        return bool(flag.value)

    @torch.no_grad()
    def forward(self, x_in: torch.Tensor, t_dev: torch.Tensor, step_dev: Optional[torch.Tensor], B: int, h: int, w: int, pose_b: int,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_in NHWC bf16 [B, h, w, conv_in.cin]; t_dev int64 device tensor; returns fp32 NCHW eps."""
        assert x_in.dtype == ops.BF16 and x_in.is_contiguous() and t_dev.dtype == torch.int64
        if self._pack_gen != self.unet._pack_gen:
            raise RuntimeError("the UNet's weights were re-packed: build a new UNetContext")
        ws = self.workspace(B, h, w, self._L)
        if out is None:
            out = torch.empty(B, self.unet.config.out_channels, h, w, dtype=torch.float32, device=x_in.device)
        self._chk(_lib.lib().pcdm_unet_forward(self._h, x_in.data_ptr(), t_dev.data_ptr(), None if step_dev is None else step_dev.data_ptr(),
                                               B, h, w, self._L, pose_b, ws.data_ptr(), out.data_ptr(), ops._stream(x_in)), "pcdm_unet_forward")
        return out
