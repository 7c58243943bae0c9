"""Thin tensor-level wrappers over the C-ABI of libpcdm.so (include/pcdm.h).

PyTorch is used for device memory and streams only; every op here is one call into the
hand-written HIP library on the caller's current stream.  Tensors must live on a ROCm device
(``cuda``); CPU tensors are accepted only when the loaded library is the test emulator build.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from pathlib import Path
from dataclasses import dataclass
from typing import Optional, Sequence, Union

import torch

from . import _lib
from ._lib import GemmParams

EPI_STORE, EPI_GEGLU, EPI_SPLIT_VT, EPI_NCHW_F32 = 0, 1, 2, 3
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
BF16 = torch.bfloat16
LAUNCH_LOG: Optional[list] = None  # set to [] by bench.py to time individual launches with HIP events
LAUNCH_KEYS: Optional[list] = None  # set to [] by tools/tune_in_step.py: (index into LAUNCH_LOG, tuning-table key) of every table-driven GEMM launch
LAUNCH_SPANS: Optional[list] = None  # ... and ("ln" key, first, end) = the LAUNCH_LOG entries of every LayerNorm -> Linear pair
AUTOTUNE = True                    # pick the GEMM tile configuration per problem shape at first use (GPU only)
DEFER_SPLITK = os.environ.get("PCDM_DEFER_SPLITK", "1") != "0"   # split-K reduce folded into the consuming GroupNorm (A/B switch)


def _stream(t: torch.Tensor) -> Optional[int]:
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    if not _lib.is_emulator():
        raise RuntimeError("pcdms_amd ops need tensors on the MI355X (device 'cuda'); there is no CPU path")
    return None


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(rc: int, name: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")


def _c(t: torch.Tensor, dtype) -> torch.Tensor:
    assert t.dtype == dtype and t.is_contiguous(), (t.dtype, dtype, t.is_contiguous())
    return t


# ------------------------------------------------------------------------------------ norms
def groupnorm_ws(B: int, C: int, device) -> torch.Tensor:
    n = _lib.lib().pcdm_groupnorm_ws_floats(B, C)
    return torch.zeros(n, dtype=torch.float32, device=device)   # arrival counters must start at zero


def groupnorm_cluster_timeouts(ws: torch.Tensor) -> int:
    """How many workgroups of the in-launch-exchange GroupNorm path gave up waiting for their partners (and computed alone) since the
    workspace was allocated; 0 unless something else held CUs during a launch.  Synchronous (diagnostics / tests)."""
    n = C.c_uint(0)
    _chk(_lib.lib().pcdm_groupnorm_cluster_timeouts(_ptr(ws), C.byref(n), None), "pcdm_groupnorm_cluster_timeouts")
    return int(n.value)


class DeferredGemm:
    """What ``gemm(..., defer_reduce=True)`` returns when the tuned configuration splits K: the fp32 partial slabs are in the split-K
    workspace, the reduce launch has NOT run, ``out`` (the bf16 tensor the GEMM would have written) is still unwritten.  The next
    ``groupnorm`` takes this object as its first source (``pcdm_groupnorm_splitk``): it reduces while it loads, and writes ``out`` iff
    ``store`` (something else -- a residual, a skip -- reads the tensor later).  Nothing else may touch the split-K workspace in between."""

    __slots__ = ("part", "split_k", "M", "N", "Npad", "bias", "rowvec", "ldrv", "rowvec_step", "rowvec_step_stride", "rowvec_step_count", "step_error", "rpb", "residual", "ldr", "out",
                 "store", "keep")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def shape(self):
        return self.out.shape

    def tensor(self) -> torch.Tensor:
        """The bf16 tensor -- valid once the consuming GroupNorm has run with ``store``."""
        assert self.store, "this deferred GEMM's tensor is never written (only its GroupNorm reads it)"
        return self.out


def as_tensor(x: Union[torch.Tensor, "DeferredGemm"]) -> torch.Tensor:
    return x.tensor() if isinstance(x, DeferredGemm) else x


def groupnorm(x1: Union[torch.Tensor, "DeferredGemm"], x2: Optional[torch.Tensor], B: int, HW: int, groups: int, eps: float,
              gamma: torch.Tensor, beta: torch.Tensor, silu: bool, out: torch.Tensor, ws: torch.Tensor) -> torch.Tensor:
    """x1 [B*HW, C1] (+ optional x2 [B*HW, C2]) bf16 -> out [B*HW, C1+C2] bf16.  x1 may be a ``DeferredGemm``."""
    C1 = x1.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    _c(out, BF16); _c(gamma, torch.float32); _c(beta, torch.float32)
    assert ws.numel() >= _lib.lib().pcdm_groupnorm_ws_floats(B, C1 + C2)
    log = LAUNCH_LOG is not None and out.is_cuda
    if log:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if isinstance(x1, DeferredGemm):
        d = x1
        assert d.M == B * HW and d.N == C1 and (d.rowvec is None or d.rpb == HW), "rowvec rows must be the GroupNorm's batch entries"
        sp = _lib.GnSplitKSrc()
        sp.part, sp.split_k, sp.M, sp.N, sp.Npad = d.part, d.split_k, d.M, d.N, d.Npad
        sp.bias, sp.rowvec, sp.ldrv, sp.residual, sp.ldr = d.bias, d.rowvec, d.ldrv, d.residual, d.ldr
        sp.rowvec_step, sp.rowvec_step_stride = d.rowvec_step, d.rowvec_step_stride
        sp.rowvec_step_count, sp.step_error = d.rowvec_step_count, d.step_error
        sp.pre_out, sp.store_pre = _ptr(_c(d.out, BF16)), int(d.store)
        rc = _lib.lib().pcdm_groupnorm_splitk(C.byref(sp), _ptr(x2), C2, B, HW, groups, eps, _ptr(gamma), _ptr(beta), int(silu),
                                             _ptr(out), _ptr(ws), _stream(out))
        _chk(rc, "pcdm_groupnorm_splitk")
        x1 = d.out
    else:
        _c(x1, BF16)
        rc = _lib.lib().pcdm_groupnorm(_ptr(x1), C1, _ptr(x2), C2, B, HW, groups, eps, _ptr(gamma), _ptr(beta),
                                      int(silu), _ptr(out), _ptr(ws), _stream(x1))
        _chk(rc, "pcdm_groupnorm")
    if log:   # algorithmic bytes: the tensor read once + written once
        e1.record()
        LAUNCH_LOG.append(("groupnorm", 2.0 * B * HW * (C1 + C2) * 2, e0, e1, (B, HW, C1 + C2)))
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out: torch.Tensor) -> torch.Tensor:
    rows, Cc = x.shape
    _c(x, BF16); _c(out, BF16)
    log = LAUNCH_LOG is not None and x.is_cuda
    if log:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _chk(_lib.lib().pcdm_layernorm(_ptr(x), _ptr(out), rows, Cc, eps, _ptr(gamma), _ptr(beta), _stream(x)),
         "pcdm_layernorm")
    if log:
        e1.record()
        LAUNCH_LOG.append(("layernorm", 2.0 * rows * Cc * 2, e0, e1, (rows, Cc)))
    return out


# ------------------------------------------------------------------------------------ weights
def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


@dataclass
class PackedWeight:
    """bf16 [Npad, K] (K contiguous) + fp32 bias[Npad] as the GEMM kernel wants them."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    N: int
    K: int
    Npad: int
    cin: int = 0       # conv: (padded) input channels
    geglu: bool = False
    alg_nk: int = 0    # algorithmic N*K (un-padded; GEGLU counts both halves) for FLOP accounting
    exec_nk: int = 0   # N*K the launch really contracts when that is LESS than the algorithmic figure (the phase-decomposed upsample convolution); 0: = alg_nk
    wsum: Optional[torch.Tensor] = None   # fp32 [Npad]: row sums of the packed bf16 weights when they carry a folded LayerNorm


def pack_linear(w: torch.Tensor, bias: Optional[torch.Tensor], device, pad_to: int = 64) -> PackedWeight:
    """nn.Linear / 1x1-conv weight [N, K(,1,1)] -> packed."""
    w = w.reshape(w.shape[0], -1).float()
    N, K = w.shape
    assert K % 64 == 0, K
    Npad = _round_up(N, pad_to)
    wp = torch.zeros(Npad, K, dtype=torch.float32)
    wp[:N] = w
    bp = None
    if bias is not None:
        bp = torch.zeros(Npad, dtype=torch.float32)
        bp[:N] = bias.float()
        bp = bp.to(device)
    return PackedWeight(wp.to(BF16).to(device), bp, N, K, Npad, alg_nk=N * K)


def pack_conv3x3(w: torch.Tensor, bias: Optional[torch.Tensor], device, pad_to: int = 64) -> PackedWeight:
    """Conv2d weight [N, Cin, 3, 3] -> [Npad, 9*Cin_pad] with k = (ky*3+kx)*Cin_pad + c."""
    N, Cin = w.shape[:2]
    Cp = _round_up(Cin, 64)
    wp = torch.zeros(N, 3, 3, Cp, dtype=torch.float32)
    wp[..., :Cin] = w.float().permute(0, 2, 3, 1)
    pw = pack_linear(wp.reshape(N, 9 * Cp), bias, device, pad_to)
    pw.cin = Cp
    pw.alg_nk = N * 9 * Cin
    return pw


def pack_conv3x3_shortcut(w: torch.Tensor, bias: Optional[torch.Tensor], w_sc: torch.Tensor, bias_sc: Optional[torch.Tensor], device) -> PackedWeight:
    """ResnetBlock2D's conv2 [N, C, 3, 3] and conv_shortcut [N, Cx(,1,1)] as ONE contraction: rows [9 C_pad taps | Cx], bias = the sum.  The
    launch reads conv2's input through the taps and the block's input (one tensor or the two halves of the skip concat) through the extra K
    (``gemm(..., conv=..., a2=, a3=)``): out = conv2(h) + conv_shortcut(x), exactly the residual sum of resnet.py's forward."""
    N, Cin = w.shape[:2]
    Cp = _round_up(Cin, 64)
    wp = torch.zeros(N, 3, 3, Cp, dtype=torch.float32)
    wp[..., :Cin] = w.float().permute(0, 2, 3, 1)
    ws = w_sc.reshape(N, -1).float()
    assert ws.shape[1] % 64 == 0, ws.shape
    b = None
    if bias is not None or bias_sc is not None:
        b = (bias.float() if bias is not None else 0) + (bias_sc.float() if bias_sc is not None else 0)
    pw = pack_linear(torch.cat([wp.reshape(N, 9 * Cp), ws], 1), b, device)
    pw.cin = Cp
    pw.alg_nk = N * (9 * Cin + ws.shape[1])
    return pw


UPSAMPLE_TAP_LUT = sum(((3 * (a + i) + b + j) << (16 * (2 * a + b) + 4 * (2 * i + j))) for a in (0, 1) for b in (0, 1) for i in (0, 1) for j in (0, 1))


def pack_upsample_phases(w: torch.Tensor, bias: Optional[torch.Tensor], device) -> PackedWeight:
    """Upsample2D's ``conv3x3(nearest x2 (x))`` [N, C, 3, 3] as FOUR 2x2 convolutions of the low-res input, one per output phase (a, b) =
    (row parity, column parity): output pixel (2y + a, 2x + b) reads low-res rows {y - 1 + a, y + a} and columns {x - 1 + b, x + b}; the taps
    of the 3x3 kernel that land on the same low-res pixel are SUMMED (fp64, then bf16 like any weight).  Rows [phase 2a + b][N], K = [4 taps][C]
    with the tap order (i, j) of ``UPSAMPLE_TAP_LUT``: one launch with ``tap_group_n = N`` on the low-res tensor, 4/9 of the FLOPs of the
    convolution on the upsampled one, then ``pixel_shuffle2``.  Exact algebra; zero padding agrees (the upsampled border row -1 is low-res row -1)."""
    N, Cin = w.shape[:2]
    assert Cin % 64 == 0 and N % 64 == 0, (N, Cin)
    w = w.double()
    rows = {0: ([0], [1, 2]), 1: ([0, 1], [2])}      # phase parity -> (taps summed into window position 0, position 1)
    blocks = []
    for a in (0, 1):
        for b in (0, 1):
            k = torch.zeros(N, 2, 2, Cin, dtype=torch.float64)
            for i in (0, 1):
                for j in (0, 1):
                    k[:, i, j] = sum(w[:, :, ky, kx] for ky in rows[a][i] for kx in rows[b][j])
            blocks.append(k.reshape(N, 4 * Cin))
    pw = pack_linear(torch.cat(blocks, 0).float(), None if bias is None else bias.float().repeat(4), device)
    pw.cin = Cin
    pw.alg_nk = N * 9 * Cin * 4    # algorithmic FLOPs stay those of the 3x3 convolution on the upsampled tensor: 2 (4 M) N 9 C
    pw.exec_nk = 4 * N * 4 * Cin   # ... what the launch executes: 4/9 of them
    return pw


def pixel_shuffle2(x: torch.Tensor, out: torch.Tensor, B: int, H: int, W: int, Cc: int) -> torch.Tensor:
    """[B*H*W, 4*Cc] (phase-major columns) -> NHWC [B, 2H, 2W, Cc] (as [B*2H*2W, Cc]): the second half of the phase-decomposed upsample convolution."""
    assert x.dtype == BF16 and out.dtype == BF16 and x.is_contiguous() and out.is_contiguous() and x.numel() == out.numel() == B * H * W * 4 * Cc
    _chk(_lib.lib().pcdm_pixel_shuffle2(_ptr(x), _ptr(out), B, H, W, Cc, _stream(x)), "pcdm_pixel_shuffle2")
    return out


def pack_geglu(w: torch.Tensor, bias: torch.Tensor, device) -> PackedWeight:
    """GEGLU proj weight [2*D, K] (rows [h | gate]) -> rows interleaved per 64 as [32 h | 32 gate]."""
    w = w.float()
    D2, K = w.shape
    D = D2 // 2
    Dp = _round_up(D, 64)
    wh = torch.zeros(Dp, K); wh[:D] = w[:D]
    wg = torch.zeros(Dp, K); wg[:D] = w[D:]
    bh = torch.zeros(Dp); bh[:D] = bias[:D].float()
    bg = torch.zeros(Dp); bg[:D] = bias[D:].float()
    wp = torch.stack([wh.view(Dp // 32, 32, K), wg.view(Dp // 32, 32, K)], dim=1).reshape(2 * Dp, K)
    bp = torch.stack([bh.view(Dp // 32, 32), bg.view(Dp // 32, 32)], dim=1).reshape(2 * Dp)
    return PackedWeight(wp.to(BF16).to(device), bp.to(device), D, K, 2 * Dp, geglu=True, alg_nk=2 * D * K)


def fold_layernorm(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm(x; gamma, beta) @ w^T + bias  ==  x_hat @ (w diag(gamma))^T + (bias + w beta)  with x_hat = (x - mean) rstd.
    Returns (w diag(gamma), bias + w beta) in fp32."""
    w, gamma, beta = w.float().cpu(), gamma.float().cpu(), beta.float().cpu()
    b = w @ beta
    if bias is not None:
        b = b + bias.float().cpu()
    return w * gamma[None, :], b


def _with_wsum(pw: PackedWeight) -> PackedWeight:
    pw.wsum = pw.w.float().sum(dim=1).contiguous()   # of the bf16 values the MFMA multiplies, in the packed row order
    return pw


def pack_linear_ln(w, bias, gamma, beta, device, pad_to: int = 64) -> PackedWeight:
    """``pack_linear`` of a Linear that follows a LayerNorm, with the LayerNorm folded in (pcdm_gemm_params.ln_wsum)."""
    wf, bf = fold_layernorm(w, bias, gamma, beta)
    return _with_wsum(pack_linear(wf, bf, device, pad_to))


def pack_geglu_ln(w, bias, gamma, beta, device) -> PackedWeight:
    wf, bf = fold_layernorm(w, bias, gamma, beta)
    return _with_wsum(pack_geglu(wf, bf, device))


# ------------------------------------------------------------------------------------ GEMM / conv
def gemm(a: torch.Tensor, pw: PackedWeight, out: torch.Tensor, *, a2: Optional[torch.Tensor] = None, a3: Optional[torch.Tensor] = None,
         rowvec: Optional[torch.Tensor] = None, rows_per_batch: Optional[int] = None,
         residual: Optional[torch.Tensor] = None, res_mod: int = 0, epilogue: int = EPI_STORE,
         out2: Optional[torch.Tensor] = None, vt_col0: int = 0, conv: Optional[dict] = None, tile: int = 0,
         use_bias: bool = True, split_k: int = 1, w_ld: int = 0, act: int = ACT_NONE, zero_rows: int = 0,
         ln: Optional[tuple] = None, ln_buf: Optional[torch.Tensor] = None, pw_ln: Optional[PackedWeight] = None,
         defer_reduce: Optional[bool] = None, dup_rows: int = 0, rowvec_step: Optional[torch.Tensor] = None,
         rowvec_step_stride: int = 0, row_stats: Optional[torch.Tensor] = None, rowvec_step_count: int = 0,
         step_error: Optional[torch.Tensor] = None, tap_lut: int = 0, tap_group_n: int = 0) -> Union[torch.Tensor, "DeferredGemm"]:
    """out = epilogue(A @ W^T).  ``a`` [M, K1] (linear; optional ``a2`` [M, K2] = channel concat) or
    NHWC [B, Hi, Wi, cin] with ``conv=dict(B,Hi,Wi,Ho,Wo,stride,upsample)``.  A convolution whose packed weight has EXTRA K behind the nine
    taps (``pack_conv3x3_shortcut``: K = 9 cin + cx) contracts on over a 1x1 convolution of ``a2`` [M, c1] (and ``a3`` [M, cx - c1]) at the
    output pixel -- ResnetBlock2D's conv2 + conv_shortcut over the block's (concatenated) input as one launch (pcdm_gemm_params.a3, ABI 5).

    ``ln=(gamma, beta, eps)``: out = epilogue(LayerNorm(A) @ W^T).  With ``pw_ln`` (the same Linear packed by ``pack_linear_ln`` /
    ``pack_geglu_ln``: LayerNorm folded into the weights) no LayerNorm pass is needed at all: the A-in-registers kernel (tiles 31..36, K = 320; 34
    = four waves, two workgroups per CU: the one in use at level 0) takes the row statistics from the rows it holds, the tiled
    instances of ``LN_TILED_TILES`` (any K: levels 1-3) from the A tiles as they pass through LDS; otherwise the rows go through
    ``pcdm_layernorm`` into ``ln_buf`` [M, K] first (the tuner times all forms, the LayerNorm launch included, and keeps the fastest).

    ``defer_reduce`` (``True``: the reduced tensor is also read by something other than the next GroupNorm -- it gets written by that
    GroupNorm; ``False``-but-not-``None`` i.e. ``0``: only the next GroupNorm reads it): when the configuration in use splits K, skip the
    reduce launch and return a ``DeferredGemm`` for ``ops.groupnorm`` to consume.  The caller promises that the very next user of the
    result is that GroupNorm and that no other split-K GEMM runs in between.  ``None``: never defer.

    ``rowvec_step`` (device int32 counter) / ``rowvec_step_stride`` (floats): the row-vector block in use is ``rowvec + *rowvec_step *
    rowvec_step_stride`` -- the per-step slice of a table that holds the time-embedding projections of every denoise step.  ``rowvec_step_count``
    (> 0: the blocks behind ``rowvec``) bounds the counter ON THE DEVICE: a value outside ``[0, count)`` is clamped into the table and ``step_error``
    (device int32) set to 1 -- the launch never reads beyond the table (pcdm_gemm_params.rowvec_step_count, ABI 4).

    ``dup_rows`` (conv only): ``out`` has ``M + dup_rows`` rows; rows ``m + dup_rows`` get the same contraction with THEIR row-vector /
    residual rows (``pcdm_gemm_params.dup_rows``: the CFG-shared prefix of the UNet)."""
    if ln is not None and conv is None and a2 is None and ln_buf is not None:
        return _gemm_ln(a, pw, pw_ln, out, ln, ln_buf, rows_per_batch=rows_per_batch, epilogue=epilogue, out2=out2, vt_col0=vt_col0,
                        tile=tile, row_stats=row_stats)
    assert ln is None
    p = GemmParams()
    assert a.dtype == BF16 and a.stride(-1) == 1 and (conv is None or a.is_contiguous())
    p.a = _ptr(a)
    p.ldw = w_ld
    if conv is not None:
        p.conv = 1
        p.B, p.Hi, p.Wi, p.Ho, p.Wo = conv["B"], conv["Hi"], conv["Wi"], conv["Ho"], conv["Wo"]
        p.stride, p.upsample, p.cin = conv.get("stride", 1), conv.get("upsample", 0), pw.cin
        p.no_pad_lo = conv.get("no_pad_lo", 0)
        assert a.numel() == p.B * p.Hi * p.Wi * pw.cin, (a.shape, pw.cin)
        M = p.B * p.Ho * p.Wo
        if tap_group_n:      # a subset of the nine taps per output-channel group (pack_upsample_phases; pcdm_gemm_params.tap_lut)
            assert a2 is None and a3 is None and pw.K % pw.cin == 0 and pw.K // pw.cin <= 4, (pw.K, pw.cin)
            p.tap_lut, p.tap_group_n = int(tap_lut), int(tap_group_n)
        elif a2 is not None:   # extra K: the 1x1 sources behind the taps
            assert a2.dtype == BF16 and a2.dim() == 2 and a2.shape[0] == M and a2.stride(-1) == 1
            p.a2, p.lda2, p.c1 = _ptr(a2), a2.stride(0), a2.shape[1]
            cx = a2.shape[1]
            if a3 is not None:
                assert a3.dtype == BF16 and a3.dim() == 2 and a3.shape[0] == M and a3.stride(-1) == 1
                p.a3, p.lda3 = _ptr(a3), a3.stride(0)
                cx += a3.shape[1]
            assert pw.K == 9 * pw.cin + cx, (pw.K, pw.cin, cx)
        else:
            assert a3 is None and pw.K == 9 * pw.cin, (pw.K, pw.cin)
    else:
        M = a.shape[0]
        p.lda = a.stride(0)
        p.c1 = a.shape[1]
        assert a3 is None
        if a2 is not None:
            assert a2.dtype == BF16 and a2.stride(-1) == 1
            p.a2 = _ptr(a2)
            p.lda2 = a2.stride(0)
            assert a.shape[1] + a2.shape[1] == pw.K
        else:
            assert a.shape[1] == pw.K, (a.shape, pw.K)
    p.w = _ptr(pw.w)
    p.M, p.N, p.K, p.Npad = M, pw.N, pw.K, pw.Npad
    p.bias = _ptr(pw.bias) if (use_bias and pw.bias is not None) else None
    p.rows_per_batch = rows_per_batch or M
    if rowvec is not None:
        assert rowvec.dtype == torch.float32 and rowvec.shape[-1] == pw.N and rowvec.stride(-1) == 1
        p.rowvec = _ptr(rowvec)
        p.ldrv = rowvec.stride(0)
        if rowvec_step is not None:
            assert rowvec_step.dtype == torch.int32 and rowvec_step.device == rowvec.device
            p.rowvec_step, p.rowvec_step_stride = _ptr(rowvec_step), int(rowvec_step_stride)
            p.rowvec_step_count = int(rowvec_step_count)
            if step_error is not None:
                assert step_error.dtype == torch.int32 and step_error.device == rowvec.device
                p.step_error = _ptr(step_error)
    if residual is not None:
        _c(residual, BF16)
        p.residual = _ptr(residual)
        p.ldr = residual.stride(0) if residual.dim() == 2 else pw.N
        p.res_mod = res_mod
    p.epilogue = epilogue
    p.act = act
    p.zero_rows = zero_rows   # (linear) A rows < zero_rows are declared all-zero: never read
    p.dup_rows = dup_rows
    assert not dup_rows or (conv is not None and out.shape[0] >= M + dup_rows)
    p.vt_col0 = vt_col0
    p.out = _ptr(out)
    p.ldo = out.stride(0) if (out.dim() == 2 and epilogue != EPI_NCHW_F32) else pw.N
    if out2 is not None:
        p.out2 = _ptr(out2)
        p.ldo2 = out2.shape[-1]
    stream = _stream(a)
    split = 1
    key = None
    if tile == 0 and AUTOTUNE:
        key = (M, pw.Npad, pw.K, p.conv, p.stride + (10 if p.no_pad_lo else 0) + (20 if tap_group_n else 0), p.upsample, epilogue,
               a2 is not None, residual is not None) + ((True,) if zero_rows else ((2,) if dup_rows else ()))   # (w_ld does not change the best tile)
        tile, split = _TUNED.get(key, (0, 1))
        if tile in ROWGEMM_TILES and (rowvec is not None or act != ACT_NONE or (residual is not None and res_mod < M)):
            # the key does not carry these (none of the UNet's K = 320 residual linears has them): the A-in-registers kernel the table names for the
            # shape cannot serve this call -- the library's heuristic tile does (a table entry is a preference, never a reason to refuse a launch)
            tile, split = 0, 1
        elif tile == 0 and a.is_cuda and not torch.cuda.is_current_stream_capturing():   # (the emulator build runs the library's heuristic)
            tile, split = _TUNED[key] = _autotune(p, stream, pw, epilogue, a.device, out)
    elif split_k > 1:
        split = split_k
    p.tile = tile
    if row_stats is not None:
        # producer of LayerNorm partials (pcdm_gemm_params.row_stats_out): [M][N / 32][2] fp32, written by the STORE epilogue of the tiles that
        # have such an instance; any other configuration leaves the buffer alone and says so (the consumer then takes its own statistics)
        ok = tile in STATS_TILES and split == 1 and conv is None and epilogue == EPI_STORE and act == ACT_NONE and pw.N % 32 == 0
        _STATS_VALID[row_stats.untyped_storage().data_ptr()] = ok   # (keyed on the storage: a consumer may read a row slice of the buffer)
        if ok:
            assert row_stats.dtype == torch.float32 and row_stats.is_contiguous() and row_stats.numel() >= M * (pw.N // 32) * 2
            p.row_stats_out = _ptr(row_stats)
    deferred = None
    if split > 1:
        ws = _splitk_ws(a.device, split * M * pw.Npad)
        p.split_k, p.ws, p.ws_floats = split, ws.data_ptr(), ws.numel()
        if defer_reduce is not None and DEFER_SPLITK and epilogue == EPI_STORE and act == ACT_NONE and (residual is None or res_mod in (0, M)) \
                and (rowvec is None or (rows_per_batch and M % rows_per_batch == 0)) and out.is_contiguous() and out.shape == (M, pw.N):
            p.defer_reduce = 1
            deferred = DeferredGemm(part=ws.data_ptr(), split_k=split, M=M, N=pw.N, Npad=pw.Npad, bias=p.bias, rowvec=p.rowvec,
                                    ldrv=p.ldrv if rowvec is not None else 0, rowvec_step=p.rowvec_step, rowvec_step_stride=p.rowvec_step_stride,
                                    rowvec_step_count=p.rowvec_step_count, step_error=p.step_error,
                                    rpb=p.rows_per_batch, residual=p.residual, ldr=p.ldr, out=out,
                                    store=bool(defer_reduce), keep=(ws, rowvec, residual, pw, rowvec_step, step_error))
    if LAUNCH_LOG is not None and a.is_cuda:  # bench.py: per-launch HIP events on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _chk(_lib.lib().pcdm_gemm(C.byref(p), stream), "pcdm_gemm")
        e1.record()
        # algorithmic FLOPs stay the un-hoisted 2*M*N*K also when zero_rows skips part of the contraction; the sixth
        # field is what the launch executes
        LAUNCH_LOG.append(("gemm_kernel", 2.0 * M * pw.alg_nk, e0, e1, (M, pw.N, pw.K, bool(conv), tile, split),
                           2.0 * (M - zero_rows) * (pw.exec_nk or pw.alg_nk)))
        if LAUNCH_KEYS is not None and key is not None:
            LAUNCH_KEYS.append((len(LAUNCH_LOG) - 1, key))
        return out if deferred is None else deferred
    _chk(_lib.lib().pcdm_gemm(C.byref(p), stream), "pcdm_gemm")
    return out if deferred is None else deferred


def row_stats_valid(row_stats: Optional[torch.Tensor]) -> bool:
    """did the last ``gemm(..., row_stats=buf)`` on this buffer actually write the partials?"""
    return row_stats is not None and _STATS_VALID.get(row_stats.untyped_storage().data_ptr(), False)


def ln_wants_row_stats(M: int, pw: "PackedWeight", epilogue: int) -> bool:
    """should the producer of the rows of this LayerNorm -> Linear pair write partials (``gemm(..., row_stats=)``)?  Yes when the pair's tuned
    choice merges partials (mode 2), and while the pair is untuned (the tuner can only measure mode 2 if it is handed partials)."""
    if not LN_TILED or pw.K == 320 or pw.K % 64 or pw.K > 1280:
        return False
    ch = _TUNED.get(("ln", M, pw.Npad, pw.K, epilogue))
    return ch is None or (len(ch) > 1 and ch[0] > 0 and ch[1] == 2)


def row_stats_reference(a: torch.Tensor) -> torch.Tensor:
    """[M][K / 32][2] {sum, M2 about the run's own mean} of every 32-column run of ``a`` (torch; the tuner and the tests)."""
    M, K = a.shape
    v = a.float().view(M, K // 32, 32)
    sm = v.sum(-1)
    return torch.stack([sm, ((v - sm.unsqueeze(-1) / 32) ** 2).sum(-1)], -1).contiguous()


def _gemm_ln(a, pw, pw_ln, out, ln, ln_buf, *, rows_per_batch, epilogue, out2, vt_col0, tile, row_stats=None, mode=None):
    """(wrapper: records which entries of LAUNCH_LOG one LayerNorm -> Linear pair produced -- one launch or two, depending on the table -- for
    tools/tune_in_step.py)"""
    if LAUNCH_SPANS is None or LAUNCH_LOG is None:
        return _gemm_ln_impl(a, pw, pw_ln, out, ln, ln_buf, rows_per_batch=rows_per_batch, epilogue=epilogue, out2=out2, vt_col0=vt_col0, tile=tile,
                             row_stats=row_stats, mode=mode)
    p0 = len(LAUNCH_LOG)
    r = _gemm_ln_impl(a, pw, pw_ln, out, ln, ln_buf, rows_per_batch=rows_per_batch, epilogue=epilogue, out2=out2, vt_col0=vt_col0, tile=tile,
                      row_stats=row_stats, mode=mode)
    if not tile:
        LAUNCH_SPANS.append((("ln", a.shape[0], pw.Npad, pw.K, epilogue), p0, len(LAUNCH_LOG)))
    return r


def _gemm_ln_impl(a, pw, pw_ln, out, ln, ln_buf, *, rows_per_batch, epilogue, out2, vt_col0, tile, row_stats=None, mode=None):
    """LayerNorm + GEMM: folded into the A-in-registers kernel (K = 320) or an extended instance of the tiled kernel when that wins for the
    shape, two launches otherwise.  Tuned choice per shape = (tile, mode): mode 1 = the kernel takes the row statistics itself, mode 2 = it
    merges the partials its producer wrote (``row_stats``; falls back to mode 1 on the same tile when the producer could not write them)."""
    gamma, beta, eps = ln
    M = a.shape[0]
    assert a.dtype == BF16 and a.stride(1) == 1 and a.shape[1] == pw.K and ln_buf.shape[0] >= M
    key = ("ln", M, pw.Npad, pw.K, epilogue)
    have_stats = row_stats_valid(row_stats)
    choice = (tile, mode or (2 if have_stats else 1)) if tile else _TUNED.get(key)   # (an explicit tile: partials are used when there are valid ones)
    if not LN_TILED and not tile and pw.K != 320:   # PCDM_LN_TILED=0: A/B switch -- levels 1-3 keep their LayerNorm launches
        pw_ln = None
    stream = _stream(a)

    def fused(t, stats=None):
        p = GemmParams()
        p.a, p.lda, p.c1 = _ptr(a), a.stride(0), a.shape[1]
        p.w, p.M, p.N, p.K, p.Npad = _ptr(pw_ln.w), M, pw_ln.N, pw_ln.K, pw_ln.Npad
        p.bias = _ptr(pw_ln.bias)
        p.rows_per_batch = rows_per_batch or M
        p.epilogue, p.vt_col0 = epilogue, vt_col0
        p.out = _ptr(out)
        p.ldo = out.stride(0)
        if out2 is not None:
            p.out2, p.ldo2 = _ptr(out2), out2.shape[-1]
        p.ln_wsum, p.ln_eps = _ptr(_c(pw_ln.wsum, torch.float32)), float(eps)
        if stats is not None:
            p.ln_row_stats = _ptr(stats)
        p.tile = t
        return _lib.lib().pcdm_gemm(C.byref(p), stream)

    def two_launches():
        n = layernorm(a, gamma, beta, eps, ln_buf[:M])
        return gemm(n, pw, out, rows_per_batch=rows_per_batch, epilogue=epilogue, out2=out2, vt_col0=vt_col0)

    if pw_ln is None or pw_ln.wsum is None:
        assert not tile
        two_launches()
        return out
    if choice is None and AUTOTUNE and a.is_cuda and not torch.cuda.is_current_stream_capturing():
        two_launches()                         # (tunes the plain GEMM of this shape on the way)
        ref = out.float().clone()

        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            e0.record()
            for _ in range(TUNE_ITERS):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1)
        best, best_t = (0, 1), timed(two_launches)
        ref_stats = row_stats_reference(a) if (row_stats is not None and pw.K != 320 and pw.K % 64 == 0 and pw.K <= 1280) else None   # (a caller that has a producer)
        for t in (ROWGEMM_TILES if pw.K == 320 else ()) + LN_TILED_TILES + LN_PARTIALS_TILES:
            for md in (1, 2):
                st_ = None if md == 1 else ref_stats
                if (md == 2 and (st_ is None or t in ROWGEMM_TILES)) or fused(t, st_) != 0:   # (the library refuses what a tile cannot do: Npad not a
                    continue                                                                    #  multiple of its BN, GEGLU on a 32-wide wave tile, ...)
                if not bool(((out.float() - ref).abs().max() <= 3e-2 * ref.abs().max() + 1e-3).item()):
                    continue
                tt = timed(lambda: fused(t, st_))
                if tt < best_t:
                    best, best_t = (t, md), tt
        choice = _TUNED[key] = best
    md = 1
    if isinstance(choice, tuple):
        choice, md = choice[0], (choice[1] if len(choice) > 1 else 1)
    if choice and pw_ln is not None and pw_ln.wsum is not None:
        stats = row_stats if (md == 2 and have_stats) else None     # (mode 2 without valid partials: the same tile takes its own statistics)
        fused_ = lambda t: fused(t, stats)  # noqa: E731
        # The table key does not hold rows_per_batch: (2816, 3840, 1280, q|k|v^T) is level 2 at UNet batch 8 (352 tokens per image: fine) and
        # level 3 at batch 32 (88 tokens: the folded instances need a multiple of 32 for the V^T pass and refuse).  A refusal (-1) comes
        # before anything is launched: take the two-launch form then -- also inside a graph capture.
        log = LAUNCH_LOG is not None and a.is_cuda
        if log:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = fused_(choice)
        if rc == -1 and not tile:
            two_launches()
            return out
        _chk(rc, "pcdm_gemm (LayerNorm folded)")
        if log:
            e1.record()
            LAUNCH_LOG.append(("gemm_kernel", 2.0 * M * pw.alg_nk, e0, e1, (M, pw.N, pw.K, False, choice, 1), 2.0 * M * pw.alg_nk))
            if LAUNCH_KEYS is not None and not tile:
                LAUNCH_KEYS.append((len(LAUNCH_LOG) - 1, key))
        return out
    two_launches()
    return out


ROWGEMM_TILES = (31, 32, 33, 34, 35, 36)   # rowgemm.hip (K = 320): id -> (BM, BN) below
STATS_TILES = (2, 4, 5, 6, 7, 8, 10, 18)   # gemm_ext.hip EXT = 3: the tiles whose STORE epilogue can also write LayerNorm partials (row_stats)
_STATS_VALID: dict = {}
LN_TILED = os.environ.get("PCDM_LN_TILED", "1") != "0"
LN_TILED_TILES = (18, 4, 7, 17, 26, 8, 2)   # gemm.hip dispatch_tile_ln(): the tiled instances that take a folded LayerNorm (any K; row statistics
#                                             from the A tiles as they pass through LDS) -- levels 1-3 of the UNet, the prior, the encoders
LN_PARTIALS_TILES = (23,)                     # consumers of producer partials only (mode 2; round 6: the 176-row GEGLU-capable tile)
# gemm.hip dispatch_tile(): id -> (BM, BN)
TILE_SHAPES = {1: (256, 128), 2: (64, 64), 3: (256, 64), 4: (128, 128), 5: (128, 64), 6: (256, 64), 7: (128, 128),
               8: (64, 64), 9: (256, 128), 10: (128, 64), 11: (256, 128), 12: (256, 64), 13: (256, 64), 14: (256, 64),
               15: (128, 64), 16: (512, 64), 17: (256, 256), 18: (128, 128),
               21: (192, 320), 26: (192, 256), 22: (176, 320), 23: (176, 256),
               31: (192, 128), 32: (192, 64), 33: (96, 128), 34: (192, 64), 35: (128, 64), 36: (64, 64)}
_TUNED: dict = {}
_WS: dict = {}


_WS_RETIRED: list = []   # outgrown workspaces are kept: a captured hipGraph may still hold their address
_WS_GEN: dict = {}


def workspace_generation(device) -> int:
    """Changes whenever the split-K workspace of ``device`` is re-allocated (a graph captured before that must be re-captured
    to use the new one; the old one stays allocated, so replaying a stale graph is slow-path-correct, never a use-after-free)."""
    return _WS_GEN.get(torch.device(device), 0)


def _splitk_ws(device, floats: int) -> torch.Tensor:
    """fp32 split-K workspace (one per device; only grows outside graph capture; never freed)."""
    ws = _WS.get(device)
    if ws is None or ws.numel() < floats:
        if ws is not None and device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("split-K workspace would grow during graph capture")
        if ws is not None:
            _WS_RETIRED.append(ws)
        ws = _WS[device] = torch.empty(max(floats, 1 << 24), dtype=torch.float32, device=device)
        _WS_GEN[device] = _WS_GEN.get(device, 0) + 1
    return ws


TUNE_ITERS = 3     # timed launches per candidate (tools/tune_gemm_shapes.py raises it for the committed table)
TUNE_REPEATS = 1
TUNE_COLD = False  # tools/tune_gemm_shapes.py --cold: evict L2 / Infinity Cache before every timed launch (inside the denoise step a
#                    GEMM never finds its 1.74 GB of weights cached; back-to-back launches of one problem do)
_FLUSH: dict = {}


def _flush_caches(device):
    buf = _FLUSH.get(device)
    if buf is None:
        buf = _FLUSH[device] = torch.empty(96 << 20, dtype=torch.float32, device=device)   # 384 MiB > 256 MiB Infinity Cache
    buf.zero_()


def _autotune(p: GemmParams, stream, pw: PackedWeight, epilogue: int, device, out: Optional[torch.Tensor] = None):
    """Time every valid (tile configuration, split-K factor) of gemm.hip on this exact problem (HIP events on
    the launch stream, 1 warm + TUNE_ITERS timed launches each, best of TUNE_REPEATS) and return the fastest.
    Runs once per problem shape, outside graph capture; the launches are idempotent (same inputs, same output)."""
    fn = _lib.lib().pcdm_gemm
    nkt = pw.K // 64
    best, best_t = (0, 1), float("inf")
    ref = None   # output of the first valid configuration: a candidate that disagrees with it is never selected
    for tile, (bm, bn) in TILE_SHAPES.items():
        ntiles = -(-p.M // bm) * (pw.Npad // bn if pw.Npad % bn == 0 else 0)
        if ntiles == 0:
            continue
        splits = [1]
        if tile in ROWGEMM_TILES and (pw.K != 320 or p.conv):
            continue
        if epilogue == EPI_STORE and tile not in ROWGEMM_TILES:   # split K only when the tile grid alone cannot fill the 256 CUs
            splits += [s for s in (2, 3, 4, 6, 8, 12, 16) if ntiles * s <= 1024 and nkt // s >= 4 and ntiles < 512]
        for sk in splits:
            p.tile = tile
            if sk > 1:
                ws = _splitk_ws(device, sk * p.M * pw.Npad)
                p.split_k, p.ws, p.ws_floats = sk, ws.data_ptr(), ws.numel()
            else:
                p.split_k, p.ws, p.ws_floats = 0, None, 0
            if fn(C.byref(p), stream) != 0:   # configuration not valid for this N / epilogue
                break
            if out is not None:
                cur = out.float()
                if ref is None:
                    ref = cur.clone()
                elif not bool(((cur - ref).abs().max() <= 2e-2 * ref.abs().max() + 1e-3).item()):
                    continue
            t = float("inf")
            for _ in range(TUNE_REPEATS):
                if TUNE_COLD:
                    evs = []
                    for _ in range(TUNE_ITERS):
                        _flush_caches(device)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        fn(C.byref(p), stream)
                        e1.record()
                        evs.append((e0, e1))
                    evs[-1][1].synchronize()
                    t = min(t, sum(a.elapsed_time(b) for a, b in evs))
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(TUNE_ITERS):
                    fn(C.byref(p), stream)
                e1.record()
                e1.synchronize()
                t = min(t, e0.elapsed_time(e1))
            if t < best_t:
                best, best_t = (tile, sk), t
    p.split_k, p.ws, p.ws_floats = 0, None, 0
    if best[0] == 0:
        raise RuntimeError("no valid GEMM tile configuration")
    return best


# ---- committed per-shape tuning table (measured on MI355X by tools/tune_gemm_shapes.py); shapes not in the table
# are tuned online at first use.  Keys: "M,Npad,K,conv,stride,upsample,epilogue,two_source,residual".
TUNING_FILE = Path(os.environ.get("PCDM_TUNING_TABLE") or Path(__file__).resolve().parent / "tuning" / "gfx950.json")   # (env: A/B runs)


def load_tuning(path: Path = TUNING_FILE) -> int:
    if not Path(path).exists():
        return 0
    tab = json.loads(Path(path).read_text())
    for k, v in tab.get("gemm", {}).items():
        key = tuple((x == "True") if x in ("True", "False") else (int(x) if x.lstrip("-").isdigit() else x) for x in k.split(","))
        _TUNED[key] = (int(v[0]), int(v[1]))
    return len(tab.get("gemm", {}))


def save_tuning(path: Path = TUNING_FILE, note: str = "") -> None:
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    tab = {"note": note, "gemm": {",".join(str(x) for x in k): list(v) for k, v in sorted(_TUNED.items(), key=lambda kv: str(kv[0]))}}
    Path(path).write_text(json.dumps(tab, indent=0))


if os.environ.get("PCDM_NO_TUNING_TABLE") != "1":   # (the online autotuner alone, for A/B runs)
    load_tuning()


def flash_attn(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, B: int, H: int, Lq: int,
               Lk: int, scale: Optional[float] = None, thr: Optional[float] = None) -> torch.Tensor:
    """q [B*Lq, ldq], k [B*Lk, ldk] (views allowed: row stride = .stride(0)), vt [B, H*64, ldvt], out [B*Lq, ldo]."""
    for t in (q, k, vt, out):
        assert t.dtype == BF16
    assert q.stride(1) == 1 and k.stride(1) == 1 and vt.is_contiguous() and out.stride(1) == 1
    scale = scale if scale is not None else 1.0 / math.sqrt(64)
    log = LAUNCH_LOG is not None and q.is_cuda
    if log:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if thr is None:
        rc = _lib.lib().pcdm_flash_attn(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(vt), vt.shape[-1], _ptr(out),
                                       out.stride(0), B, H, Lq, Lk, scale, _stream(q))
    else:   # explicit lazy-rescale threshold (log2 units; 0 = eager online softmax)
        rc = _lib.lib().pcdm_flash_attn_thr(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(vt), vt.shape[-1], _ptr(out),
                                           out.stride(0), B, H, Lq, Lk, scale, float(thr), _stream(q))
    _chk(rc, "pcdm_flash_attn")
    if log:
        e1.record()
        LAUNCH_LOG.append(("flash_attn_kernel", 4.0 * B * H * Lq * Lk * 64, e0, e1, (B, H, Lq, Lk)))
    return out


FP8 = torch.uint8   # e4m3 bytes (torch.float8_e4m3fn views of these buffers are never computed on by PyTorch)


def quantize_fp8(x: torch.Tensor, out: torch.Tensor, cols: Optional[int] = None, scale: float = 1.0) -> torch.Tensor:
    """bf16 [rows, >= cols] (row stride = .stride(0)) -> e4m3 bytes out [rows, cols_pad]; columns >= cols are zeroed."""
    assert x.dtype == BF16 and out.dtype == FP8 and x.stride(-1) == 1 and out.stride(-1) == 1 and x.dim() == 2 and out.dim() == 2
    rows = x.shape[0]
    cols = cols if cols is not None else x.shape[1]
    _chk(_lib.lib().pcdm_quantize_fp8(_ptr(x), _ptr(out), rows, cols, out.shape[1], x.stride(0), out.stride(0), float(scale), _stream(x)),
         "pcdm_quantize_fp8")
    return out


def flash_attn_fp8(q: torch.Tensor, k8: torch.Tensor, vt8: torch.Tensor, out: torch.Tensor, B: int, H: int, Lq: int, Lk: int,
                   scale: Optional[float] = None, k_descale: float = 1.0, v_descale: float = 1.0, thr: float = 5.0) -> torch.Tensor:
    """q bf16 [B*Lq, ldq]; k8 e4m3 [B*Lk, ldk]; vt8 e4m3 [B, H*64, ldvt]; out bf16 [B*Lq, ldo]."""
    assert q.dtype == BF16 and out.dtype == BF16 and k8.dtype == FP8 and vt8.dtype == FP8
    assert q.stride(1) == 1 and k8.stride(1) == 1 and vt8.is_contiguous() and out.stride(1) == 1
    scale = scale if scale is not None else 1.0 / math.sqrt(64)
    log = LAUNCH_LOG is not None and q.is_cuda
    if log:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib.lib().pcdm_flash_attn_fp8(_ptr(q), q.stride(0), _ptr(k8), k8.stride(0), _ptr(vt8), vt8.shape[-1], _ptr(out), out.stride(0),
                                       B, H, Lq, Lk, scale, float(k_descale), float(v_descale), float(thr), _stream(q))
    _chk(rc, "pcdm_flash_attn_fp8")
    if log:
        e1.record()
        LAUNCH_LOG.append(("flash_attn_fp8_kernel", 4.0 * B * H * Lq * Lk * 64, e0, e1, (B, H, Lq, Lk)))
    return out


# ------------------------------------------------------------------------------------ small ops
def timestep_embedding(t_dev: torch.Tensor, step_dev: Optional[torch.Tensor], out: torch.Tensor, flip: bool = True,
                       shift: float = 0.0) -> torch.Tensor:
    assert t_dev.dtype == torch.int64 and out.dtype == torch.float32
    B, dim = out.shape
    _chk(_lib.lib().pcdm_timestep_embedding(_ptr(t_dev), _ptr(step_dev), _ptr(out), B, dim, int(flip), shift,
                                            _stream(out)), "pcdm_timestep_embedding")
    return out


def timestep_embedding_rows(t_dev: torch.Tensor, out: torch.Tensor, flip: bool = True, shift: float = 0.0) -> torch.Tensor:
    """out[i, :] = Timesteps(t_dev[i]) for a whole timestep table (fp32 [n, dim])."""
    assert t_dev.dtype == torch.int64 and out.dtype == torch.float32 and out.shape[0] == t_dev.numel() and out.is_contiguous()
    _chk(_lib.lib().pcdm_timestep_embedding_rows(_ptr(t_dev), t_dev.numel(), _ptr(out), out.shape[1], int(flip), shift, _stream(out)),
         "pcdm_timestep_embedding_rows")
    return out


def time_class_combine(emb_t: torch.Tensor, cls: Optional[torch.Tensor], out: torch.Tensor, B: int) -> torch.Tensor:
    """out[i * B + b] = bf16(silu(emb_t[i] + cls[b])): emb_t fp32 [n, D], cls fp32 [B, D] or None, out bf16 [n * B, D]."""
    n, D = emb_t.shape
    _c(emb_t, torch.float32); _c(out, BF16)
    assert out.shape == (n * B, D) and (cls is None or (_c(cls, torch.float32).shape == (B, D)))
    _chk(_lib.lib().pcdm_time_class_combine(_ptr(emb_t), _ptr(cls), _ptr(out), n, B, D, _stream(out)), "pcdm_time_class_combine")
    return out


def small_linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *,
                 add: Optional[torch.Tensor] = None, act_in: bool = False, act_out: Union[bool, int] = False) -> torch.Tensor:
    """x fp32 [B,K], w bf16 [N,K] -> out fp32 [B,N].  act_out: True/1 = SiLU before ``add``, 2 = SiLU after it."""
    _c(x, torch.float32); _c(w, BF16); _c(out, torch.float32)
    B, K = x.shape
    N = w.shape[0]
    _chk(_lib.lib().pcdm_small_linear(_ptr(x), _ptr(w), _ptr(bias), _ptr(add), _ptr(out), B, K, N, int(act_in),
                                      int(act_out), _stream(x)), "pcdm_small_linear")
    return out


def assemble_input(latents: torch.Tensor, rep: int, mask: Optional[torch.Tensor], masked: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """latents fp32 [N,4,h,w]; mask [1|rep*N,1,h,w] or None (no mask channel: stage 3); masked [1|rep*N,4,h,w]
    -> out bf16 [rep*N,h,w,cpad]."""
    N, _, h, w = latents.shape
    _c(latents, torch.float32); _c(masked, torch.float32); _c(out, BF16)
    if mask is not None:
        _c(mask, torch.float32)
    cpad = out.shape[-1]
    _chk(_lib.lib().pcdm_assemble_input(_ptr(latents), N, rep, _ptr(mask), 1 if mask is None else mask.shape[0], _ptr(masked),
                                        masked.shape[0], _ptr(out), h, w, cpad, _stream(out)), "pcdm_assemble_input")
    return out


def nchw_to_nhwc_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None, cpad: Optional[int] = None) -> torch.Tensor:
    """NCHW (any float dtype) -> NHWC bf16, channels zero-padded to ``cpad``."""
    B, Cc, H, W = x.shape
    x = _c(x.float().contiguous(), torch.float32)
    cpad = cpad or _round_up(Cc, 8)
    if out is None:
        out = torch.empty(B, H, W, cpad, dtype=BF16, device=x.device)
    assert out.shape[-1] == cpad
    _chk(_lib.lib().pcdm_nchw_f32_to_nhwc_bf16(_ptr(x), _ptr(out), B, Cc, cpad, H * W, _stream(x)), "nchw_to_nhwc")
    return out


def nhwc_bf16_to_nchw(x: torch.Tensor, B: int, Cc: int, H: int, W: int) -> torch.Tensor:
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device)
    _chk(_lib.lib().pcdm_nhwc_bf16_to_nchw_f32(_ptr(_c(x, BF16)), _ptr(out), B, Cc, H * W, _stream(x)), "nhwc_to_nchw")
    return out


def f32_to_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _c(x, torch.float32)
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    _chk(_lib.lib().pcdm_f32_to_bf16(_ptr(x), _ptr(out), x.numel(), _stream(x)), "f32_to_bf16")
    return out


def cfg_step(eps: torch.Tensor, cfg: bool, g: float, x: Optional[torch.Tensor], x_prev: Optional[torch.Tensor],
             coef: Optional[torch.Tensor], step_dev: Optional[torch.Tensor] = None,
             noise: Optional[torch.Tensor] = None, eps_out: Optional[torch.Tensor] = None) -> None:
    n = eps.numel() // (2 if cfg else 1)
    _c(eps, torch.float32)
    _chk(_lib.lib().pcdm_cfg_step(_ptr(eps), int(cfg), g, _ptr(x), _ptr(noise), _ptr(x_prev), _ptr(eps_out),
                                  _ptr(coef), _ptr(step_dev), n, _stream(eps)), "pcdm_cfg_step")


def unipc_step(eps: torch.Tensor, cfg: bool, g: float, x: torch.Tensor, m1: torch.Tensor, m2: torch.Tensor, last: torch.Tensor,
               coef: torch.Tensor, step_dev: Optional[torch.Tensor] = None) -> None:
    """pcdm_unipc_step: CFG combine + one UniPC step in place on (x, m1, m2, last); coef fp32 [steps, 12] on the device."""
    n = x.numel()
    for t in (eps, x, m1, m2, last, coef):
        _c(t, torch.float32)
    assert eps.numel() == (2 * n if cfg else n) and m1.numel() == m2.numel() == last.numel() == n and coef.shape[-1] == 12
    _chk(_lib.lib().pcdm_unipc_step(_ptr(eps), int(cfg), float(g), _ptr(x), _ptr(m1), _ptr(m2), _ptr(last), _ptr(coef),
                                    _ptr(step_dev), n, _stream(x)), "pcdm_unipc_step")


def unclip_step(pred: torch.Tensor, cfg: bool, g: float, x: torch.Tensor, noise: Optional[torch.Tensor], out: torch.Tensor,
                coefs: Sequence[float]) -> torch.Tensor:
    """pcdm_unclip_step: coefs = (p_x, p_e, clip, c_x0, c_x, c_noise, out_scale, out_shift); fp32 tensors."""
    _c(pred, torch.float32); _c(x, torch.float32); _c(out, torch.float32)
    n = x.numel()
    assert pred.numel() == (2 * n if cfg else n) and out.numel() == n and len(coefs) == 8
    if noise is not None:
        assert _c(noise, torch.float32).numel() == n
    carr = (C.c_float * 8)(*[float(c) for c in coefs])
    _chk(_lib.lib().pcdm_unclip_step(_ptr(pred), int(cfg), float(g), _ptr(x), _ptr(noise), _ptr(out), carr, n, _stream(x)),
         "pcdm_unclip_step")
    return out


def unclip_step_dev(pred: torch.Tensor, cfg: bool, g: float, x: torch.Tensor, noise_all: Optional[torch.Tensor], coef: torch.Tensor,
                    step_dev: torch.Tensor) -> None:
    """pcdm_unclip_step_dev: in place on x; coef fp32 [steps, 8], noise_all fp32 [steps, x.numel()] (or None), both on the device."""
    _c(pred, torch.float32); _c(x, torch.float32); _c(coef, torch.float32)
    n = x.numel()
    assert pred.numel() == (2 * n if cfg else n) and coef.shape[-1] == 8
    if noise_all is not None:
        assert _c(noise_all, torch.float32).numel() == coef.shape[0] * n
    _chk(_lib.lib().pcdm_unclip_step_dev(_ptr(pred), int(cfg), float(g), _ptr(x), _ptr(noise_all), _ptr(coef), _ptr(step_dev), n,
                                         _stream(x)), "pcdm_unclip_step_dev")


def lincomb(out: torch.Tensor, xs: Sequence[torch.Tensor], cs: Sequence[float]) -> torch.Tensor:
    n = len(xs)
    assert 1 <= n <= 6 and len(cs) == n
    for t in xs:
        _c(t, torch.float32)
        assert t.numel() == out.numel()
    arr = (C.c_void_p * n)(*[t.data_ptr() for t in xs])
    carr = (C.c_float * n)(*[float(c) for c in cs])
    _chk(_lib.lib().pcdm_lincomb(_ptr(out), n, arr, carr, out.numel(), _stream(out)), "pcdm_lincomb")
    return out


def rescale_noise_cfg(cfg_eps: torch.Tensor, text_eps: torch.Tensor, out: torch.Tensor, guidance_rescale: float) -> torch.Tensor:
    """fp32 [N, ...] tensors; per-sample std matching (ref stage2_inpaint_pipeline.py:52-63)."""
    _c(cfg_eps, torch.float32); _c(text_eps, torch.float32); _c(out, torch.float32)
    N = cfg_eps.shape[0]
    _chk(_lib.lib().pcdm_rescale_noise_cfg(_ptr(cfg_eps), _ptr(text_eps), _ptr(out), N, cfg_eps.numel() // N,
                                           float(guidance_rescale), _stream(out)), "pcdm_rescale_noise_cfg")
    return out


def softmax_rows(s: torch.Tensor, out: torch.Tensor, scale: float) -> torch.Tensor:
    """fp32 [rows, cols] -> bf16 [rows, cols] = softmax(scale * s, dim=-1)."""
    assert s.dtype == torch.float32 and out.dtype == BF16 and s.stride(1) == 1 and out.stride(1) == 1
    rows, cols = s.shape
    _chk(_lib.lib().pcdm_softmax_rows(_ptr(s), _ptr(out), rows, cols, s.stride(0), out.stride(0), float(scale), _stream(s)),
         "pcdm_softmax_rows")
    return out


def advance_step(step_dev: torch.Tensor) -> None:
    _chk(_lib.lib().pcdm_advance_step(_ptr(step_dev), _stream(step_dev)), "pcdm_advance_step")
