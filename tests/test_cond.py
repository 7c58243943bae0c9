"""Per-pair conditioning nets: pcdms_amd.cond (HIP) vs oracle/cond.py (fp32 CPU).

Stated tolerance: bf16 activations through 8 conv+SiLU layers / a 2-layer MLP vs fp32: rel-L2 <= 2e-2.
"""
from __future__ import annotations

import pytest
import torch

from oracle import cond as O
from pcdms_amd import ControlNetConditioningEmbedding, ImageProjModel_p, ops


def _rel(a, b):
    a, b = a.float().cpu(), b.float()
    return ((a - b).norm() / b.norm()).item()


def test_gemm_activation_epilogue(backend):
    """out = act(A W^T + bias) + residual for SiLU / GELU(erf), LDS-staged and direct epilogues, and split-K."""
    dev = backend.device
    g = torch.Generator().manual_seed(0)
    M, K, N = (96, 128, 64) if backend.is_emu else (1000, 640, 320)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.3
    r = torch.randn(M, N, generator=g).to(torch.bfloat16)
    pw = ops.pack_linear(w, b, dev)
    lin = a.float() @ w.to(torch.bfloat16).float().t() + b
    for act, fn in ((ops.ACT_SILU, torch.nn.functional.silu), (ops.ACT_GELU, torch.nn.functional.gelu)):
        for sk in (1, 2):
            out = ops.gemm(a.to(dev), pw, torch.empty(M, N, dtype=torch.bfloat16, device=dev), residual=r.to(dev), res_mod=M,
                           act=act, tile=2, split_k=sk)
            backend.sync()
            ref = fn(lin) + r.float()
            assert (out.float().cpu() - ref).abs().max() <= 2e-2 * ref.abs().max(), (act, sk)
        o32 = torch.empty(1, N, M, dtype=torch.float32, device=dev)   # direct (non-staged) epilogue path
        ops.gemm(a.to(dev), pw, o32, act=act, tile=2, epilogue=ops.EPI_NCHW_F32, rows_per_batch=M)
        backend.sync()
        assert torch.allclose(o32[0].t().cpu(), fn(lin), atol=5e-3, rtol=5e-3)


def test_pose_embedding(backend):
    boc = (16, 32, 96, 256)
    B, H, W = (1, 16, 16) if backend.is_emu else (2, 64, 96)
    sd = O.synth(O.pose_param_shapes(320, 3, boc), seed=1)
    m = ControlNetConditioningEmbedding(320, 3, boc)
    assert m.expected_shapes() == {k: tuple(v) for k, v in O.pose_param_shapes(320, 3, boc)}
    with pytest.raises(RuntimeError):
        m.load_state_dict({"conv_in.weight": torch.zeros(16, 3, 3, 3)})
    m.load_state_dict(sd)
    m.to(backend.device)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(2)) * 2 - 1
    y = m(x.to(backend.device))
    backend.sync()
    ref = O.pose_embedding(sd, x)
    assert y.shape == ref.shape == (B, 320, H // 8, W // 8) and y.dtype == torch.float32
    assert _rel(y, ref) <= 2e-2, _rel(y, ref)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 12, 16))


def test_image_proj_model_p(backend):
    dims = (128, 64, 64) if backend.is_emu else (1536, 768, 1024)
    L = 9 if backend.is_emu else 257
    sd = O.synth(O.image_proj_param_shapes(*dims), seed=3, gain=1.0)
    m = ImageProjModel_p(*dims)
    m.load_state_dict(sd)
    m.to(backend.device)
    x = torch.randn(1, L, dims[0], generator=torch.Generator().manual_seed(4))
    y = m(x.to(backend.device))
    backend.sync()
    ref = O.image_proj_p(sd, x)
    assert y.shape == ref.shape and y.dtype == x.dtype
    assert _rel(y, ref) <= 2e-2, _rel(y, ref)


@pytest.mark.gpu
def test_pose_embedding_full_canvas(gpu_backend):
    """The driver's shapes: pose canvas [1,3,512,704] -> st_pose_f [1,320,64,88] (stage2_batchtest_inpaint_model.py:172-174)."""
    sd = O.synth(O.pose_param_shapes(), seed=5)
    m = ControlNetConditioningEmbedding().to(gpu_backend.device)
    m.load_state_dict(sd)
    x = torch.rand(1, 3, 512, 704, generator=torch.Generator().manual_seed(6)) * 2 - 1
    y = m(x.to(gpu_backend.device))
    ref = O.pose_embedding(sd, x)
    assert y.shape == (1, 320, 64, 88) and _rel(y, ref) <= 2e-2, _rel(y, ref)
