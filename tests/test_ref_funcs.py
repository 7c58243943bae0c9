"""tests/golden/ref_funcs.npz: outputs of pure-torch functions the REFERENCE itself holds, executed from /root/reference by
tests/golden/make_ref_funcs_fixture.py (no oracle, no diffusers arithmetic underneath).

CPU (``-m "not gpu"``): the oracle's restatements reproduce them to fp32 round-off -- this PINS those oracle functions
(``rescale_noise_cfg``, the Attention arithmetic of SURVEY.md Appendix A-7, ``ImageProjModel_p``, ``ImageProjection``) to
reference-executed numbers, and the host helpers (``retrieve_timesteps``, ``split_list_into_chunks``) to reference behaviour.
GPU / emulator (``backend``): the HIP path reproduces them within the bf16 tolerances stated per test.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import cond as OC
from oracle import pipeline as OP
from oracle import unet as OU

Z = np.load(Path(__file__).resolve().parent / "golden" / "ref_funcs.npz")


def _t(k):
    return torch.from_numpy(np.asarray(Z[k]))


def _rel(a, b):
    return ((a.float().cpu() - b).norm() / b.norm()).item()


# ------------------------------------------------------------------------------------------ CPU: oracle / host logic vs the reference
@pytest.mark.parametrize("gr", [0.0, 0.7, 1.0])
def test_oracle_rescale_noise_cfg_vs_reference(gr):
    out = OP.rescale_noise_cfg(_t("rescale_cfg"), _t("rescale_text"), gr)
    assert torch.allclose(out, _t(f"rescale_out_{int(gr * 10):02d}"), atol=1e-6, rtol=1e-6)


def _attn_sd(kind):
    C = Z[f"attn_{kind}_wo"].shape[0]
    sd = {"to_out.0.weight": _t(f"attn_{kind}_wo"), "to_out.0.bias": _t(f"attn_{kind}_bo")}
    if kind == "self":
        w = _t("attn_self_wqkv")
        sd.update({"to_q.weight": w[:C], "to_k.weight": w[C:2 * C], "to_v.weight": w[2 * C:]})
    else:
        wkv = _t("attn_cross_wkv")
        sd.update({"to_q.weight": _t("attn_cross_wq"), "to_k.weight": wkv[:C], "to_v.weight": wkv[C:]})
    return sd


def test_oracle_attention_vs_reference_fused_processor():
    """oracle.unet.attention (Appendix A-7) == the reference's FusedAttnProcessor2_0.__call__ (PCDMs_pipeline.py:59-153):
    self-attention through the fused qkv weight, cross-attention through the fused kv weight, and the NCHW entry."""
    H = int(Z["attn_heads"])
    out = OU.attention(_attn_sd("self"), "", _t("attn_self_x"), None, H)
    assert torch.allclose(out, _t("attn_self_out"), atol=2e-5, rtol=1e-5)
    x4 = _t("attn_self_x4")
    B, C, h, w = x4.shape
    out4 = OU.attention(_attn_sd("self"), "", x4.view(B, C, h * w).transpose(1, 2), None, H).transpose(1, 2).reshape(B, C, h, w)
    assert torch.allclose(out4, _t("attn_self_out4"), atol=2e-5, rtol=1e-5)
    outc = OU.attention(_attn_sd("cross"), "", _t("attn_cross_x"), _t("attn_cross_ctx"), H)
    assert torch.allclose(outc, _t("attn_cross_out"), atol=2e-5, rtol=1e-5)


def test_oracle_image_proj_nets_vs_reference():
    sd = {k[len("ipm_sd."):]: _t(k) for k in Z.files if k.startswith("ipm_sd.")}
    assert torch.allclose(OC.image_proj_p(sd, _t("ipm_x")), _t("ipm_y"), atol=2e-5, rtol=1e-5)
    sd = {k[len("iproj_sd."):]: _t(k) for k in Z.files if k.startswith("iproj_sd.")}
    y = _t("iproj_y")
    assert torch.allclose(OC.image_projection(sd, _t("iproj_x"), y.shape[1]), y, atol=2e-5, rtol=1e-5)


def test_retrieve_timesteps_matches_reference():
    from pcdms_amd.pipeline import retrieve_timesteps

    class SchedPlain:   # the two toy schedulers of the fixture generator
        def set_timesteps(self, num_inference_steps, device=None):
            self.timesteps = torch.arange(num_inference_steps - 1, -1, -1) * 7 + 1

    class SchedCustom(SchedPlain):
        def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
            self.timesteps = torch.tensor(timesteps) if timesteps is not None else torch.arange(num_inference_steps - 1, -1, -1)

    ts, n = retrieve_timesteps(SchedPlain(), 5, "cpu")
    assert n == int(Z["rt_plain_n"]) and torch.equal(ts, _t("rt_plain_ts"))
    ts, n = retrieve_timesteps(SchedCustom(), None, None, timesteps=[900, 500, 100])
    assert n == int(Z["rt_custom_n"]) and torch.equal(ts, _t("rt_custom_ts"))
    with pytest.raises(ValueError) as e:
        retrieve_timesteps(SchedPlain(), None, None, timesteps=[3, 2, 1])
    assert str(e.value).startswith(str(Z["rt_error_prefix"]))
    # the product's own schedulers take no custom schedule, like diffusers 0.24's DDIM / UniPC: same ValueError
    from pcdms_amd.schedulers import DDIMScheduler
    with pytest.raises(ValueError):
        retrieve_timesteps(DDIMScheduler(), None, None, timesteps=[3, 2, 1])


def test_split_list_into_chunks_matches_reference():
    from pcdms_amd.parallel import split_list_into_chunks
    for k in Z.files:
        if k.startswith("chunks_"):
            total, parts = (int(v) for v in k.split("_")[1:])
            chunks = split_list_into_chunks(list(range(total)), parts)
            assert [len(c) for c in chunks] == Z[k].tolist() and sum(chunks, []) == list(range(total))


# ------------------------------------------------------------------------------------------ HIP (emulator on CPU, MI355X with -m gpu)
@pytest.mark.parametrize("gr", [0.0, 0.7, 1.0])
def test_hip_rescale_noise_cfg_vs_reference(backend, gr):
    from pcdms_amd import ops
    a, b = _t("rescale_cfg").to(backend.device), _t("rescale_text").to(backend.device)
    out = ops.rescale_noise_cfg(a, b, torch.empty_like(a), gr)
    backend.sync()
    assert torch.allclose(out.cpu(), _t(f"rescale_out_{int(gr * 10):02d}"), atol=2e-5, rtol=2e-5)   # fp32 kernel


def _hip_attention(dev, x, ctx, wq, wk, wv, wo, bo, H):
    """The product's attention sequence (pcdms_amd/unet.py transformer(): fused projection GEMM with the V^T epilogue ->
    flash_attn -> to_out GEMM) on standalone weights."""
    from pcdms_amd import ops
    B, N, C = x.shape
    xb = x.reshape(B * N, C).to(dev, ops.BF16).contiguous()
    at = torch.empty(B * N, C, dtype=ops.BF16, device=dev)
    if ctx is None:
        qkv = ops.pack_linear(torch.cat([wq, wk, wv], 0), None, dev)
        qk = torch.empty(B * N, 2 * C, dtype=ops.BF16, device=dev)
        vt = torch.zeros(B, C, (N + 7) // 8 * 8, dtype=ops.BF16, device=dev)
        ops.gemm(xb, qkv, qk, rows_per_batch=N, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * C)
        ops.flash_attn(qk[:, :C], qk[:, C:], vt, at, B, H, N, N)
    else:
        L = ctx.shape[1]
        cb = ctx.reshape(B * L, -1).to(dev, ops.BF16).contiguous()
        q = ops.gemm(xb, ops.pack_linear(wq, None, dev), torch.empty(B * N, C, dtype=ops.BF16, device=dev))
        k = torch.empty(B * L, C, dtype=ops.BF16, device=dev)
        vt = torch.zeros(B, C, (L + 7) // 8 * 8, dtype=ops.BF16, device=dev)
        ops.gemm(cb, ops.pack_linear(torch.cat([wk, wv], 0), None, dev), k, rows_per_batch=L, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=C)
        ops.flash_attn(q, k, vt, at, B, H, N, L)
    out = ops.gemm(at, ops.pack_linear(wo, bo, dev), torch.empty(B * N, C, dtype=ops.BF16, device=dev))
    return out.view(B, N, C)


def test_hip_attention_vs_reference_fused_processor(backend):
    """bf16 inputs / weights / P, fp32 accumulation: rel-L2 <= 1.5e-2 against the reference's fp32 numbers."""
    H = int(Z["attn_heads"])
    sd = _attn_sd("self")
    out = _hip_attention(backend.device, _t("attn_self_x"), None, sd["to_q.weight"], sd["to_k.weight"], sd["to_v.weight"],
                         sd["to_out.0.weight"], sd["to_out.0.bias"], H)
    backend.sync()
    assert _rel(out, _t("attn_self_out")) <= 1.5e-2, _rel(out, _t("attn_self_out"))
    sd = _attn_sd("cross")
    out = _hip_attention(backend.device, _t("attn_cross_x"), _t("attn_cross_ctx"), sd["to_q.weight"], sd["to_k.weight"], sd["to_v.weight"],
                         sd["to_out.0.weight"], sd["to_out.0.bias"], H)
    backend.sync()
    assert _rel(out, _t("attn_cross_out")) <= 1.5e-2, _rel(out, _t("attn_cross_out"))


def test_hip_image_proj_nets_vs_reference(backend):
    from pcdms_amd.cond import ImageProjection, ImageProjModel_p
    sd = {k[len("ipm_sd."):]: _t(k) for k in Z.files if k.startswith("ipm_sd.")}
    m = ImageProjModel_p(128, 64, 64)
    m.load_state_dict(sd)
    y = m.to(backend.device)(_t("ipm_x").to(backend.device))
    backend.sync()
    assert _rel(y, _t("ipm_y")) <= 2e-2, _rel(y, _t("ipm_y"))
    sd = {k[len("iproj_sd."):]: _t(k) for k in Z.files if k.startswith("iproj_sd.")}
    yr = _t("iproj_y")
    p = ImageProjection(cross_attention_dim=yr.shape[2], clip_embeddings_dim=Z["iproj_x"].shape[1], num_tokens=yr.shape[1])
    p.load_state_dict(sd)
    y = p.to(backend.device)(_t("iproj_x").to(backend.device))
    backend.sync()
    assert y.shape == yr.shape and _rel(y, yr) <= 2e-2, _rel(y, yr)
