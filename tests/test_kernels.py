"""Per-kernel parity: every C-ABI op of libpcdm vs a plain PyTorch fp32 reference of the same op.

Each test runs under the lane emulator on tiny shapes (CPU suite) and on the MI355X (``-m gpu``) on
larger, awkward shapes (ragged M, non-power-of-two widths such as 88/44/22/11, 258 context tokens).
Inputs are rounded to bf16 first so the only error is fp32 accumulation order + the final bf16
rounding: tolerance = 1% relative + 1% of the output scale (bf16 has 8 mantissa bits: 0.4%).
"""
from __future__ import annotations

import math

import pytest
import torch
import torch.nn.functional as F

from pcdms_amd import ops

BF16 = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF16)


def close(out, ref, tol=1e-2):
    out = out.float().cpu()
    ref = ref.float().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert torch.isfinite(out).all()
    s = ref.abs().max().item() + 1e-6
    err = (out - ref).abs().max().item()
    assert err <= tol * s + tol * 0, f"max err {err:.4g} vs scale {s:.4g}"
    rel = ((out - ref).norm() / (ref.norm() + 1e-12)).item()
    assert rel < tol, f"rel-L2 {rel:.4g}"


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("case", ["single", "concat_straddle", "wide", "fused16", "fused32", "fused1k", "two_pass_big", "rows", "rows_tail"])
def test_groupnorm(backend, case):
    dev = backend.device
    if case == "rows":       # long slabs that no single workgroup holds (emulator: two-kernel path; GPU: cluster kernel), two-source concat
        B, HW, C1, C2, G = (1, 4000, 64, 32, 8) if backend.is_emu else (8, 32 * 44, 640, 320, 32)
    elif case == "rows_tail":  # HW not a multiple of the chunk count, group size 10 (octets straddle groups)
        B, HW, C1, C2, G = (2, 2777, 80, 0, 8) if backend.is_emu else (8, 64 * 88 - 3, 320, 0, 32)
    elif case == "fused16":      # single-pass kernel, 512 threads; group size 10 (octets span two groups)
        B, HW, C1, C2, G = (2, 700, 80, 0, 8) if backend.is_emu else (8, 32 * 44, 640, 640, 32)
    elif case == "fused32":
        B, HW, C1, C2, G = (1, 1500, 40, 40, 8) if backend.is_emu else (8, 32 * 44, 640, 0, 32)
    elif case == "fused1k":    # group size 60 over a two-source concat (emu: 1024-thread single pass; GPU shape: two-pass)
        B, HW, C1, C2, G = (1, 500, 160, 80, 4) if backend.is_emu else (8, 32 * 44, 1280, 640, 32)
    elif case == "two_pass_big":   # slab too large for registers -> stats + apply kernels
        B, HW, C1, C2, G = (1, 3000, 16, 0, 2) if backend.is_emu else (16, 64 * 88, 320, 0, 32)
    elif case == "single":
        B, HW, C1, C2, G = (2, 37, 64, 0, 32) if backend.is_emu else (8, 64 * 88, 320, 0, 32)
    elif case == "concat_straddle":  # group size 30: groups straddle the x1|x2 boundary (up-block 640+320)
        B, HW, C1, C2, G = (2, 19, 64, 32, 32) if backend.is_emu else (8, 32 * 44, 640, 320, 32)
    else:
        B, HW, C1, C2, G = (1, 11, 256, 256, 32) if backend.is_emu else (8, 16 * 22, 1280, 1280, 32)
    x1 = rnd(B * HW, C1, seed=1) + 0.5
    x2 = rnd(B * HW, C2, seed=2) * 2 if C2 else None
    C = C1 + C2
    gamma = torch.rand(C, generator=torch.Generator().manual_seed(3)) + 0.5
    beta = torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.2
    for silu in (False, True):
        out = torch.empty(B * HW, C, dtype=BF16, device=dev)
        ws = ops.groupnorm_ws(B, C, dev)
        ops.groupnorm(x1.to(dev), None if x2 is None else x2.to(dev), B, HW, G, 1e-5, gamma.to(dev), beta.to(dev), silu,
                      out, ws)
        backend.sync()
        xx = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], 1)
        xr = xx.view(B, HW, C).permute(0, 2, 1)  # [B, C, HW]
        ref = F.group_norm(xr, G, gamma, beta, 1e-5)
        if silu:
            ref = F.silu(ref)
        close(out.view(B, HW, C), ref.permute(0, 2, 1))


@pytest.mark.parametrize("case", ["fused", "two_pass", "cluster"])
def test_groupnorm_large_mean(backend, case):
    """Checkpoint-like statistics: every channel offset by 50 standard deviations, per-channel offsets on top, so
    that E[x^2] - mean^2 in fp32 would lose 2500 x of its precision.  All three GroupNorm paths against an fp64 reference on the same
    bf16 inputs: the single-pass kernel (centred two-pass on registers), the two-kernel path (shifted sums + Chan merges) and -- GPU
    only -- the cluster kernel (centred chunk statistics + Chan merge across the partner workgroups)."""
    dev = backend.device
    if case == "fused":
        B, HW, C, G = (2, 300, 80, 8) if backend.is_emu else (8, 32 * 44, 640, 32)
    elif case == "two_pass":
        B, HW, C, G = (1, 3000, 48, 4) if backend.is_emu else (8, 64 * 88, 960, 32)
    else:
        if backend.is_emu:
            pytest.skip("the cluster kernel needs co-resident workgroups: GPU only")
        B, HW, C, G = 8, 64 * 88, 320, 32
    g = torch.Generator().manual_seed(11)
    base = torch.randn(B * HW, C, generator=g)                           # unit spread
    chan = torch.randn(1, C, generator=g) * 3.0                          # per-channel offsets inside a group
    x = (base + chan + 50.0).to(BF16)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 3.0                             # (GroupNorm beta ~ N(0, 3^2))
    out = torch.empty(B * HW, C, dtype=BF16, device=dev)
    ws = ops.groupnorm_ws(B, C, dev)
    ops.groupnorm(x.to(dev), None, B, HW, G, 1e-5, gamma.to(dev), beta.to(dev), False, out, ws)
    backend.sync()
    xr = x.double().view(B, HW, C).permute(0, 2, 1)
    ref = F.group_norm(xr, G, gamma.double(), beta.double(), 1e-5).permute(0, 2, 1)
    # the statistic itself: recover rstd-scaled values and compare tightly (bf16 output rounding is 2^-9 of |out| <= ~12)
    close(out.view(B, HW, C), ref, tol=6e-3)


def test_layernorm(backend):
    dev = backend.device
    for rows, C in ([(9, 64), (13, 320), (7, 640), (3, 1280), (5, 768), (2, 1536), (3, 2048)] if backend.is_emu else
                    [(8 * 5632, 320), (2816, 1280), (703, 640), (12, 2048), (7, 4096), (257, 1536), (257, 768), (1001, 512)]):
        x = rnd(rows, C, seed=5) * 3 + 1
        gamma = torch.rand(C, generator=torch.Generator().manual_seed(6)) + 0.5
        beta = torch.randn(C, generator=torch.Generator().manual_seed(7))
        out = torch.empty(rows, C, dtype=BF16, device=dev)
        ops.layernorm(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, out)
        backend.sync()
        close(out, F.layer_norm(x.float(), (C,), gamma, beta, 1e-5))


@pytest.mark.gpu
def test_groupnorm_cluster_rearm_and_determinism(gpu_backend):
    """The cluster GroupNorm (a slab split over S workgroups that exchange partial statistics through global memory and wait for
    each other) over many back-to-back launches that share ONE workspace with each other AND with the two-kernel path: alternating
    shapes / cluster sizes / grids, differently scaled data every time; every result equal to the reference and bit-identical to the
    first of its (shape, scale) -- the counters re-arm, nothing else ever writes them (round 2: they once shared the area with the
    other launches' partial statistics and the poll fell through), no stale partials are read, and the launch never hangs
    (pytest-timeout would kill it)."""
    dev = gpu_backend.device
    # cluster: grids 256, 256, 128, 128; two-kernel path: (8, 5632, 960) and (3, 5632, 640)
    shapes = [(8, 64 * 88, 320, 32), (8, 32 * 44, 640, 32), (4, 64 * 88, 320, 32), (8, 32 * 44, 320, 32), (8, 64 * 88, 960, 32),
              (3, 64 * 88, 640, 32)]
    ws = ops.groupnorm_ws(16, 4096, dev)
    data, first = {}, {}
    for i, (B, HW, C, G) in enumerate(shapes):
        x = rnd(B * HW, C, seed=90 + i)
        gamma = (torch.rand(C, generator=torch.Generator().manual_seed(3)) + 0.5).to(dev)
        beta = (torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.2).to(dev)
        data[i] = (x, gamma, beta)
    for it in range(48):
        i = (it * 7 + it // 3) % len(shapes)
        k = (it // 2) % 3                        # scale / shift of this launch's data
        B, HW, C, G = shapes[i]
        x0, gamma, beta = data[i]
        x = (x0.float() * (1.0 + 1.5 * k) + 2.0 * k - 1.0).to(BF16).to(dev)
        out = torch.full((B * HW, C), float("nan"), dtype=BF16, device=dev)
        ops.groupnorm(x, None, B, HW, G, 1e-5, gamma, beta, True, out, ws)
        if (i, k) in first:
            assert torch.equal(out, first[i, k]), (it, i, k)
        else:
            first[i, k] = out.clone()
            ref = F.silu(F.group_norm(x.float().cpu().view(B, HW, C).permute(0, 2, 1), G, gamma.cpu(), beta.cpu(), 1e-5))
            close(out.view(B, HW, C), ref.permute(0, 2, 1))


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_groupnorm_cluster_under_contention(gpu_backend):
    """The in-launch-exchange GroupNorm while ANOTHER stream keeps the CUs busy with long kernels (the co-residency of a cluster's
    workgroups is then not guaranteed by a plain launch -- VERDICT r2 #8 / ADVICE r2): the launch must not hang (bounded poll, the
    workgroup then computes the slab's statistics alone) and the result must stay the reference's.  Launches that saw no time-out are
    bit-identical to the uncontended result; the counter says how many workgroups gave up waiting."""
    dev = gpu_backend.device
    B, HW, C, G = 8, 64 * 88, 320, 32                      # level 0: 128 slabs x S = 2 workgroups = one per CU
    x = rnd(B * HW, C, seed=411).to(dev)
    gamma = (torch.rand(C, generator=torch.Generator().manual_seed(3)) + 0.5).to(dev)
    beta = (torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.2).to(dev)
    ws = ops.groupnorm_ws(8, 4096, dev)
    quiet = torch.empty(B * HW, C, dtype=BF16, device=dev)
    ops.groupnorm(x, None, B, HW, G, 1e-5, gamma, beta, True, quiet, ws)
    torch.cuda.synchronize()
    ref = F.silu(F.group_norm(x.float().cpu().view(B, HW, C).permute(0, 2, 1), G, gamma.cpu(), beta.cpu(), 1e-5))
    close(quiet.view(B, HW, C), ref.permute(0, 2, 1))
    assert ops.groupnorm_cluster_timeouts(ws) == 0
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=dev, dtype=BF16)
    outs = []
    for rounds in (4, 40):                                  # ~4 ms and ~40 ms of back-to-back 8192^3 GEMMs on the other stream
        with torch.cuda.stream(side):
            for _ in range(rounds):
                torch.mm(a, a)
        for _ in range(6):
            out = torch.full((B * HW, C), float("nan"), dtype=BF16, device=dev)
            ops.groupnorm(x, None, B, HW, G, 1e-5, gamma, beta, True, out, ws)
            outs.append(out)
        torch.cuda.synchronize()
    t = ops.groupnorm_cluster_timeouts(ws)
    print("cluster GroupNorm under contention: workgroups that timed out and computed alone:", t)
    for out in outs:
        if t == 0:
            assert torch.equal(out, quiet)
        close(out.view(B, HW, C), ref.permute(0, 2, 1))
    # the counters are re-armed whatever happened: a quiet launch afterwards is the quiet result again
    again = torch.empty_like(quiet)
    ops.groupnorm(x, None, B, HW, G, 1e-5, gamma, beta, True, again, ws)
    torch.cuda.synchronize()
    assert torch.equal(again, quiet)


@pytest.mark.gpu
def test_groupnorm_cluster_no_stale_partials_back_to_back(gpu_backend):
    """Cluster GroupNorm launches enqueued BACK TO BACK on one workspace, each with differently distributed data, nothing in between
    to flush the L2s: the partial statistics a workgroup reads must be the ones its partners wrote in THIS launch (the hand-off uses
    system-scope stores / loads and no fences; the per-XCD L2s are not coherent with each other, and the same workspace lines are
    re-read launch after launch)."""
    dev = gpu_backend.device
    B, HW, C, G = 2, 32 * 44, 320, 32           # 16 slabs x S = 2..8 workgroups: small enough that its own traffic leaves the L2s alone
    ws = ops.groupnorm_ws(16, 4096, dev)
    gamma = (torch.rand(C, generator=torch.Generator().manual_seed(3)) + 0.5).to(dev)
    beta = (torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.2).to(dev)
    n = 24
    xs = [(rnd(B * HW, C, seed=200 + i) * (1.0 + 0.7 * (i % 5)) + 1.5 * ((i * 3) % 7 - 3)).to(dev) for i in range(n)]
    outs = [torch.empty(B * HW, C, dtype=BF16, device=dev) for _ in range(n)]
    for rep in range(3):
        for o in outs:
            o.fill_(float("nan"))
        torch.cuda.synchronize()
        for x, o in zip(xs, outs):               # enqueue only
            ops.groupnorm(x, None, B, HW, G, 1e-5, gamma, beta, True, o, ws)
        torch.cuda.synchronize()
        for i, (x, o) in enumerate(zip(xs, outs)):
            ref = F.silu(F.group_norm(x.float().cpu().view(B, HW, C).permute(0, 2, 1), G, gamma.cpu(), beta.cpu(), 1e-5))
            close(o.view(B, HW, C), ref.permute(0, 2, 1))


# ------------------------------------------------------------------------------------------------ GEMM
def _gemm_sizes(backend):
    # (M, K, N, tile)
    if backend.is_emu:
        return [(70, 128, 64, 2), (200, 64, 192, 3), (130, 128, 128, 1), (300, 192, 128, 4), (97, 320, 64, 5),
                (150, 192, 64, 6), (140, 256, 128, 7), (90, 320, 64, 8), (260, 192, 128, 9), (100, 256, 64, 10),
                (300, 320, 128, 11), (270, 64, 128, 11), (130, 128, 64, 12), (257, 448, 64, 12),
                (300, 192, 64, 13), (280, 256, 128, 14), (150, 128, 64, 15), (600, 128, 64, 16),
                (300, 128, 256, 17), (200, 192, 128, 18)]
    return [(45056, 320, 320, 0), (2816, 1280, 1280, 0), (704, 1280, 1280, 0), (11264, 640, 1920, 0),
            (1000, 192, 320, 0), (999, 64, 128, 1), (999, 128, 128, 4)] + [(777, 2560, 640, t) for t in range(1, 17)] + [(2816, 1280, 1280, 17), (777, 640, 256, 17), (777, 2560, 640, 18)]


def test_gemm_linear_bias_residual_rowvec(backend):
    dev = backend.device
    for (M, K, N, tile) in _gemm_sizes(backend):
        rpb = 7 if M % 7 == 0 else (M // 2 if M % 2 == 0 else M)
        a = rnd(M, K, seed=10)
        w = rnd(N, K, seed=11, scale=1 / math.sqrt(K))
        bias = torch.randn(N, generator=torch.Generator().manual_seed(12))
        res = rnd(M, N, seed=13)
        rowvec = torch.randn(M // rpb, N, generator=torch.Generator().manual_seed(14))
        pw = ops.pack_linear(w.float(), bias, dev)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        ops.gemm(a.to(dev), pw, out, rowvec=rowvec.to(dev), rows_per_batch=rpb, residual=res.to(dev), res_mod=M, tile=tile)
        backend.sync()
        ref = a.float() @ w.float().t() + bias + res.float() + rowvec.repeat_interleave(rpb, 0)
        close(out, ref)


def test_table_named_rowgemm_tile_is_a_preference(backend, monkeypatch):
    """The tuning key carries neither the row vector nor res_mod.  When the table names the A-in-registers kernel (tiles 31..36: K = 320, no
    row vector, residual rows for every output row) for a shape and a call of that shape asks for more, the launch takes the library's
    heuristic tile instead of failing -- round 6's in-step pass moved the (45056, 320, 320) + residual key to tile 34 and the full-size
    row-vector case of test_gemm_linear_bias_residual_rowvec was refused (-1) until ops.gemm / unet_ctx.hip learnt this."""
    dev = backend.device
    M, K, N = (96, 320, 64) if backend.is_emu else (45056, 320, 320)
    a, w = rnd(M, K, seed=210), rnd(N, K, seed=211, scale=1 / math.sqrt(K))
    bias = torch.randn(N, generator=torch.Generator().manual_seed(212))
    res, rowvec = rnd(M, N, seed=213), torch.randn(2, N, generator=torch.Generator().manual_seed(214))
    pw = ops.pack_linear(w.float(), bias, dev)
    key = (M, pw.Npad, K, 0, 0, 0, ops.EPI_STORE, False, True)
    monkeypatch.setitem(ops._TUNED, key, (34, 1))
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm(a.to(dev), pw, out, rowvec=rowvec.to(dev), rows_per_batch=M // 2, residual=res.to(dev), res_mod=M)
    backend.sync()
    close(out, a.float() @ w.float().t() + bias + res.float() + rowvec.repeat_interleave(M // 2, 0))
    # ... and without the row vector the table's tile runs (same key)
    out2 = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm(a.to(dev), pw, out2, residual=res.to(dev), res_mod=M)
    backend.sync()
    close(out2, a.float() @ w.float().t() + bias + res.float())


def test_gemm_split_k(backend):
    """K split over several workgroups per tile + fp32 reduce kernel (small-M / huge-K level-3 convs)."""
    dev = backend.device
    cases = [(70, 256, 64, 2, 2), (130, 512, 128, 4, 3)] if backend.is_emu else \
        [(704, 11520, 1280, 2, 8), (704, 23040, 1280, 4, 6), (2816, 11520, 1280, 1, 2), (999, 1280, 320, 5, 3)]
    for (M, K, N, tile, sk) in cases:
        a = rnd(M, K, seed=15)
        w = rnd(N, K, seed=16, scale=1 / math.sqrt(K))
        bias = torch.randn(N, generator=torch.Generator().manual_seed(17))
        res = rnd(M, N, seed=18)
        rowvec = torch.randn(1, N, generator=torch.Generator().manual_seed(19))
        pw = ops.pack_linear(w.float(), bias, dev)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        ops.gemm(a.to(dev), pw, out, rowvec=rowvec.to(dev), rows_per_batch=M, residual=res.to(dev), res_mod=M,
                 tile=tile, split_k=sk)
        backend.sync()
        close(out, a.float() @ w.float().t() + bias + res.float() + rowvec)


def test_gemm_two_source_and_broadcast_residual(backend):
    dev = backend.device
    M, K1, K2, N = (96, 64, 128, 64) if backend.is_emu else (11264, 1280, 640, 640)
    a1, a2 = rnd(M, K1, seed=20), rnd(M, K2, seed=21)
    w = rnd(N, K1 + K2, seed=22, scale=1 / math.sqrt(K1 + K2))
    bias = torch.randn(N, generator=torch.Generator().manual_seed(23))
    rm = M // 2
    res = rnd(rm, N, seed=24)  # broadcast over 2 "batches" (pose feature add)
    pw = ops.pack_linear(w.float(), bias, dev)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm(a1.to(dev), pw, out, a2=a2.to(dev), residual=res.to(dev), res_mod=rm)
    backend.sync()
    ref = torch.cat([a1, a2], 1).float() @ w.float().t() + bias + res.float().repeat(2, 1)
    close(out, ref)


def test_gemm_geglu(backend):
    dev = backend.device
    M, K, D = (70, 64, 128) if backend.is_emu else (11264, 640, 2560)
    a = rnd(M, K, seed=30)
    w = rnd(2 * D, K, seed=31, scale=1 / math.sqrt(K))
    bias = torch.randn(2 * D, generator=torch.Generator().manual_seed(32)) * 0.5
    pw = ops.pack_geglu(w.float(), bias, dev)
    out = torch.empty(M, D, dtype=BF16, device=dev)
    ops.gemm(a.to(dev), pw, out, epilogue=ops.EPI_GEGLU)
    backend.sync()
    pr = a.float() @ w.float().t() + bias
    h, g = pr.chunk(2, -1)
    close(out, h * F.gelu(g))


def test_gemm_split_vt_and_nchw(backend):
    dev = backend.device
    B, L, K, Cc = (2, 37, 64, 64) if backend.is_emu else (8, 1408, 640, 640)
    M = B * L
    Lp = (L + 7) // 8 * 8
    a = rnd(M, K, seed=40)
    w = rnd(3 * Cc, K, seed=41, scale=1 / math.sqrt(K))
    pw = ops.pack_linear(w.float(), None, dev)
    qk = torch.empty(M, 2 * Cc, dtype=BF16, device=dev)
    vt = torch.zeros(B, Cc, Lp, dtype=BF16, device=dev)
    ops.gemm(a.to(dev), pw, qk, rows_per_batch=L, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * Cc)
    backend.sync()
    ref = a.float() @ w.float().t()
    close(qk, ref[:, : 2 * Cc])
    close(vt[:, :, :L], ref[:, 2 * Cc:].view(B, L, Cc).permute(0, 2, 1))
    assert (vt[:, :, L:] == 0).all()
    # token counts that are multiples of 32 take the staged epilogue: q | k as 16-byte row stores, V^T as 16-byte stores along the token
    # axis (transposed read-back from LDS); several tile shapes incl. one whose N tiles straddle vt_col0 (256-wide tiles, vt_col0 = 128:
    # the waves of one workgroup take different paths) and a bias
    for (B2, L2, C2, tiles) in ([(2, 64, 64, (2, 3, 4)), (3, 32, 128, (4, 17, 13))] if backend.is_emu else
                                [(8, 1408, 640, (0, 3, 4, 13, 17, 18, 21)), (2, 352, 1280, (17, 21, 26)), (8, 5632, 320, (21, 13))]):
        a2 = rnd(B2 * L2, K, seed=44)
        w2 = rnd(3 * C2, K, seed=45, scale=1 / math.sqrt(K))
        b2 = torch.randn(3 * C2, generator=torch.Generator().manual_seed(46))
        pw2 = ops.pack_linear(w2.float(), b2, dev)
        ref2 = a2.float() @ w2.float().t() + b2
        for tile in tiles:
            if tile and pw2.Npad % ops.TILE_SHAPES[tile][1]:
                continue
            qk2 = torch.zeros(B2 * L2, 2 * C2, dtype=BF16, device=dev)
            vt2 = torch.zeros(B2, C2, L2 + 8, dtype=BF16, device=dev)
            ops.gemm(a2.to(dev), pw2, qk2, rows_per_batch=L2, epilogue=ops.EPI_SPLIT_VT, out2=vt2, vt_col0=2 * C2, tile=tile)
            backend.sync()
            close(qk2, ref2[:, : 2 * C2])
            close(vt2[:, :, :L2], ref2[:, 2 * C2:].view(B2, L2, C2).permute(0, 2, 1))
            assert (vt2[:, :, L2:] == 0).all(), tile
    # NCHW fp32 output with N = 4 (conv_out shape class)
    w4 = rnd(4, K, seed=42, scale=1 / math.sqrt(K))
    b4 = torch.randn(4, generator=torch.Generator().manual_seed(43))
    pw4 = ops.pack_linear(w4.float(), b4, dev)
    o4 = torch.empty(B, 4, L, dtype=torch.float32, device=dev)
    ops.gemm(a.to(dev), pw4, o4, rows_per_batch=L, epilogue=ops.EPI_NCHW_F32)
    backend.sync()
    ref4 = (a.float() @ w4.float().t() + b4).view(B, L, 4).permute(0, 2, 1)
    close(o4, ref4, tol=2e-3)


def test_gemm_narrow_tiles_geglu_and_conv(backend):
    """tiles 13-16 (BN = 64, one wave column, 64x64 wave tiles): GEGLU pairs inside one wave, conv gather, M tails."""
    dev = backend.device
    M, K, D = (70, 64, 128) if backend.is_emu else (11264, 640, 2560)
    a = rnd(M, K, seed=30)
    w = rnd(2 * D, K, seed=31, scale=1 / math.sqrt(K))
    bias = torch.randn(2 * D, generator=torch.Generator().manual_seed(32)) * 0.5
    pw = ops.pack_geglu(w.float(), bias, dev)
    pr = a.float() @ w.float().t() + bias
    h, g = pr.chunk(2, -1)
    for tile in (13, 16, 26) if backend.is_emu else (13, 14, 15, 16, 26):
        if pw.Npad % ops.TILE_SHAPES[tile][1]:
            continue
        out = torch.empty(M, D, dtype=BF16, device=dev)
        ops.gemm(a.to(dev), pw, out, epilogue=ops.EPI_GEGLU, tile=tile)
        backend.sync()
        close(out, h * F.gelu(g))
    B, H, W, Cin, Cout = (2, 6, 5, 64, 64) if backend.is_emu else (8, 32, 44, 640, 320)
    x = rnd(B, Cin, H, W, seed=50)
    wc = rnd(Cout, Cin, 3, 3, seed=51, scale=1 / math.sqrt(9 * Cin))
    bc = torch.randn(Cout, generator=torch.Generator().manual_seed(52))
    ref = F.conv2d(x.float(), wc.float(), bc, padding=1).permute(0, 2, 3, 1)
    pwc = ops.pack_conv3x3(wc.float(), bc, dev)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    for tile in (13, 14, 15, 16):
        out = torch.empty(B * H * W, Cout, dtype=BF16, device=dev)
        ops.gemm(xh, pwc, out, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), tile=tile)
        backend.sync()
        close(out.view(B, H, W, Cout), ref)


@pytest.mark.parametrize("mode", ["s1", "s2", "up", "up_size"])
def test_conv3x3(backend, mode):
    dev = backend.device
    if backend.is_emu:
        B, H, W, Cin, Cout = 2, 6, 5, 64, 64
    else:
        B, H, W, Cin, Cout = (8, 32, 44, 640, 640) if mode not in ("up", "up_size") else (8, 16, 22, 1280, 1280)
    x = rnd(B, Cin, H, W, seed=50)
    w = rnd(Cout, Cin, 3, 3, seed=51, scale=1 / math.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(52))
    xh = x.permute(0, 2, 3, 1).contiguous()
    if mode == "s1":
        Ho, Wo, st, up = H, W, 1, 0
        ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    elif mode == "s2":
        Ho, Wo, st, up = (H - 1) // 2 + 1, (W - 1) // 2 + 1, 2, 0
        ref = F.conv2d(x.float(), w.float(), bias, stride=2, padding=1)
    elif mode == "up":
        Ho, Wo, st, up = 2 * H, 2 * W, 1, 1
        ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), bias, padding=1)
    else:   # Upsample2D(output_size=skip size): latents not divisible by 2**num_upsamplers (odd skip sizes)
        Ho, Wo, st, up = 2 * H - 1, 2 * W - 1, 1, 1
        ref = F.conv2d(F.interpolate(x.float(), size=(Ho, Wo), mode="nearest"), w.float(), bias, padding=1)
    pw = ops.pack_conv3x3(w.float(), bias, dev)
    temb = torch.randn(B, Cout, generator=torch.Generator().manual_seed(53))
    out = torch.empty(B * Ho * Wo, Cout, dtype=BF16, device=dev)
    ops.gemm(xh.to(dev), pw, out, conv=dict(B=B, Hi=H, Wi=W, Ho=Ho, Wo=Wo, stride=st, upsample=up),
             rowvec=temb.to(dev), rows_per_batch=Ho * Wo)
    backend.sync()
    ref = ref + temb[:, :, None, None]
    close(out.view(B, Ho, Wo, Cout), ref.permute(0, 2, 3, 1))


@pytest.mark.parametrize("case", ["one_source", "two_sources", "split_k", "tiles"])
def test_conv3x3_with_shortcut_k(backend, case):
    """ResnetBlock2D's ``conv2(h) + conv_shortcut(x)`` as ONE contraction (pcdm_gemm_params.a3, ABI 5; ops.pack_conv3x3_shortcut): the 1x1
    shortcut over the block's input -- one tensor, or the two halves [x | skip] of the up path's concat, as row-strided views -- is
    K-concatenated behind the nine taps.  Against F.conv2d of both layers in fp32; ``tiles``: every full-size tile family the UNet's
    conv2 launches use gives the same answer to rounding; ``split_k``: the extra K-tiles may fall into any K slice."""
    dev = backend.device
    if backend.is_emu:
        B, H, W, C, C1, C2 = 2, 6, 5, 64, 128, 64
    else:
        B, H, W, C, C1, C2 = (8, 64, 88, 320, 320, 320) if case != "split_k" else (8, 16, 22, 1280, 1280, 640)
    if case == "one_source":
        C2 = 0
    M = B * H * W
    h = rnd(B, C, H, W, seed=250)
    w = rnd(C, C, 3, 3, seed=251, scale=1 / math.sqrt(9 * C))
    bias = torch.randn(C, generator=torch.Generator().manual_seed(252))
    xcat = rnd(M, C1 + C2, seed=253)
    wsc = rnd(C, C1 + C2, 1, 1, seed=254, scale=1 / math.sqrt(C1 + C2))
    bsc = torch.randn(C, generator=torch.Generator().manual_seed(255))
    pw = ops.pack_conv3x3_shortcut(w.float(), bias, wsc.float(), bsc, dev)
    assert pw.K == 9 * C + C1 + C2 and pw.cin == C
    ref = F.conv2d(h.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(M, C) + xcat.float() @ wsc.float().reshape(C, -1).t() + bsc
    hh = h.permute(0, 2, 3, 1).contiguous().to(dev)
    # the sources as column windows of wider buffers (row stride != width), like the skip tensors of the UNet
    wide = torch.zeros(M, C1 + C2 + 64, dtype=BF16)
    wide[:, :C1 + C2] = xcat
    wide = wide.to(dev)
    a2 = wide[:, :C1]
    a3 = wide[:, C1:C1 + C2] if C2 else None
    cv = dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W)
    if case == "tiles":
        tiles = [(2, 1), (4, 1)] if backend.is_emu else [(22, 1), (21, 1), (11, 1), (4, 1), (1, 1), (17, 1)]
    elif case == "split_k":
        tiles = [(2, 3)] if backend.is_emu else [(21, 4), (4, 8), (23, 12)]
    else:
        tiles = [(0, 1)]
    for tile, sk in tiles:
        if tile and pw.Npad % ops.TILE_SHAPES[tile][1]:
            continue
        out = torch.empty(M, C, dtype=BF16, device=dev)
        ops.gemm(hh, pw, out, conv=cv, a2=a2, a3=a3, tile=tile, split_k=sk)
        backend.sync()
        close(out, ref)
    # refusals (pcdm.h: -1): extra K without its source, a source on a plain convolution, stride 2
    with pytest.raises((RuntimeError, AssertionError)):
        ops.gemm(hh, pw, torch.empty(M, C, dtype=BF16, device=dev), conv=cv)
    with pytest.raises((RuntimeError, AssertionError)):
        ops.gemm(hh, ops.pack_conv3x3(w.float(), bias, dev), torch.empty(M, C, dtype=BF16, device=dev), conv=cv, a2=a2)


@pytest.mark.parametrize("case", ["default", "tiles"])
def test_upsample_conv_phase_decomposition(backend, case):
    """Upsample2D's ``conv3x3(nearest x2 (x))`` as one 3x3 launch on the LOW-RES tensor with N = 4 C -- every output-channel group (= output phase
    (a, b)) contracts over its four taps only (pcdm_gemm_params.tap_lut; ops.pack_upsample_phases sums the taps that land on the same low-res
    pixel) -- followed by ``pixel_shuffle2``: against F.conv2d on the F.interpolate'd tensor in fp32 and against the gather form of the same
    library (tolerance: bf16 rounding of the summed weights).  Borders included (the zero padding of the upsampled tensor is the low-res
    tensor's).  ``tiles``: every tile family a tuner may pick; a tile whose N width does not divide the group is refused (-1)."""
    dev = backend.device
    B, H, W, C = (2, 5, 3, 64) if backend.is_emu else (8, 32, 44, 640)
    x = rnd(B, C, H, W, seed=260)
    w = rnd(C, C, 3, 3, seed=261, scale=1 / math.sqrt(9 * C))
    bias = torch.randn(C, generator=torch.Generator().manual_seed(262))
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, C)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    pw4 = ops.pack_upsample_phases(w.float(), bias, dev)
    assert pw4.N == 4 * C and pw4.K == 4 * C and pw4.cin == C
    tiles = [0] if case == "default" else ([2, 8] if backend.is_emu else [22, 21, 11, 4, 1, 18, 10])
    for tile in tiles:
        ph = torch.empty(B * H * W, 4 * C, dtype=BF16, device=dev)
        ops.gemm(xh, pw4, ph, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), tap_lut=ops.UPSAMPLE_TAP_LUT, tap_group_n=C, tile=tile)
        out = ops.pixel_shuffle2(ph, torch.empty(B * 4 * H * W, C, dtype=BF16, device=dev), B, H, W, C)
        backend.sync()
        close(out, ref)
    # the gather form of the same convolution (upsample folded into the implicit GEMM: nine taps on the upsampled grid)
    g = torch.empty(B * 4 * H * W, C, dtype=BF16, device=dev)
    ops.gemm(xh, ops.pack_conv3x3(w.float(), bias, dev), g, conv=dict(B=B, Hi=H, Wi=W, Ho=2 * H, Wo=2 * W, upsample=1))
    backend.sync()
    assert ((out.float().cpu() - g.float().cpu()).norm() / g.float().cpu().norm()).item() < 6e-3
    if not backend.is_emu:   # a 256-wide N tile does not divide a 640-channel group
        with pytest.raises(RuntimeError):
            ops.gemm(xh, pw4, ph, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), tap_lut=ops.UPSAMPLE_TAP_LUT, tap_group_n=C, tile=17)
        # the library's own heuristic (tile 0, no tuner: a host without Python) must pick a tile that fits the group: 64-channel groups at a
        # size where it would otherwise take a 128-wide tile
        Cn = 64
        xn = rnd(B, Cn, H, W, seed=263)
        wn = rnd(Cn, Cn, 3, 3, seed=264, scale=1 / math.sqrt(9 * Cn))
        pwn = ops.pack_upsample_phases(wn.float(), None, dev)
        phn = torch.empty(B * H * W, 4 * Cn, dtype=BF16, device=dev)
        auto = ops.AUTOTUNE
        ops.AUTOTUNE = False
        try:
            ops.gemm(xn.permute(0, 2, 3, 1).contiguous().to(dev), pwn, phn, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), tap_lut=ops.UPSAMPLE_TAP_LUT,
                     tap_group_n=Cn)
        finally:
            ops.AUTOTUNE = auto
        outn = ops.pixel_shuffle2(phn, torch.empty(B * 4 * H * W, Cn, dtype=BF16, device=dev), B, H, W, Cn)
        backend.sync()
        close(outn, F.conv2d(F.interpolate(xn.float(), scale_factor=2.0, mode="nearest"), wn.float(), None, padding=1).permute(0, 2, 3, 1).reshape(-1, Cn))


def test_conv_in_padded_channels(backend):
    """conv_in: 9 input channels zero-padded to 64 (weights too) == the 9-channel conv."""
    dev = backend.device
    B, H, W, Cout = (1, 4, 6, 64) if backend.is_emu else (8, 64, 88, 320)
    x = rnd(B, 9, H, W, seed=60)
    w = rnd(Cout, 9, 3, 3, seed=61, scale=1 / 9)
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(62))
    pw = ops.pack_conv3x3(w.float(), bias, dev)
    assert pw.cin == 64
    xh = torch.zeros(B, H, W, 64, dtype=BF16)
    xh[..., :9] = x.permute(0, 2, 3, 1)
    out = torch.empty(B * H * W, Cout, dtype=BF16, device=dev)
    ops.gemm(xh.to(dev), pw, out, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W))
    backend.sync()
    close(out.view(B, H, W, Cout), F.conv2d(x.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1))


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, B, H, Lq, Lk):
    qh = q.float().view(B, Lq, H, 64).transpose(1, 2)
    kh = k.float().view(B, Lk, H, 64).transpose(1, 2)
    vh = v.float().view(B, Lk, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / 8.0
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * Lq, H * 64)


@pytest.mark.parametrize("case", ["self", "cross258", "tiny88", "spike", "grid18"])
def test_flash_attn(backend, case):
    """("grid18": 3 query blocks x 2 heads x 3 batch entries = 18 workgroups, not a multiple of 8 -- the XCD-aware placement of attn.hip
    (attn_block_coords) is then a non-trivial bijection with a ragged last eighth; every (block, head, batch) must still be computed once.)"""
    dev = backend.device
    if backend.is_emu:
        B, H, Lq, Lk = {"self": (1, 2, 70, 70), "cross258": (1, 1, 40, 66), "tiny88": (2, 1, 24, 24),
                        "spike": (1, 1, 33, 130), "grid18": (3, 2, 260, 24)}[case]
    else:
        B, H, Lq, Lk = {"self": (8, 5, 5632, 5632), "cross258": (8, 10, 1408, 258), "tiny88": (8, 20, 88, 88),
                        "spike": (2, 5, 1408, 1408), "grid18": (3, 2, 260, 200)}[case]
    Cc = H * 64
    q, k, v = rnd(B * Lq, Cc, seed=70), rnd(B * Lk, Cc, seed=71), rnd(B * Lk, Cc, seed=72)
    if case == "spike":  # force large running-max jumps late in the key sequence (online-softmax rescale)
        k = k.clone()
        k[Lk - 3] = (q[5].float() * 6).to(BF16)
        k[Lk // 2] = (q[7].float() * 3).to(BF16)
    Lp = (Lk + 7) // 8 * 8
    vt = torch.zeros(B, Cc, Lp, dtype=BF16)
    vt[:, :, :Lk] = v.view(B, Lk, Cc).permute(0, 2, 1)
    # q / k as strided views of a fused buffer, like the UNet does
    qk = torch.zeros(B * max(Lq, Lk), 2 * Cc, dtype=BF16)
    qk[: B * Lq, :Cc] = q
    qk[: B * Lk, Cc:] = k
    qk = qk.to(dev)
    out = torch.empty(B * Lq, Cc, dtype=BF16, device=dev)
    ops.flash_attn(qk[: B * Lq, :Cc], qk[: B * Lk, Cc:], vt.to(dev), out, B, H, Lq, Lk)
    backend.sync()
    ref = _attn_ref(q, k, v, B, H, Lq, Lk)
    close(out, ref, tol=1.5e-2)
    # the lazy-rescale threshold changes WHEN the softmax reference moves, never the result: eager (0), the default and a huge
    # threshold (the reference is then set by the first key tile only) agree to bf16 rounding (cdna_hip_programming.md rule 26)
    for thr in (0.0, 16.0):
        o2 = torch.empty_like(out)
        ops.flash_attn(qk[: B * Lq, :Cc], qk[: B * Lk, Cc:], vt.to(dev), o2, B, H, Lq, Lk, thr=thr)
        backend.sync()
        close(o2, ref, tol=1.5e-2)


def test_flash_attn_strided_output(backend):
    """The output epilogue writes whole rows of `out` through LDS (16-byte stores): with `out` a column window of a wider buffer and a ragged
    query count, the bytes either side of the window and the rows of the next batch stay untouched, the window is bit-identical to the
    contiguous call; a row stride that is not a multiple of 8 elements is refused (pcdm.h: -1)."""
    dev = backend.device
    B, H, Lq, Lk = (2, 1, 37, 66) if backend.is_emu else (2, 5, 1003, 258)
    Cc = H * 64
    q, k, v = rnd(B * Lq, Cc, seed=170), rnd(B * Lk, Cc, seed=171), rnd(B * Lk, Cc, seed=172)
    Lp = (Lk + 7) // 8 * 8
    vt = torch.zeros(B, Cc, Lp, dtype=BF16)
    vt[:, :, :Lk] = v.view(B, Lk, Cc).permute(0, 2, 1)
    qd, kd, vtd = q.to(dev), k.to(dev), vt.to(dev)
    ref = torch.empty(B * Lq, Cc, dtype=BF16, device=dev)
    ops.flash_attn(qd, kd, vtd, ref, B, H, Lq, Lk)
    wide = torch.full((B * Lq, Cc + 16), 7.0, dtype=BF16, device=dev)
    ops.flash_attn(qd, kd, vtd, wide[:, 8:8 + Cc], B, H, Lq, Lk)
    backend.sync()
    w = wide.cpu()
    assert torch.equal(w[:, 8:8 + Cc], ref.cpu())
    assert (w[:, :8] == 7.0).all() and (w[:, 8 + Cc:] == 7.0).all()
    close(ref, _attn_ref(q, k, v, B, H, Lq, Lk), tol=1.5e-2)
    odd = torch.empty(B * Lq, Cc + 4, dtype=BF16, device=dev)
    with pytest.raises(RuntimeError):
        ops.flash_attn(qd, kd, vtd, odd[:, :Cc], B, H, Lq, Lk)


def _e4m3(x: torch.Tensor) -> torch.Tensor:
    """OCP e4m3fn round trip (RNE, saturating) -- torch's own float8_e4m3fn cast, used only as the test's quantiser."""
    return x.float().clamp(-448, 448).to(torch.float8_e4m3fn).float()


@pytest.mark.parametrize("case", ["self", "cross258", "tiny88", "spike"])
def test_flash_attn_fp8(backend, case):
    """N4: e4m3 K / V^T / Q / P on the MX-scaled fp8 MFMA.  Two checks:
    * against the fp32 softmax attention of the SAME quantised operands (q*c, K, V rounded to e4m3 as the kernel does): what is left
      is the e4m3 rounding of P and fp32 summation order -- rel-L2 <= 4e-2;
    * against the un-quantised fp32 attention: the stated fp8 tolerance of this path, rel-L2 <= 8e-2 (bf16 kernel: 1.5e-2 max-abs)."""
    dev = backend.device
    if backend.is_emu:
        B, H, Lq, Lk = {"self": (1, 2, 70, 70), "cross258": (1, 1, 40, 66), "tiny88": (2, 1, 24, 24), "spike": (1, 1, 33, 130)}[case]
    else:
        B, H, Lq, Lk = {"self": (8, 5, 5632, 5632), "cross258": (4, 10, 1408, 258), "tiny88": (8, 20, 88, 88), "spike": (2, 5, 1408, 1408)}[case]
    Cc = H * 64
    q, k, v = rnd(B * Lq, Cc, seed=70), rnd(B * Lk, Cc, seed=71), rnd(B * Lk, Cc, seed=72)
    if case == "spike":
        k = k.clone()
        k[Lk - 3] = (q[5].float() * 6).to(BF16)
        k[Lk // 2] = (q[7].float() * 3).to(BF16)
    # quantised operands through the library's own kernel; checked against torch's e4m3 cast
    Lp = (Lk + 15) // 16 * 16
    k8 = torch.empty(B * Lk, Cc, dtype=torch.uint8, device=dev)
    ops.quantize_fp8(k.to(dev), k8)
    vt = v.view(B, Lk, Cc).permute(0, 2, 1).contiguous().view(B * Cc, Lk)
    vt8 = torch.full((B * Cc, Lp), 0x7f, dtype=torch.uint8, device=dev)
    ops.quantize_fp8(vt.to(dev), vt8, cols=Lk)
    backend.sync()
    assert torch.equal(k8.cpu().view(torch.float8_e4m3fn).float(), _e4m3(k))
    assert torch.equal(vt8.cpu().view(torch.float8_e4m3fn).float()[:, :Lk], _e4m3(vt)) and (vt8.cpu()[:, Lk:] == 0).all()
    out = torch.empty(B * Lq, Cc, dtype=BF16, device=dev)
    ops.flash_attn_fp8(q.to(dev), k8, vt8.view(B, Cc, Lp), out, B, H, Lq, Lk)
    backend.sync()
    c = 0.125 * 1.44269504088896341
    qq = (_e4m3(q.float() * c) / c).to(torch.float32)
    ref_q = _attn_ref(qq, _e4m3(k), _e4m3(v), B, H, Lq, Lk)
    ref = _attn_ref(q, k, v, B, H, Lq, Lk)
    o = out.float().cpu()
    r1 = ((o - ref_q).norm() / ref_q.norm()).item()
    r2 = ((o - ref).norm() / ref.norm()).item()
    assert torch.isfinite(o).all() and r1 <= 4e-2 and r2 <= 8e-2, (r1, r2)


def test_flash_attn_fp8_hot_queries(backend):
    """Range guard of the fp8 attention (VERDICT r3 #3e): queries whose scaled rows ``q * scale * log2e`` exceed e4m3's 448 (to_q gains of a
    trained checkpoint, outlier channels) used to be CLAMPED, silently changing their scores.  They are now divided by a per-query power
    of two before the conversion and their scores multiplied back in fp32 -- exact.  Every third query carries one huge component
    (+-6000, ~2.4x the clamp point) against a key column of +-2^-9 (exact in e4m3), i.e. +-1.46 of its scores ride on that component (+-0.61 when clamped); a
    second outlier group sits just BELOW the limit (no rescale).  Checked against fp32 attention on the same quantised operands."""
    dev = backend.device
    B, H, Lq, Lk = (1, 1, 48, 70) if backend.is_emu else (2, 5, 1408, 1408)
    Cc = H * 64
    q, k, v = rnd(B * Lq, Cc, seed=170).float(), rnd(B * Lk, Cc, seed=171).float(), rnd(B * Lk, Cc, seed=172)
    sgn = (torch.rand(B * Lk, generator=torch.Generator().manual_seed(173)) < 0.5).float() * 2 - 1
    for hd in range(H):
        k[:, hd * 64 + 7] = sgn * 2.0 ** -9
        k[:, hd * 64 + 9] = -sgn * 2.0 ** -9
        q[0::3, hd * 64 + 7] = 6000.0      # rescaled rows (6000 * 0.18 = 1082 > 448)
        q[1::3, hd * 64 + 9] = -2400.0     # just inside (2400 * 0.18 = 433): must stay on the plain path
    q, k = q.to(BF16), k.to(BF16)
    Lp = (Lk + 15) // 16 * 16
    k8 = torch.empty(B * Lk, Cc, dtype=torch.uint8, device=dev)
    ops.quantize_fp8(k.to(dev), k8)
    vt = v.view(B, Lk, Cc).permute(0, 2, 1).contiguous().view(B * Cc, Lk)
    vt8 = torch.zeros(B * Cc, Lp, dtype=torch.uint8, device=dev)
    ops.quantize_fp8(vt.to(dev), vt8, cols=Lk)
    out = torch.empty(B * Lq, Cc, dtype=BF16, device=dev)
    ops.flash_attn_fp8(q.to(dev), k8, vt8.view(B, Cc, Lp), out, B, H, Lq, Lk)
    backend.sync()
    # reference on the quantised operands: a row's own power of two commutes with the e4m3 rounding (barring underflow of its small elements)
    c = 0.125 * 1.44269504088896341
    qs = q.float() * c
    amax = qs.view(B * Lq, H, 64).abs().amax(-1, keepdim=True)
    e = torch.where(amax > 448, torch.ceil(torch.log2(amax / 448)), torch.zeros_like(amax))
    qq = (_e4m3((qs.view(B * Lq, H, 64) * 2.0 ** -e)) * 2.0 ** e).view(B * Lq, Cc) / c
    ref_q = _attn_ref(qq, _e4m3(k), _e4m3(v), B, H, Lq, Lk)
    o = out.float().cpu()
    for rows in (slice(0, None, 3), slice(1, None, 3), slice(2, None, 3)):
        r = ((o[rows] - ref_q[rows]).norm() / ref_q[rows].norm()).item()
        assert torch.isfinite(o).all() and r <= 4e-2, (rows, r)
    # what clamping would have given is measurably different -- the test would have caught the old behaviour
    qc = (_e4m3((q.float() * c).clamp(-448, 448)) / c)
    ref_clamped = _attn_ref(qc, _e4m3(k), _e4m3(v), B, H, Lq, Lk)
    assert ((ref_clamped[0::3] - ref_q[0::3]).norm() / ref_q[0::3].norm()).item() > 0.2


# ------------------------------------------------------------------------------------------------ small ops
def test_timestep_embedding_and_small_linear(backend):
    dev = backend.device
    from oracle.unet import timestep_embedding
    B, dim = 4, 64 if backend.is_emu else 320
    ts = torch.tensor([981, 961, 1, 500], dtype=torch.int64, device=dev)
    for idx in (0, 2, 3):
        step = torch.tensor([idx], dtype=torch.int32, device=dev)
        out = torch.empty(B, dim, dtype=torch.float32, device=dev)
        ops.timestep_embedding(ts, step, out)
        backend.sync()
        ref = timestep_embedding(ts[idx].cpu().expand(B), dim)
        assert (out.cpu() - ref).abs().max() < 2e-4
    K, N = (64, 40) if backend.is_emu else (1280, 20160)
    x = torch.randn(B, K, generator=torch.Generator().manual_seed(80))
    w = rnd(N, K, seed=81, scale=1 / math.sqrt(K))
    bias = torch.randn(N, generator=torch.Generator().manual_seed(82))
    add = torch.randn(B, N, generator=torch.Generator().manual_seed(83))
    for act_in, act_out in ((False, False), (True, False), (False, True)):
        out = torch.empty(B, N, dtype=torch.float32, device=dev)
        ops.small_linear(x.to(dev), w.to(dev), bias.to(dev), out, add=add.to(dev), act_in=act_in, act_out=act_out)
        backend.sync()
        xi = F.silu(x) if act_in else x
        ref = xi @ w.float().t() + bias
        ref = (F.silu(ref) if act_out else ref) + add
        assert (out.cpu() - ref).abs().max() < 2e-3


def test_assemble_cfg_step_lincomb_layout(backend):
    dev = backend.device
    N, h, w = (2, 4, 6) if backend.is_emu else (4, 64, 88)
    g = torch.Generator().manual_seed(90)
    lat = torch.randn(N, 4, h, w, generator=g)
    mask = torch.cat([torch.ones(1, 1, h, w // 2), torch.zeros(1, 1, h, w // 2)], 3)
    masked = torch.randn(1, 4, h, w, generator=g)
    out = torch.empty(2 * N, h, w, 64, dtype=BF16, device=dev)
    ops.assemble_input(lat.to(dev), 2, mask.to(dev), masked.to(dev), out)
    backend.sync()
    ref = torch.cat([torch.cat([lat] * 2), mask.expand(2 * N, -1, -1, -1), masked.expand(2 * N, -1, -1, -1)], 1)
    o = out.float().cpu()
    assert torch.equal(o[..., :9], ref.to(BF16).float().permute(0, 2, 3, 1))
    assert (o[..., 9:] == 0).all()
    # layout conversions round-trip
    x = torch.randn(2, 16, h, w, generator=g)
    xh = ops.nchw_to_nhwc_bf16(x.to(dev))
    back = ops.nhwc_bf16_to_nchw(xh, 2, 16, h, w)
    backend.sync()
    assert torch.equal(back.cpu(), x.to(BF16).float())
    assert torch.equal(ops.f32_to_bf16(x.to(dev)).cpu(), x.to(BF16))
    # CFG + step with device-side coefficient table
    eps = torch.randn(2 * N, 4, h, w, generator=g)
    coef = torch.tensor([[9., 9., 9., 0.], [1.25, -0.5, 0.3, 0.]], dtype=torch.float32)
    step = torch.tensor([1], dtype=torch.int32, device=dev)
    noise = torch.randn(N, 4, h, w, generator=g)
    xp = torch.empty(N, 4, h, w, dtype=torch.float32, device=dev)
    eo = torch.empty_like(xp)
    ops.cfg_step(eps.to(dev), True, 2.0, lat.to(dev), xp, coef.to(dev), step, noise=noise.to(dev), eps_out=eo)
    ops.advance_step(step)
    backend.sync()
    assert int(step.item()) == 2
    u, c = eps.chunk(2)
    e = u + 2.0 * (c - u)
    assert torch.allclose(eo.cpu(), e, atol=1e-6)
    assert torch.allclose(xp.cpu(), 1.25 * lat - 0.5 * e + 0.3 * noise, atol=1e-5)
    y = torch.empty_like(xp)
    ops.lincomb(y, [xp, eo, noise.to(dev)], [0.5, -2.0, 3.0])
    backend.sync()
    assert torch.allclose(y.cpu(), 0.5 * xp.cpu() - 2 * eo.cpu() + 3 * noise, atol=1e-5)


def test_gemm_full_row_tiles_and_zero_rows(backend):
    """Tile 21 (192x320 block, 16x16x32 fragments, wave tile 96x80: LDS-staged epilogue in passes of 64 + 16 channels) and 26 (192x256): linear with bias / residual / row vector, the qkv split epilogue (V^T), a conv with M tail, split-K; and
    ``zero_rows`` (A rows declared all-zero: not read, tiles entirely inside run the epilogue only) on old and new tiles."""
    dev = backend.device
    tiles = (21,)
    M, K, N = (300, 192, 320) if backend.is_emu else (5632 * 2 + 100, 640, 640)
    a = rnd(M, K, seed=60)
    w = rnd(N, K, seed=61, scale=1 / math.sqrt(K))
    bias = torch.randn(N, generator=torch.Generator().manual_seed(62))
    res = rnd(M, N, seed=63)
    rpb = M // 2
    rowvec = torch.randn(2, N, generator=torch.Generator().manual_seed(64))
    pw = ops.pack_linear(w.float(), bias, dev)
    ref = a.float() @ w.float().t() + bias + res.float() + rowvec.repeat_interleave(rpb, 0)
    for tile in tiles:
        out = torch.empty(M, N, dtype=BF16, device=dev)
        ops.gemm(a.to(dev), pw, out, rowvec=rowvec.to(dev), rows_per_batch=rpb, residual=res.to(dev), res_mod=M, tile=tile)
        backend.sync()
        close(out, ref)
    # zero_rows: rows [0, z) of A hold GARBAGE that must not be read; out = bias + residual there
    for tile, z in ((21, 192), (21, 100), (3, 256), (4, 130)) if backend.is_emu else ((21, M // 2), (21, 777), (13, M // 2), (18, 4321), (0, M // 2)):
        if pw.Npad % ops.TILE_SHAPES.get(tile, (0, 64))[1]:
            continue
        ag = a.clone()
        ag[:z] = float("nan")
        out = torch.empty(M, N, dtype=BF16, device=dev)
        ops.gemm(ag.to(dev), pw, out, residual=res.to(dev), res_mod=M, tile=tile, zero_rows=z)
        backend.sync()
        az = a.float().clone()
        az[:z] = 0
        close(out, az @ w.float().t() + bias + res.float())
    # split-K on a full-row tile
    if not backend.is_emu:
        a2, w2 = rnd(704, 11520, seed=65), rnd(1280, 11520, seed=66, scale=1 / math.sqrt(11520))
        pw2 = ops.pack_linear(w2.float(), None, dev)
        out = torch.empty(704, 1280, dtype=BF16, device=dev)
        for tile, sk in ((21, 4), (21, 7), (26, 3)):
            ops.gemm(a2.to(dev), pw2, out, tile=tile, split_k=sk)
            backend.sync()
            close(out, a2.float() @ w2.float().t())
    # fused q|k|v projection with the V^T epilogue (N = 3C = 960 at level 0) on a full-row tile
    Bq, T, C = (2, 20, 320) if backend.is_emu else (8, 5632, 320)
    x = rnd(Bq * T, C if not backend.is_emu else 64, seed=67)
    wq = rnd(3 * C, x.shape[1], seed=68, scale=1 / math.sqrt(x.shape[1]))
    pwq = ops.pack_linear(wq.float(), None, dev)
    qk = torch.empty(Bq * T, 2 * C, dtype=BF16, device=dev)
    Tp = (T + 7) // 8 * 8
    vt = torch.zeros(Bq, C, Tp, dtype=BF16, device=dev)
    pr = x.float() @ wq.float().t()
    for tile in (21,):
        qk.zero_(); vt.zero_()
        ops.gemm(x.to(dev), pwq, qk, rows_per_batch=T, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * C, tile=tile)
        backend.sync()
        close(qk, pr[:, : 2 * C])
        close(vt[:, :, :T], pr[:, 2 * C:].view(Bq, T, C).permute(0, 2, 1))
    # conv (halo, M tail) on the full-row tiles
    B, H, W, Cin, Cout = (2, 6, 5, 64, 320) if backend.is_emu else (3, 30, 44, 640, 320)
    xc = rnd(B, Cin, H, W, seed=70)
    wc = rnd(Cout, Cin, 3, 3, seed=71, scale=1 / math.sqrt(9 * Cin))
    bc = torch.randn(Cout, generator=torch.Generator().manual_seed(72))
    refc = F.conv2d(xc.float(), wc.float(), bc, padding=1).permute(0, 2, 3, 1)
    pwc = ops.pack_conv3x3(wc.float(), bc, dev)
    xh = xc.permute(0, 2, 3, 1).contiguous().to(dev)
    for tile in tiles:
        out = torch.empty(B * H * W, Cout, dtype=BF16, device=dev)
        ops.gemm(xh, pwc, out, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), tile=tile)
        backend.sync()
        close(out.view(B, H, W, Cout), refc)


def test_rowvec_step_counter_is_bounded_on_the_device(backend):
    """ABI 4: the device step counter that selects the per-step block of a row-vector table (the time-embedding projections of every
    denoise step) is BOUNDED inside the kernels: a value beyond the table is clamped to its last block (negative: to the first) and a device
    flag raised -- the launch never reads foreign memory.  The plain epilogue, the split-K reduce kernel and the GroupNorm that consumes
    deferred split-K slabs all take the counter; an in-range counter leaves the flag alone."""
    dev = backend.device
    M, K, N, nblk = (96, 128, 64, 3) if backend.is_emu else (2816, 2560, 1280, 5)
    B = 2
    rpb = M // B
    a = rnd(M, K, seed=500)
    pw = ops.pack_linear(rnd(N, K, seed=501, scale=1 / math.sqrt(K)).float(), None, dev)
    table = torch.randn(nblk, B, N, generator=torch.Generator().manual_seed(502))
    tab_dev = torch.cat([table.reshape(-1), torch.full((4 * B * N,), float("nan"))]).to(dev)   # NaN behind the table: an unclamped read would show
    base = a.float() @ pw.w[:N].float().cpu().t()
    for step_v, want_blk, want_flag in ((1, 1, 0), (nblk - 1, nblk - 1, 0), (nblk, nblk - 1, 1), (nblk + 3, nblk - 1, 1), (-2, 0, 1)):
        for sk in (1, 2):
            step = torch.tensor([step_v], dtype=torch.int32, device=dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            out = torch.empty(M, N, dtype=BF16, device=dev)
            ops.gemm(a.to(dev), pw, out, rowvec=tab_dev[: B * N].view(B, N), rows_per_batch=rpb, rowvec_step=step, rowvec_step_stride=B * N,
                     rowvec_step_count=nblk, step_error=err, tile=2, split_k=sk)
            backend.sync()
            close(out, base + table[want_blk].repeat_interleave(rpb, 0))
            assert int(err.item()) == want_flag, (step_v, sk)
    # the deferred split-K reduce inside the GroupNorm
    G = 8
    gamma, beta = torch.rand(N, generator=torch.Generator().manual_seed(503)) + 0.5, torch.zeros(N)
    ws = ops.groupnorm_ws(B, N, dev)
    for step_v, want_blk, want_flag in ((0, 0, 0), (nblk + 1, nblk - 1, 1)):
        step = torch.tensor([step_v], dtype=torch.int32, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        pre = torch.empty(M, N, dtype=BF16, device=dev)
        d = ops.gemm(a.to(dev), pw, pre, rowvec=tab_dev[: B * N].view(B, N), rows_per_batch=rpb, rowvec_step=step, rowvec_step_stride=B * N,
                     rowvec_step_count=nblk, step_error=err, tile=2, split_k=2, defer_reduce=True)
        assert isinstance(d, ops.DeferredGemm)
        y = torch.empty(M, N, dtype=BF16, device=dev)
        ops.groupnorm(d, None, B, rpb, G, 1e-5, gamma.to(dev), beta.to(dev), False, y, ws)
        backend.sync()
        x = (base + table[want_blk].repeat_interleave(rpb, 0)).to(BF16).float()
        ref = F.group_norm(x.view(B, rpb, N).permute(0, 2, 1), G, gamma, beta, 1e-5).permute(0, 2, 1).reshape(M, N)
        close(y, ref, tol=2e-2)
        close(pre, x)
        assert int(err.item()) == want_flag, step_v


def test_gemm_uneven_176_row_tiles(backend):
    """Round 6: the 176-row block tiles (22 = 176 x 320, 23 = 176 x 256; wave rows of 6 + 5 fragment rows, gemm_kernel.inc UNEVEN) -- 45056 =
    256 x 176, so the level-0 launches fill the 256 CUs exactly.  Same fragments, same K order as tiles 21 / 26: the outputs must be
    BIT-IDENTICAL to theirs (and within tolerance of fp32).  Linear with bias / row vector / residual over several M tiles and an M tail, the
    two-source concat, ``zero_rows``, split-K, GEGLU (tile 23), the 3x3 convolution with halo / stride 2 / nearest-x2 / ``dup_rows``.  The
    rows a tile must NOT write (the five-fragment wave row's sixth fragment = the next tile's first 16 rows) are checked with an in-place
    residual: a second write of a row would add its bias twice."""
    dev = backend.device
    M, K, N = (176 * 2 + 100, 128, 320) if backend.is_emu else (176 * 70 + 100, 640, 640)
    a = rnd(M, K, seed=160)
    w = rnd(N, K, seed=161, scale=1 / math.sqrt(K))
    bias = torch.randn(N, generator=torch.Generator().manual_seed(162))
    res = rnd(M, N, seed=163)
    rpb = M // 2
    rowvec = torch.randn(2, N, generator=torch.Generator().manual_seed(164))
    pw = ops.pack_linear(w.float(), bias, dev)
    ref = a.float() @ w.float().t() + bias + res.float() + rowvec.repeat_interleave(rpb, 0)
    outs = {}
    for tile in (21, 22):
        out = torch.full((M + 40, N), 7.0, dtype=BF16, device=dev)     # 40 guard rows behind the tensor
        ops.gemm(a.to(dev), pw, out[:M], rowvec=rowvec.to(dev), rows_per_batch=rpb, residual=res.to(dev), res_mod=M, tile=tile)
        backend.sync()
        close(out[:M], ref)
        assert (out[M:] == 7.0).all(), tile
        outs[tile] = out[:M].clone()
    assert torch.equal(outs[21], outs[22])
    # in place (out is the residual): every row written exactly once
    for tile in (22, 23):
        if pw.Npad % ops.TILE_SHAPES[tile][1]:
            continue
        io = res.clone().to(dev)
        ops.gemm(a.to(dev), pw, io, residual=io, res_mod=M, tile=tile)
        backend.sync()
        close(io, a.float() @ w.float().t() + bias + res.float())
    # two-source concat + zero_rows (tiles entirely inside skip their K loop)
    K1 = K // 2
    z = 176 + 50
    ag = a.clone()
    ag[:z] = float("nan")
    az = a.float().clone()
    az[:z] = 0
    got = {}
    for tile in (21, 22):
        out = torch.empty(M, N, dtype=BF16, device=dev)
        ops.gemm(ag[:, :K1].contiguous().to(dev), pw, out, a2=ag[:, K1:].contiguous().to(dev), residual=res.to(dev), res_mod=M, tile=tile, zero_rows=z)
        backend.sync()
        close(out, az @ w.float().t() + bias + res.float())
        got[tile] = out.clone()
    assert torch.equal(got[21], got[22])
    # split-K (raw fp32 slabs + reduce): the ghost fragment rows must not reach the slabs of the next tile either
    Ms, Ks, Ns = (176 + 60, 512, 320) if backend.is_emu else (704, 11520, 1280)
    a2, w2 = rnd(Ms, Ks, seed=165), rnd(Ns, Ks, seed=166, scale=1 / math.sqrt(Ks))
    pw2 = ops.pack_linear(w2.float(), None, dev)
    got = {}
    for tile, sk in ((21, 4), (22, 4), (22, 3), (23, 2)):
        if pw2.Npad % ops.TILE_SHAPES[tile][1]:
            continue
        out = torch.empty(Ms, Ns, dtype=BF16, device=dev)
        ops.gemm(a2.to(dev), pw2, out, tile=tile, split_k=sk)
        backend.sync()
        close(out, a2.float() @ w2.float().t())
        got[(tile, sk)] = out.clone()
    assert torch.equal(got[(21, 4)], got[(22, 4)])
    # GEGLU on the 64-wide wave tiles: 23 against 26
    Mg, Kg, Ng = (176 + 30, 64, 128) if backend.is_emu else (176 * 33 + 8, 320, 1280)
    ag_ = rnd(Mg, Kg, seed=167)
    wg = rnd(2 * Ng, Kg, seed=168, scale=1 / math.sqrt(Kg))
    bg = torch.randn(2 * Ng, generator=torch.Generator().manual_seed(169))
    pg = ops.pack_geglu(wg.float(), bg, dev)
    pr = ag_.float() @ wg.float().t() + bg
    refg = pr[:, :Ng] * F.gelu(pr[:, Ng:])
    got = {}
    for tile in (26, 23):
        if pg.Npad % ops.TILE_SHAPES[tile][1]:
            continue
        out = torch.empty(Mg, Ng, dtype=BF16, device=dev)
        ops.gemm(ag_.to(dev), pg, out, epilogue=ops.EPI_GEGLU, tile=tile)
        backend.sync()
        close(out, refg)
        got[tile] = out.clone()
    if len(got) == 2:
        assert torch.equal(got[26], got[23])
    # the q | k | v^T epilogue is refused (whole 32-token passes only), not mis-executed
    Tq = 32
    xq = rnd(2 * Tq, 64, seed=170)
    pwq = ops.pack_linear(rnd(3 * 320, 64, seed=171).float(), None, dev)
    with pytest.raises(RuntimeError):
        ops.gemm(xq.to(dev), pwq, torch.empty(2 * Tq, 640, dtype=BF16, device=dev), rows_per_batch=Tq, epilogue=ops.EPI_SPLIT_VT,
                 out2=torch.zeros(2, 320, Tq, dtype=BF16, device=dev), vt_col0=640, tile=22)
    # 3x3 convolutions: halo + M tail, stride 2, nearest-x2 upsample, dup_rows with per-half row vectors and residuals
    B, H, W, Cin, Cout = (2, 13, 9, 64, 320) if backend.is_emu else (3, 30, 44, 640, 320)
    xc = rnd(B, Cin, H, W, seed=172)
    wc = rnd(Cout, Cin, 3, 3, seed=173, scale=1 / math.sqrt(9 * Cin))
    bc = torch.randn(Cout, generator=torch.Generator().manual_seed(174))
    pwc = ops.pack_conv3x3(wc.float(), bc, dev)
    xh = xc.permute(0, 2, 3, 1).contiguous().to(dev)
    for name, kw, refc in (
            ("s1", dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), F.conv2d(xc.float(), wc.float(), bc, padding=1)),
            ("s2", dict(B=B, Hi=H, Wi=W, Ho=(H + 1) // 2, Wo=(W + 1) // 2, stride=2), F.conv2d(xc.float(), wc.float(), bc, padding=1, stride=2)),
            ("up", dict(B=B, Hi=H, Wi=W, Ho=2 * H, Wo=2 * W, upsample=1),
             F.conv2d(F.interpolate(xc.float(), scale_factor=2.0, mode="nearest"), wc.float(), bc, padding=1))):
        got = {}
        for tile in (21, 22):
            Mo = B * kw["Ho"] * kw["Wo"]
            out = torch.empty(Mo, Cout, dtype=BF16, device=dev)
            ops.gemm(xh, pwc, out, conv=kw, tile=tile)
            backend.sync()
            close(out.view(B, kw["Ho"], kw["Wo"], Cout), refc.permute(0, 2, 3, 1))
            got[tile] = out.clone()
        assert torch.equal(got[21], got[22]), name
    Mo = B * H * W
    rv2 = torch.randn(2 * B, Cout, generator=torch.Generator().manual_seed(175))
    res2 = rnd(2 * Mo, Cout, seed=176)
    base = F.conv2d(xc.float(), wc.float(), bc, padding=1).permute(0, 2, 3, 1).reshape(Mo, Cout)
    refd = torch.cat([base, base]) + rv2.repeat_interleave(H * W, 0) + res2.float()
    got = {}
    for tile in (21, 22):
        out = torch.empty(2 * Mo, Cout, dtype=BF16, device=dev)
        ops.gemm(xh, pwc, out, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), rowvec=rv2.to(dev), rows_per_batch=H * W, residual=res2.to(dev),
                 res_mod=2 * Mo, tile=tile, dup_rows=Mo)
        backend.sync()
        close(out, refd)
        got[tile] = out.clone()
    assert torch.equal(got[21], got[22])


def test_rowgemm_thin_k(backend):
    """rowgemm.hip (tiles 31..34; 34 = four waves, two workgroups per CU, N tiles split over gridDim.y): the K = 320 GEMM with the activation rows stationary in registers and the weights streamed through one
    LDS ring across all N tiles -- bias + residual store (with an M tail and ``zero_rows``), GEGLU, the q|k / V^T split epilogue, and
    the FOLDED LayerNorm (gamma / beta in the packed weights, row statistics taken in the kernel) against LayerNorm -> GEMM in fp32."""
    dev = backend.device
    K = 320
    g = torch.Generator().manual_seed(90)
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3

    def ln_ref(x):
        return F.layer_norm(x.float(), (K,), gamma, beta, 1e-5).to(BF16).float()   # the fused path rounds LN(x) to bf16, like pcdm_layernorm

    # ---- plain store: bias + residual, M tail, several N-tile counts
    for (M, N, tiles) in ([(200, 128, (31, 32, 33, 34, 35, 36)), (100, 320, (32, 35, 34))] if backend.is_emu else
                          [(45056, 320, (31, 32, 33, 35, 36, 34)), (22528 + 40, 320, (32, 33, 36, 34)), (45056, 960, (32, 35, 34)), (1000, 1280, (31, 32, 33, 36, 34))]):
        a = rnd(M, K, seed=91)
        w = rnd(N, K, seed=92, scale=1 / math.sqrt(K))
        bias = torch.randn(N, generator=torch.Generator().manual_seed(93))
        res = rnd(M, N, seed=94)
        pw = ops.pack_linear(w.float(), bias, dev)
        pw_ln = ops.pack_linear_ln(w.float(), bias, gamma, beta, dev)
        ref = a.float() @ w.float().t() + bias + res.float()
        ref_ln = ln_ref(a) @ w.float().t() + bias
        for tile in tiles:
            if pw.Npad % ops.TILE_SHAPES[tile][1]:
                continue
            out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
            ops.gemm(a.to(dev), pw, out, residual=res.to(dev), res_mod=M, tile=tile)
            backend.sync()
            close(out, ref)
            out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
            ops.gemm(a.to(dev), pw, out, tile=tile, ln=(gamma.to(dev), beta.to(dev), 1e-5), ln_buf=torch.empty(M, K, dtype=BF16, device=dev),
                     pw_ln=pw_ln)
            backend.sync()
            close(out, ref_ln)
        # zero_rows: garbage in the declared-zero rows must not be read (one value inside a workgroup's rows, one covering whole workgroups)
        for z in ((40, 192) if backend.is_emu else (777, 192 * 100)):
            ag = a.clone()
            ag[:z] = float("nan")
            az = a.float().clone()
            az[:z] = 0
            out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
            ops.gemm(ag.to(dev), pw, out, residual=res.to(dev), res_mod=M, tile=tiles[-1] if not pw.Npad % ops.TILE_SHAPES[tiles[-1]][1] else 32,
                     zero_rows=min(z, M))
            backend.sync()
            close(out, az @ w.float().t() + bias + res.float())
    # ---- GEGLU (+ LayerNorm): tiles whose waves own 64 columns
    M, D = (120, 128) if backend.is_emu else (45056, 1280)
    a = rnd(M, K, seed=95)
    w = rnd(2 * D, K, seed=96, scale=1 / math.sqrt(K))
    bias = torch.randn(2 * D, generator=torch.Generator().manual_seed(97)) * 0.5
    pw = ops.pack_geglu(w.float(), bias, dev)
    pw_ln = ops.pack_geglu_ln(w.float(), bias, gamma, beta, dev)
    for use_ln in (False, True):
        pr = (ln_ref(a) if use_ln else a.float()) @ w.float().t() + bias
        h, gt = pr.chunk(2, -1)
        for tile in (31, 35, 34):
            out = torch.full((M, D), float("nan"), dtype=BF16, device=dev)
            kw = dict(ln=(gamma.to(dev), beta.to(dev), 1e-5), ln_buf=torch.empty(M, K, dtype=BF16, device=dev), pw_ln=pw_ln) if use_ln else {}
            ops.gemm(a.to(dev), pw, out, epilogue=ops.EPI_GEGLU, tile=tile, **kw)
            backend.sync()
            close(out, h * F.gelu(gt))
    # ---- q | k | v^T (+ LayerNorm): N = 3C, tokens per batch entry a multiple of 16
    Bq, T, Cc = (2, 48, 64) if backend.is_emu else (8, 5632, 320)
    x = rnd(Bq * T, K, seed=98)
    wq = rnd(3 * Cc, K, seed=99, scale=1 / math.sqrt(K))
    pwq = ops.pack_linear(wq.float(), None, dev)
    pwq_ln = ops.pack_linear_ln(wq.float(), None, gamma, beta, dev)
    for use_ln in (False, True):
        pr = (ln_ref(x) if use_ln else x.float()) @ wq.float().t()
        for tile in (31, 32, 33, 34, 36):
            if pwq.Npad % ops.TILE_SHAPES[tile][1] or (2 * Cc) % (ops.TILE_SHAPES[tile][1] // {31: 2, 32: 2, 33: 4, 34: 1, 35: 1, 36: 1}[tile]):
                continue
            qk = torch.full((Bq * T, 2 * Cc), float("nan"), dtype=BF16, device=dev)
            vt = torch.zeros(Bq, Cc, T + 8, dtype=BF16, device=dev)
            kw = dict(ln=(gamma.to(dev), beta.to(dev), 1e-5), ln_buf=torch.empty(Bq * T, K, dtype=BF16, device=dev), pw_ln=pwq_ln) if use_ln else {}
            ops.gemm(x.to(dev), pwq, qk, rows_per_batch=T, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * Cc, tile=tile, **kw)
            backend.sync()
            close(qk, pr[:, : 2 * Cc])
            close(vt[:, :, :T], pr[:, 2 * Cc:].view(Bq, T, Cc).permute(0, 2, 1))
            assert (vt[:, :, T:] == 0).all(), tile
    # the kernel is K = 320 only and says so
    pw64 = ops.pack_linear(rnd(64, 64, seed=1).float(), None, dev)
    with pytest.raises(RuntimeError):
        ops.gemm(rnd(32, 64, seed=2).to(dev), pw64, torch.empty(32, 64, dtype=BF16, device=dev), tile=31)


def test_rowgemm_folded_layernorm_large_mean(backend):
    """The folded LayerNorm ``rstd (x W'^T - mean wsum) + b'`` (rowgemm.hip) on rows whose |mean| / std is 50 (the
    cancellation between the two terms grows with |mean| / std).  Both terms are exact bf16 products accumulated in fp32, so the
    difference keeps ~2^-24 x 50 relative accuracy; the test states it: against LayerNorm (fp64, on the same bf16 rows) -> fp64 GEMM,
    within the kernels' usual 1 % (what remains is the bf16 rounding of W' = W diag(gamma) and of the output)."""
    dev = backend.device
    K = 320
    g = torch.Generator().manual_seed(190)
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 3.0      # LayerNorm beta ~ N(0, 3^2)
    M, N = (100, 128) if backend.is_emu else (45056, 960)
    sign = torch.where(torch.rand(M, 1, generator=g) < 0.5, -1.0, 1.0)
    a = (torch.randn(M, K, generator=g) + 50.0 * sign).to(BF16)                           # every row: |mean| = 50 std
    w = rnd(N, K, seed=192, scale=1 / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    pw = ops.pack_linear(w.float(), bias, dev)
    pw_ln = ops.pack_linear_ln(w.float(), bias, gamma, beta, dev)
    ref = F.layer_norm(a.double(), (K,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + bias.double()
    for tile in (34, 32):
        out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
        ops.gemm(a.to(dev), pw, out, tile=tile, ln=(gamma.to(dev), beta.to(dev), 1e-5), ln_buf=torch.empty(M, K, dtype=BF16, device=dev),
                 pw_ln=pw_ln)
        backend.sync()
        close(out, ref)


def test_gemm_folded_layernorm_tiled(backend):
    """The LNF instances of the tiled GEMM (gemm.hip ``dispatch_tile_ln``; round 5): LayerNorm folded into the weights, the row statistics
    taken inside the kernel from the A tiles as they pass through LDS (shifted sums) -- the K = 640 / 1280 linears of UNet levels 1-3.
    All three epilogues (store, GEGLU, q | k | v^T) on every instance, ragged M, rows with |mean| = 30 std mixed in; reference:
    LayerNorm (fp64) -> GEMM (fp64) on the same bf16 rows."""
    dev = backend.device
    g = torch.Generator().manual_seed(290)
    for K, M, N in ([(128, 200, 256), (192, 96, 256)] if backend.is_emu else [(640, 11264, 1920), (1280, 2816 - 24, 1280), (1280, 704, 3840)]):
        gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
        a = torch.randn(M, K, generator=g) * (torch.rand(M, 1, generator=g) * 3 + 0.2)
        a[::7] += 30.0 * a[::7].std(-1, keepdim=True)                    # rows whose |mean| >> std
        a = a.to(BF16)
        ln64 = F.layer_norm(a.double(), (K,), gamma.double(), beta.double(), 1e-5)
        # ---- store
        w = rnd(N, K, seed=291, scale=1 / math.sqrt(K))
        bias = torch.randn(N, generator=g)
        pw = ops.pack_linear(w.float(), bias, dev)
        pw_ln = ops.pack_linear_ln(w.float(), bias, gamma, beta, dev)
        ref = ln64 @ w.double().t() + bias.double()
        for tile in ops.LN_TILED_TILES:
            if pw.Npad % ops.TILE_SHAPES[tile][1]:
                continue
            out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
            ops.gemm(a.to(dev), pw, out, tile=tile, ln=(gamma.to(dev), beta.to(dev), 1e-5), ln_buf=torch.empty(M, K, dtype=BF16, device=dev), pw_ln=pw_ln)
            backend.sync()
            close(out, ref)
        # ---- GEGLU (64-wide wave tiles)
        D = N // 2
        wg = rnd(2 * D, K, seed=292, scale=1 / math.sqrt(K))
        bg = torch.randn(2 * D, generator=g) * 0.5
        pg, pg_ln = ops.pack_geglu(wg.float(), bg, dev), ops.pack_geglu_ln(wg.float(), bg, gamma, beta, dev)
        h, gt = (ln64 @ wg.double().t() + bg.double()).chunk(2, -1)
        for tile in (18, 4, 7, 17, 26):
            if pg.Npad % ops.TILE_SHAPES[tile][1]:
                continue
            out = torch.full((M, D), float("nan"), dtype=BF16, device=dev)
            ops.gemm(a.to(dev), pg, out, epilogue=ops.EPI_GEGLU, tile=tile, ln=(gamma.to(dev), beta.to(dev), 1e-5),
                     ln_buf=torch.empty(M, K, dtype=BF16, device=dev), pw_ln=pg_ln)
            backend.sync()
            close(out, h * F.gelu(gt))
        # ---- q | k | v^T: tokens per batch entry a multiple of 32, M a multiple of 32
        Bq = 2
        T = (M // Bq) // 32 * 32
        Mq, Cc = Bq * T, 128 if backend.is_emu else N // 3 // 64 * 64
        pr = ln64[:Mq] @ rnd(3 * Cc, K, seed=293, scale=1 / math.sqrt(K)).double().t()
        wq = rnd(3 * Cc, K, seed=293, scale=1 / math.sqrt(K))
        pq, pq_ln = ops.pack_linear(wq.float(), None, dev), ops.pack_linear_ln(wq.float(), None, gamma, beta, dev)
        for tile in (18, 26, 2):
            if pq.Npad % ops.TILE_SHAPES[tile][1]:
                continue
            qk = torch.full((Mq, 2 * Cc), float("nan"), dtype=BF16, device=dev)
            vt = torch.zeros(Bq, Cc, T + 8, dtype=BF16, device=dev)
            ops.gemm(a[:Mq].to(dev), pq, qk, rows_per_batch=T, epilogue=ops.EPI_SPLIT_VT, out2=vt, vt_col0=2 * Cc, tile=tile,
                     ln=(gamma.to(dev), beta.to(dev), 1e-5), ln_buf=torch.empty(Mq, K, dtype=BF16, device=dev), pw_ln=pq_ln)
            backend.sync()
            close(qk, pr[:, : 2 * Cc])
            close(vt[:, :, :T], pr[:, 2 * Cc:].view(Bq, T, Cc).permute(0, 2, 1))
    # what the instances do not implement is refused, not silently mis-computed
    with pytest.raises(RuntimeError):
        ops.gemm(a.to(dev), pw, torch.empty(M, N, dtype=BF16, device=dev), tile=21, ln=(gamma.to(dev), beta.to(dev), 1e-5),
                 ln_buf=torch.empty(M, K, dtype=BF16, device=dev), pw_ln=pw_ln)


def test_gemm_row_stats_producer_and_consumer(backend):
    """Round 5: the LayerNorm statistics of a row travel from the linear that WRITES the row to the linear that reads it.  Producer
    (gemm_ext.hip EXT = 3, ``row_stats=``): a STORE launch with bias + residual also leaves {sum, M2} of every 32-column run of the
    bf16 values it stores -- compared element for element with the same quantities taken from its output tensor; consumer (EXT = 2):
    the folded-LayerNorm GEMM merges them (Chan) instead of taking statistics in its K loop -- against LayerNorm (fp64) -> GEMM
    (fp64) of the stored rows, and against the in-loop form (EXT = 1) of the same tile."""
    dev = backend.device
    g = torch.Generator().manual_seed(390)
    for (M, K0, C, N) in ([(200, 64, 128, 256)] if backend.is_emu else [(11264, 640, 640, 1920), (2816 - 24, 1280, 1280, 1280)]):
        x = rnd(M, K0, seed=391)
        res = (torch.randn(M, C, generator=g) * 2 + 8.0 * torch.randn(M, 1, generator=g)).to(BF16)   # rows with a large common offset
        w0 = rnd(C, K0, seed=392, scale=1 / math.sqrt(K0))
        b0 = torch.randn(C, generator=g)
        pw0 = ops.pack_linear(w0.float(), b0, dev)
        gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
        w1 = rnd(N, C, seed=393, scale=1 / math.sqrt(C))
        b1 = torch.randn(N, generator=g)
        pw1, pw1_ln = ops.pack_linear(w1.float(), b1, dev), ops.pack_linear_ln(w1.float(), b1, gamma, beta, dev)
        for ptile in ops.STATS_TILES:
            if pw0.Npad % ops.TILE_SHAPES[ptile][1]:
                continue
            t = torch.full((M, C), float("nan"), dtype=BF16, device=dev)
            stats = torch.full((M, C // 32, 2), float("nan"), dtype=torch.float32, device=dev)
            ops.gemm(x.to(dev), pw0, t, residual=res.to(dev), res_mod=M, tile=ptile, row_stats=stats)
            backend.sync()
            assert ops.row_stats_valid(stats), ptile
            close(t, x.float() @ w0.float().t() + b0 + res.float())
            want = ops.row_stats_reference(t.cpu())
            got = stats.cpu()
            assert torch.isfinite(got).all(), ptile
            assert (got[..., 0] - want[..., 0]).abs().max() <= 1e-4 * want[..., 0].abs().max() + 1e-4, ptile
            assert (got[..., 1] - want[..., 1]).abs().max() <= 1e-3 * want[..., 1].abs().max() + 1e-3, ptile
        # ---- consumer on the last producer's tensor + partials
        ref = F.layer_norm(t.cpu().double(), (C,), gamma.double(), beta.double(), 1e-5) @ w1.double().t() + b1.double()
        for ctile in (18, 4, 7, 8, 2, 17, 26) + ops.LN_PARTIALS_TILES:
            if pw1.Npad % ops.TILE_SHAPES[ctile][1]:
                continue
            outs = []
            for st in (stats, None) if ctile not in ops.LN_PARTIALS_TILES else (stats, stats):   # (tile 23: the partials form only)
                out = torch.full((M, N), float("nan"), dtype=BF16, device=dev)
                ops.gemm(t, pw1, out, tile=ctile, ln=(gamma.to(dev), beta.to(dev), 1e-5), ln_buf=torch.empty(M, C, dtype=BF16, device=dev),
                         pw_ln=pw1_ln, row_stats=st)
                backend.sync()
                close(out, ref)
                outs.append(out.float().cpu())
            assert (outs[0] - outs[1]).abs().max() <= 2e-2 * ref.abs().max(), ctile     # partials vs in-loop statistics: same numbers, other order
        # ---- GEGLU through the partials form on the 64-wide full-row tiles (round 6: tile 23 = 176 x 256 carries level 1's projection)
        D = N // 2
        wg = rnd(2 * D, C, seed=394, scale=1 / math.sqrt(C))
        bg = torch.randn(2 * D, generator=g) * 0.5
        pg, pg_ln = ops.pack_geglu(wg.float(), bg, dev), ops.pack_geglu_ln(wg.float(), bg, gamma, beta, dev)
        hh, gt = (F.layer_norm(t.cpu().double(), (C,), gamma.double(), beta.double(), 1e-5) @ wg.double().t() + bg.double()).chunk(2, -1)
        got = {}
        for ctile in (26, 23):
            if pg.Npad % ops.TILE_SHAPES[ctile][1]:
                continue
            out = torch.full((M, D), float("nan"), dtype=BF16, device=dev)
            ops.gemm(t, pg, out, epilogue=ops.EPI_GEGLU, tile=ctile, ln=(gamma.to(dev), beta.to(dev), 1e-5), ln_buf=torch.empty(M, C, dtype=BF16, device=dev),
                     pw_ln=pg_ln, row_stats=stats)
            backend.sync()
            close(out, hh * F.gelu(gt))
            got[ctile] = out.clone()
        if len(got) == 2:
            assert torch.equal(got[26], got[23])
    # a configuration without a producer instance leaves the buffer alone and says so
    st2 = torch.zeros(M, C // 32, 2, dtype=torch.float32, device=dev)
    ops.gemm(x.to(dev), pw0, t, tile=21 if pw0.Npad % 320 == 0 else 1, row_stats=st2) if pw0.Npad % 128 == 0 else None
    assert not ops.row_stats_valid(st2)


def _all_bf16_in(lo: float, hi: float, stride: int = 1) -> torch.Tensor:
    bits = torch.arange(0, 1 << 16, dtype=torch.int32)
    v = bits.to(torch.int16).view(BF16)
    v = v[torch.isfinite(v.float()) & (v.float() >= lo) & (v.float() <= hi)]
    return v[::stride].contiguous()


def test_gelu_accuracy(backend):
    """The kernels' erf-GELU (pcdm_device.h ``gelu_erf_f`` / ``gelu_erf_f2``: max(x, 0) - |x| 2^Q(|x|), Q a degree-5 fit of
    log2 Phi(-t); tools/fit_gelu.py) against ``F.gelu`` in fp64 on EVERY bf16 value in [-9, 9]: |err| <= 1e-6 absolute, <= 2^-11
    relative wherever |gelu(x)| >= 2e-3 (VERDICT r3 #5 asked for the bound to be stated and tested).  Scalar form: pcdm_gemm's generic
    epilogue with act = GELU, identity weights, fp32 output.  Two-wide form: the GEGLU epilogues of gemm.hip and rowgemm.hip with
    h = 1 (zero weights, bias 1) -- bf16 output, so the bound there adds the result's own rounding (2^-8 relative)."""
    dev = backend.device
    x = _all_bf16_in(-9.0, 9.0, stride=23 if backend.is_emu else 1)
    exact = F.gelu(x.double())
    # ---- scalar form, fp32 out
    K = N = 64
    M = (x.numel() + K - 1) // K
    xp = torch.zeros(M * K, dtype=BF16)
    xp[: x.numel()] = x
    a = xp.view(M, K)
    pw = ops.pack_linear(torch.eye(N, K), None, dev)
    out = torch.empty(1, N, M, dtype=torch.float32, device=dev)
    ops.gemm(a.to(dev), pw, out, rows_per_batch=M, epilogue=ops.EPI_NCHW_F32, act=ops.ACT_GELU, tile=2)
    backend.sync()
    got = out[0].t().reshape(-1)[: x.numel()].double().cpu()
    err = (got - exact).abs()
    assert err.max().item() <= 1e-6, err.max().item()
    big = exact.abs() >= 2e-3
    assert (err[big] / exact[big].abs()).max().item() <= 2.0 ** -11
    # ---- two-wide form through the GEGLU epilogues (tiled kernel: K = 64; A-in-registers kernel: K = 320)
    for (Kg, tile) in ((64, 4), (320, 34)):
        D = Kg
        Mg = (x.numel() + D - 1) // D
        xg = torch.zeros(Mg * D, dtype=BF16)
        xg[: x.numel()] = x
        w = torch.cat([torch.zeros(D, Kg), torch.eye(D, Kg)], 0)            # rows [h | gate]: h = bias = 1, gate = x
        bias = torch.cat([torch.ones(D), torch.zeros(D)])
        pg = ops.pack_geglu(w, bias, dev)
        og = torch.empty(Mg, D, dtype=BF16, device=dev)
        ops.gemm(xg.view(Mg, D).to(dev), pg, og, epilogue=ops.EPI_GEGLU, tile=tile)
        backend.sync()
        gg = og.reshape(-1)[: x.numel()].double().cpu()
        # bf16 output: the exact value's own rounding (half an ulp <= 2^-8 relative) + the 1e-6 of the approximation
        assert ((gg - exact).abs() <= 2.0 ** -8 * exact.abs() + 1e-6).all(), (Kg, tile, (gg - exact).abs().max().item())


@pytest.mark.parametrize("case", ["conv1_temb", "conv2_res_concat", "two_pass", "cluster"])
def test_splitk_reduce_folded_into_groupnorm(backend, case):
    """``pcdm_gemm(defer_reduce=1)`` + ``pcdm_groupnorm_splitk``: the GroupNorm that follows a split-K convolution reduces the fp32
    partial slabs itself (VERDICT r3 next-round #1b).  Same arithmetic in the same order as the reduce kernel, so the result -- and the
    bf16 pre-norm tensor it writes for the residual / skip readers -- must be BIT-IDENTICAL to reduce-then-normalise; also checked
    against fp32 PyTorch.  Cases: conv1 (+ bias + time-embedding row, tensor not stored), conv2 (+ residual, stored, second concat
    source), a slab too large for the single-pass kernels (falls back to reduce + two-kernel norm), and (GPU only) the shape that takes
    the in-launch cluster exchange."""
    dev = backend.device
    if case == "cluster" and backend.is_emu:
        pytest.skip("the cluster kernel's workgroups wait for each other: GPU only")
    if case == "conv1_temb":
        B, H, Wd, Cin, Cout, C2, G, tile, sk = (2, 5, 6, 64, 64, 0, 8, 2, 3) if backend.is_emu else (8, 8, 11, 1280, 1280, 0, 32, 4, 8)
    elif case == "conv2_res_concat":
        B, H, Wd, Cin, Cout, C2, G, tile, sk = (2, 5, 6, 64, 128, 64, 8, 2, 2) if backend.is_emu else (8, 16, 22, 1280, 1280, 640, 32, 21, 4)
    elif case == "two_pass":
        B, H, Wd, Cin, Cout, C2, G, tile, sk = (1, 50, 60, 64, 64, 0, 2, 2, 2) if backend.is_emu else (2, 128, 176, 64, 64, 0, 4, 2, 2)
    else:   # level 1: HW = 1408, C = 640 -> 128 slabs -> cluster kernel
        B, H, Wd, Cin, Cout, C2, G, tile, sk = (8, 32, 44, 640, 640, 0, 32, 21, 2)
    HW, M = H * Wd, B * H * Wd
    x = rnd(B, H, Wd, Cin, seed=201)
    w = rnd(Cout, Cin, 3, 3, seed=202, scale=1 / math.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(203))
    temb = torch.randn(B, Cout, generator=torch.Generator().manual_seed(204)) if case == "conv1_temb" else None
    res = rnd(M, Cout, seed=205) if case != "conv1_temb" else None
    x2 = rnd(M, C2, seed=206) * 2 if C2 else None
    C = Cout + C2
    gamma = (torch.rand(C, generator=torch.Generator().manual_seed(207)) + 0.5).to(dev)
    beta = (torch.randn(C, generator=torch.Generator().manual_seed(208)) * 0.2).to(dev)
    pw = ops.pack_conv3x3(w.float(), bias, dev)
    cv = dict(B=B, Hi=H, Wi=Wd, Ho=H, Wo=Wd)
    kw = dict(conv=cv, tile=tile, split_k=sk)
    if temb is not None:
        kw.update(rowvec=temb.to(dev), rows_per_batch=HW)
    if res is not None:
        kw.update(residual=res.to(dev), res_mod=M)
    x2d = None if x2 is None else x2.to(dev)
    store = case != "conv1_temb"
    # reference path: reduce kernel, then the plain GroupNorm
    pre_ref = torch.empty(M, Cout, dtype=BF16, device=dev)
    ops.gemm(x.to(dev), pw, pre_ref, **kw)
    y_ref = torch.empty(M, C, dtype=BF16, device=dev)
    ws = ops.groupnorm_ws(B, C, dev)
    ops.groupnorm(pre_ref, x2d, B, HW, G, 1e-5, gamma, beta, True, y_ref, ws)
    # folded path
    pre = torch.full((M, Cout), float("nan"), dtype=BF16, device=dev)
    d = ops.gemm(x.to(dev), pw, pre, defer_reduce=store, **kw)
    assert isinstance(d, ops.DeferredGemm) and d.split_k == sk
    y = torch.full((M, C), float("nan"), dtype=BF16, device=dev)
    ops.groupnorm(d, x2d, B, HW, G, 1e-5, gamma, beta, True, y, ws)
    backend.sync()
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
    if store or case == "two_pass":
        assert torch.equal(pre.view(torch.int16), pre_ref.view(torch.int16))
    else:
        assert torch.isnan(pre.float()).all()   # never written: only the norm reads this tensor
    # ... and against fp32 PyTorch
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    if temb is not None:
        ref = ref + temb.repeat_interleave(HW, 0)
    if res is not None:
        ref = ref + res.float()
    ref = ref.to(BF16).float()
    if x2 is not None:
        ref = torch.cat([ref, x2.float()], 1)
    gn = F.silu(F.group_norm(ref.view(B, HW, C).permute(0, 2, 1), G, gamma.cpu(), beta.cpu(), 1e-5)).permute(0, 2, 1)
    close(y.view(B, HW, C), gn, tol=2e-2)


def test_conv_dup_rows_shared_cfg_prefix(backend):
    """``pcdm_gemm_params.dup_rows`` (VERDICT r3 next-round #1a): one contraction of a 3x3 convolution, two epilogues -- the second writes
    rows m + dup_rows with THEIR row-vector (time embedding) and residual rows.  Must be BIT-IDENTICAL to the same tile configuration run
    on the doubled batch (the two CFG halves carry the same input), for conv_in's form (+ residual) and conv1's (+ bias + temb row)."""
    dev = backend.device
    for (Bh, H, Wd, Cin, Cout, tiles, form) in ([(1, 6, 8, 64, 64, (2, 5), "temb"), (2, 4, 8, 64, 128, (2, 4), "res")] if backend.is_emu else
                                                [(4, 64, 88, 320, 320, (21, 5, 11), "temb"), (4, 64, 88, 64, 320, (21, 10), "res"),
                                                 (3, 13, 11, 128, 192, (2, 5), "temb")]):
        HW, Mh = H * Wd, Bh * H * Wd
        if form == "temb" and HW < 32:
            continue
        x = rnd(Bh, H, Wd, Cin, seed=301)
        w = rnd(Cout, Cin, 3, 3, seed=302, scale=1 / math.sqrt(9 * Cin))
        bias = torch.randn(Cout, generator=torch.Generator().manual_seed(303))
        temb = torch.randn(2 * Bh, Cout, generator=torch.Generator().manual_seed(304)).to(dev) if form == "temb" else None
        res = rnd(2 * Mh, Cout, seed=305).to(dev) if form == "res" else None
        pw = ops.pack_conv3x3(w.float(), bias, dev)
        x2 = torch.cat([x, x]).to(dev)                      # the doubled batch: both halves the same input
        for tile in tiles:
            if pw.Npad % ops.TILE_SHAPES[tile][1]:
                continue
            kw_full, kw_dup = dict(tile=tile), dict(tile=tile, dup_rows=Mh)
            if temb is not None:
                kw_full.update(rowvec=temb, rows_per_batch=HW); kw_dup.update(rowvec=temb, rows_per_batch=HW)
            if res is not None:
                kw_full.update(residual=res, res_mod=2 * Mh); kw_dup.update(residual=res[:Mh], res_mod=Mh)
            ref = torch.empty(2 * Mh, Cout, dtype=BF16, device=dev)
            ops.gemm(x2, pw, ref, conv=dict(B=2 * Bh, Hi=H, Wi=Wd, Ho=H, Wo=Wd), **kw_full)
            out = torch.full((2 * Mh, Cout), float("nan"), dtype=BF16, device=dev)
            ops.gemm(x2[:Bh], pw, out, conv=dict(B=Bh, Hi=H, Wi=Wd, Ho=H, Wo=Wd), **kw_dup)
            backend.sync()
            assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), (form, tile, (out.float() - ref.float()).abs().max())
    # refused where the lean epilogue cannot take it: linear, split-K, an activation
    pwl = ops.pack_linear(rnd(64, 64, seed=1).float(), None, dev)
    with pytest.raises((RuntimeError, AssertionError)):
        ops.gemm(rnd(64, 64, seed=2).to(dev), pwl, torch.empty(128, 64, dtype=BF16, device=dev), dup_rows=64, tile=2)
