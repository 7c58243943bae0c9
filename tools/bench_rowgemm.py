#!/usr/bin/env python3
"""Level-0 (K = 320) Transformer2D linears: the A-in-registers kernel (rowgemm.hip, tiles 31..34, LayerNorm fused where the reference
has one in front) against the tiled GEMM (+ the separate LayerNorm launch).  Interleaved rounds in one process, HIP events, min / median.

    python tools/bench_rowgemm.py            (on the MI355X)"""
from __future__ import annotations

import math
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    from pcdms_amd import ops
    dev = torch.device("cuda:0")
    BF16 = torch.bfloat16
    K = 320
    g = torch.Generator().manual_seed(0)
    gamma, beta = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.3).to(dev)
    cases = [  # name, M, N(out), epilogue, ln, residual
        ("qkv  LN1 -> to_q|k|v     ", 45056, 960, ops.EPI_SPLIT_VT, True, False),
        ("ff1  LN3 -> GEGLU        ", 45056, 1280, ops.EPI_GEGLU, True, False),
        ("q2   LN2 -> to_q (cond)  ", 22528, 320, ops.EPI_STORE, True, False),
        ("o1 / proj (+ residual)   ", 45056, 320, ops.EPI_STORE, False, True),
    ]
    for name, M, N, epi, use_ln, use_res in cases:
        a = (torch.randn(M, K, generator=g) * 1.0).to(BF16).to(dev)
        if epi == ops.EPI_GEGLU:
            w = torch.randn(2 * N, K, generator=g) / math.sqrt(K)
            bb = torch.randn(2 * N, generator=g) * 0.5
            pw = ops.pack_geglu(w, bb, dev)
            pw_ln = ops.pack_geglu_ln(w, bb, gamma.cpu(), beta.cpu(), dev)
            flops = 2.0 * M * 2 * N * K
        else:
            w = torch.randn(N, K, generator=g) / math.sqrt(K)
            bb = torch.randn(N, generator=g) if epi == ops.EPI_STORE else None
            pw = ops.pack_linear(w, bb, dev)
            pw_ln = ops.pack_linear_ln(w, bb, gamma.cpu(), beta.cpu(), dev)
            flops = 2.0 * M * N * K
        res = torch.randn(M, N, generator=g).to(BF16).to(dev) if use_res else None
        kw = dict(epilogue=epi)
        if epi == ops.EPI_SPLIT_VT:
            C = N // 3
            out = torch.empty(M, 2 * C, dtype=BF16, device=dev)
            kw.update(rows_per_batch=5632, out2=torch.zeros(M // 5632, C, 5632, dtype=BF16, device=dev), vt_col0=2 * C)
        else:
            out = torch.empty(M, N, dtype=BF16, device=dev)
        if use_res:
            kw.update(residual=res, res_mod=M)
        lnb = torch.empty(M, K, dtype=BF16, device=dev)

        def baseline():
            x = ops.layernorm(a, gamma, beta, 1e-5, lnb) if use_ln else a
            ops.gemm(x, pw, out, **kw)        # committed tuning table / online tuner picks the tile
        variants = {"tiled (tuned)" + (" + LN launch" if use_ln else ""): baseline}
        for t in ops.ROWGEMM_TILES:
            if pw.Npad % ops.TILE_SHAPES[t][1]:
                continue
            def f(t=t):
                if use_ln:
                    ops.gemm(a, pw, out, tile=t, ln=(gamma, beta, 1e-5), ln_buf=lnb, pw_ln=pw_ln, **{k: v for k, v in kw.items() if k not in ("residual", "res_mod")})
                else:
                    ops.gemm(a, pw, out, tile=t, **kw)
            try:
                f()
                torch.cuda.synchronize()
            except RuntimeError:
                continue
            variants[f"rowgemm tile {t}"] = f
        baseline()
        torch.cuda.synchronize()
        ref = out.float().clone()
        times = {k: [] for k in variants}
        for k, f in variants.items():     # correctness against the tiled path
            f()
            torch.cuda.synchronize()
            err = ((out.float() - ref).norm() / ref.norm()).item()
            assert err < 3e-2, (k, err)
        for _ in range(7):
            for k, f in variants.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f()
                e0.record()
                for _ in range(5):
                    f()
                e1.record()
                e1.synchronize()
                times[k].append(e0.elapsed_time(e1) / 5 * 1e3)
        print(f"{name} M={M} N={N}:")
        for k, v in times.items():
            print(f"    {k:32s} min {min(v):7.1f} us  median {statistics.median(v):7.1f} us   {flops / min(v) / 1e6:7.1f} TF/s")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
