#!/usr/bin/env python3
"""GroupNorm statistics from the producing GEMM (pcdm_gemm_params.stats_out + pcdm_groupnorm_stats) against the plain pair
(GEMM, pcdm_groupnorm) on the producer -> norm pairs of levels 0 and 1 of the stage-2 UNet (UNet batch 8, latent 64x88): each pair
timed as a replayed hipGraph of 10 back-to-back (GEMM, norm) pairs on tensors larger than the 256 MB MALL in total, so that the norm
reads what the GEMM wrote from where the step would find it.  Per pair: GEMM and norm time with and without the feature.

    python tools/bench_gn_stats.py > gpurun_out/bench_gn_stats.txt"""
from __future__ import annotations

import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")


def timed(fn, rep=10, outer=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rep):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(outer):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (rep * outer) * 1e3


def main():
    B, G = 8, 32
    # (name, HW (h, w), Cin, Cout, conv?, residual?, temb?, instances per forward)
    pairs = [("L0 conv1 -> norm2", (64, 88), 320, 320, True, False, True, 4),
             ("L0 conv2+res -> transformer norm", (64, 88), 320, 320, True, True, False, 5),
             ("L0 proj_out+res -> norm1 / norm_out", (64, 88), 320, 320, False, True, False, 2),
             ("L0 conv1 (K 5760) -> norm2", (64, 88), 640, 320, True, False, True, 2),
             ("L1 conv1 -> norm2", (32, 44), 640, 640, True, False, True, 3),
             ("L1 conv2+res -> transformer norm", (32, 44), 640, 640, True, True, False, 5),
             ("L1 proj_out+res -> norm1", (32, 44), 640, 640, False, True, False, 1)]
    tot_off = tot_on = 0.0
    for name, (h, w), cin, cout, conv, res, temb, n in pairs:
        HW, M = h * w, B * h * w
        x = (torch.randn(B, h, w, cin, device=dev) if conv else torch.randn(M, cin, device=dev)).to(BF16)
        if conv:
            pw = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3) / math.sqrt(9 * cin), torch.randn(cout), dev)
            kw = dict(conv=dict(B=B, Hi=h, Wi=w, Ho=h, Wo=w), rows_per_batch=HW)
        else:
            pw = ops.pack_linear(torch.randn(cout, cin) / math.sqrt(cin), torch.randn(cout), dev)
            kw = dict(rows_per_batch=HW)
        if temb:
            kw["rowvec"] = torch.randn(B, cout, device=dev)
        if res:
            kw.update(residual=torch.randn(M, cout, device=dev).to(BF16), res_mod=M)
        out = torch.empty(M, cout, dtype=BF16, device=dev)
        y = torch.empty(M, cout, dtype=BF16, device=dev)
        ws = ops.groupnorm_ws(B, cout, dev)
        gm, bt = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        ops.gemm(x, pw, out, **kw)                      # (tunes the shape if the table does not hold it)
        torch.cuda.synchronize()
        t_g0 = timed(lambda: ops.gemm(x, pw, out, **kw))
        t_g1 = timed(lambda: ops.gemm(x, pw, out, gn_stats=G, **kw))
        sg = ops.gemm(x, pw, out, gn_stats=G, **kw)
        is_stats = isinstance(sg, ops.StatsGemm)
        t_p0 = timed(lambda: ops.groupnorm(ops.gemm(x, pw, out, **kw), None, B, HW, G, 1e-5, gm, bt, True, y, ws))
        t_p1 = timed(lambda: ops.groupnorm(ops.gemm(x, pw, out, gn_stats=G, **kw), None, B, HW, G, 1e-5, gm, bt, True, y, ws))
        t_n0 = timed(lambda: ops.groupnorm(out, None, B, HW, G, 1e-5, gm, bt, True, y, ws))       # the norms alone (tensor from the MALL)
        t_n1 = timed(lambda: ops.groupnorm(sg, None, B, HW, G, 1e-5, gm, bt, True, y, ws)) if is_stats else float("nan")
        print(f"{'':38s}      norm alone {t_n0:6.2f} -> {t_n1:6.2f} us", flush=True)
        print(f"{name:38s} x{n}: stats instance {is_stats!s:5s} | GEMM {t_g0:7.2f} -> {t_g1:7.2f} us | GEMM + norm {t_p0:7.2f} -> {t_p1:7.2f} us "
              f"(norm {t_p0 - t_g0:6.2f} -> {t_p1 - t_g1:6.2f})", flush=True)
        tot_off += n * t_p0
        tot_on += n * t_p1
    print(f"sum over the pairs of one forward: {tot_off:.1f} -> {tot_on:.1f} us ({tot_off - tot_on:+.1f} us per denoise step)")


if __name__ == "__main__":
    main()
