#!/usr/bin/env python3
"""Bound measurement for "GroupNorm statistics from the producer + a normalise-only pass" (bound-first rule): what a
one-pass stream of the tensor costs (the LayerNorm rows kernel) against the product's single-launch GroupNorm (cluster kernel), then -- with
PCDM_GN_PRODUCER_STATS=1 -- what the group sums cost the producing convolution and what the normalise-only launch takes; replayed hipGraphs
of 20 calls, UNet batch 8.  (profiles/r5_bench_gn_apply.txt also has a line of a stand-alone apply kernel that was removed again.)

    python tools/bench_gn_apply.py > gpurun_out/bench_gn_apply.txt
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402
from tools.bench_ln_gemm import timed  # noqa: E402

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def main():
    """the bound: the product's single-launch GroupNorm against the LayerNorm rows kernel (a one-pass stream of the same tensor)"""
    g = torch.Generator().manual_seed(0)
    for B, HW, Cc, n in [(8, 5632, 320, 12), (8, 1408, 640, 11), (8, 5632, 640, 2)]:
        x = (torch.randn(B * HW, Cc, generator=g) + 0.5).to(BF16).to(dev)
        gamma, beta = (torch.rand(Cc, generator=g) + 0.5).to(dev), (torch.randn(Cc, generator=g) * 0.2).to(dev)
        y2, y3 = torch.empty_like(x), torch.empty_like(x)
        ws = ops.groupnorm_ws(B, Cc, dev)
        t_gn = timed(lambda: ops.groupnorm(x, None, B, HW, 32, 1e-5, gamma, beta, True, y2, ws))
        t_ln = timed(lambda: ops.layernorm(x, gamma, beta, 1e-5, y3))
        mb = 2.0 * B * HW * Cc * 2 / 1e6
        print(f"B {B} HW {HW} C {Cc} ({mb:.1f} MB read + written) x{n} per step: GroupNorm launch (statistics + apply) {t_gn:6.2f} us | "
              f"LayerNorm rows kernel (one streaming pass) {t_ln:6.2f} us", flush=True)


def producer():
    """what the group sums cost the convolution that writes them, and the normalise-only launch that uses them (level 0, UNet batch 8)"""
    import math
    ops.GN_PRODUCER_STATS = True     # (opt-in in the product)
    g = torch.Generator().manual_seed(1)
    B, H, W, C = 8, 64, 88, 320
    M, HW = B * H * W, H * W
    x = torch.randn(B, H, W, C, generator=g).to(BF16).to(dev)
    pw = ops.pack_conv3x3((torch.rand(C, C, 3, 3, generator=g) * 2 - 1) / math.sqrt(9 * C), torch.randn(C, generator=g) * 0.1, dev)
    res = torch.randn(M, C, generator=g).to(BF16).to(dev)
    tv = torch.randn(B, C, generator=g).to(dev)
    out = torch.empty(M, C, dtype=BF16, device=dev)
    stats = torch.empty((M + 191) // 192, 2, 32, 2, dtype=torch.float32, device=dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
    y = torch.empty_like(out)
    ws = ops.groupnorm_ws(B, C, dev)
    cv = dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W)
    for name, kw in [("conv1 + time-embedding row", dict(rowvec=tv)), ("conv2 + residual", dict(residual=res, res_mod=M))]:
        t0 = timed(lambda: ops.gemm(x, pw, out, conv=cv, rows_per_batch=HW, tile=21, **kw))
        t1 = timed(lambda: ops.gemm(x, pw, out, conv=cv, rows_per_batch=HW, tile=21, gn_stats=stats, gn_gs=10, **kw))
        assert ops.gn_stats_for(stats, out, 10)
        t2 = timed(lambda: ops.groupnorm(out, None, B, HW, 32, 1e-5, gamma, beta, True, y, ws))
        t3 = timed(lambda: ops.groupnorm(out, None, B, HW, 32, 1e-5, gamma, beta, True, y, ws, gn_stats=stats))

        def pair(st):
            ops.gemm(x, pw, out, conv=cv, rows_per_batch=HW, tile=21, gn_stats=st, gn_gs=10, **kw)
            ops.groupnorm(out, None, B, HW, 32, 1e-5, gamma, beta, True, y, ws, gn_stats=st)
        t4, t5 = timed(lambda: pair(None)), timed(lambda: pair(stats))
        print(f"level 0 {name}: convolution {t0:6.2f} -> {t1:6.2f} us with group sums | GroupNorm {t2:6.2f} -> {t3:6.2f} us normalise-only | "
              f"back to back {t4:6.2f} -> {t5:6.2f} us", flush=True)


if __name__ == "__main__":
    main()
    producer()
