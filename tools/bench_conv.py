#!/usr/bin/env python3
"""The 3x3 convolutions of the stage-2 UNet step (UNet batch 8, latent 64x88) at their tuned tile configurations, each timed as a
replayed hipGraph of 10 launches (epilogue variants: time-embedding row / residual / plain).  A/B of two library builds:
    python tools/bench_conv.py ; python tools/with_lib.py pcdms_amd/lib/libpcdm_alt.so tools/bench_conv.py"""
from __future__ import annotations

import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")


def timed(fn, rep=10, outer=8):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rep):
            fn()
    for _ in range(3):
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
    B = 8
    tot = 0.0
    # (h, w, Cin, Cout, epilogue, launches per step)
    for h, w, cin, cout, epi, n in [(64, 88, 320, 320, "temb", 3), (64, 88, 320, 320, "res", 4), (64, 88, 640, 320, "temb", 2), (64, 88, 960, 320, "temb", 1),
                                    (32, 44, 640, 640, "temb", 3), (32, 44, 640, 640, "res", 5), (32, 44, 1280, 640, "temb", 1), (16, 22, 1280, 1280, "res", 7)]:
        M, HW = B * h * w, h * w
        x = torch.randn(B, h, w, cin, device=dev).to(BF16)
        pw = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3) / math.sqrt(9 * cin), torch.randn(cout), dev)
        kw = dict(conv=dict(B=B, Hi=h, Wi=w, Ho=h, Wo=w), rows_per_batch=HW)
        if epi == "temb":
            kw["rowvec"] = torch.randn(B, cout, device=dev)
        else:
            kw.update(residual=torch.randn(M, cout, device=dev).to(BF16), res_mod=M)
        out = torch.empty(M, cout, dtype=BF16, device=dev)
        ops.gemm(x, pw, out, **kw)
        torch.cuda.synchronize()
        us = min(timed(lambda: ops.gemm(x, pw, out, **kw)) for _ in range(2))
        key = (M, pw.Npad, pw.K, 1, 1, 0, 0, False, epi == "res")
        tf = 2.0 * M * cout * 9 * cin / us / 1e6
        print(f"conv3x3 {cin:4d}->{cout:4d} @{h}x{w} +{epi:4s} tile {ops._TUNED.get(key)}: {us:7.2f} us  {tf:6.0f} TF/s  x{n}", flush=True)
        tot += n * us
    print(f"sum over one step: {tot:.1f} us")


if __name__ == "__main__":
    main()
