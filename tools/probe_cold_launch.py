#!/usr/bin/env python3
"""What is cold when a small GEMM runs inside the denoise step?  Three cache states per problem, one launch timed at a time (HIP events,
median of 15): HOT = launched back to back (what the back-to-back tuner sees); COLD = L2 / Infinity Cache evicted before the launch (a 1 GiB
fill: every operand comes from HBM); COLD+W = evicted, then the WEIGHTS alone are read once by another kernel (they sit in the Infinity
Cache / L2 again), activations still cold; COLD+A = evicted, then the ACTIVATIONS (A, residual) alone are read.  In the step the weights are
always cold (1.7 GB of them pass per step) and the activations were written by the previous kernel: if COLD+W recovers most of HOT, a weight
prefetch ahead of the launch is worth building; if COLD+A does, it is the producer -> consumer hand-over.   MI355X; ~1 min."""
from __future__ import annotations

import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)   # 1 GiB


def evict():
    flush.fill_(1.0)


def touch(*ts):
    for t in ts:
        if t is not None:
            t.view(-1)[: t.numel() // 8 * 8].view(torch.int64).sum()


def timed(fn, pre):
    out = []
    for _ in range(15):
        pre()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(out)


def main():
    g = torch.Generator().manual_seed(0)
    cases = [("o1 level 1  (11264, 640, 640) + residual", 11264, 640, 640, None),
             ("o1 level 2  (2816, 1280, 1280) + residual", 2816, 1280, 1280, None),
             ("o1 level 0  (45056, 320, 320) + residual", 45056, 320, 320, None),
             ("ffo level 2 (2816, 1280, 6400) two-source", 2816, 1280, 6400, 5120),
             ("conv level 3 (704, 1280, 11520) 3x3", 704, 1280, 11520, "conv")]
    print(f"{'problem':46s} {'HOT':>8s} {'COLD':>8s} {'COLD+W':>8s} {'COLD+A':>8s}   us per launch (event bracket ~2-4 us included)")
    for name, M, N, K, kind in cases:
        if kind == "conv":
            B, H, W, C = 8, 8, 11, 1280
            x = (torch.randn(B, H, W, C, generator=g) * 0.5).to(BF).to(dev)
            pw = ops.pack_conv3x3(torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5, torch.zeros(N), dev)
            out = torch.empty(M, N, dtype=BF, device=dev)
            acts = (x,)
            fn = lambda: ops.gemm(x, pw, out, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W))   # noqa: E731
        else:
            k1 = kind or K
            a = (torch.randn(M, k1, generator=g) * 0.5).to(BF).to(dev)
            a2 = (torch.randn(M, K - k1, generator=g) * 0.5).to(BF).to(dev) if kind else None
            res = (torch.randn(M, N, generator=g) * 0.5).to(BF).to(dev)
            pw = ops.pack_linear(torch.randn(N, K, generator=g) / K ** 0.5, torch.zeros(N), dev)
            out = torch.empty(M, N, dtype=BF, device=dev)
            acts = (a, a2, res)
            fn = lambda: ops.gemm(a, pw, out, a2=a2, residual=res, res_mod=M)   # noqa: E731
        fn(); fn()
        torch.cuda.synchronize()
        hot = timed(fn, lambda: None)
        cold = timed(fn, evict)
        cold_w = timed(fn, lambda: (evict(), touch(pw.w)))
        cold_a = timed(fn, lambda: (evict(), touch(*acts)))
        print(f"{name:46s} {hot:8.1f} {cold:8.1f} {cold_w:8.1f} {cold_a:8.1f}", flush=True)


if __name__ == "__main__":
    main()
