"""Per-kernel micro-benchmarks at the real stage-2 shapes (UNet batch 8, latent 64x88).

Run on the GPU box:  python tools/bench_ops.py [--out gpurun_out/bench_ops.json]
Prints achieved TFLOP/s (MFMA kernels) or GB/s (HBM-bound kernels) per shape.
"""
from __future__ import annotations

import argparse
import json
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    res = []
    B = 8

    def rec(name, secs, flops=0, bytes_=0):
        r = dict(name=name, us=secs * 1e6, tflops=flops / secs / 1e12 if flops else None,
                 gbs=bytes_ / secs / 1e9 if bytes_ else None)
        res.append(r)
        print(f"{name:58s} {r['us']:9.1f} us  " + (f"{r['tflops']:8.1f} TF/s" if flops else "") +
              (f"{r['gbs']:8.0f} GB/s" if bytes_ else ""), flush=True)

    # ---- conv3x3 per level (Cin -> Cout @ HxW)
    convs = [(64, 88, 320, 320), (64, 88, 640, 320), (64, 88, 960, 320), (32, 44, 640, 640), (32, 44, 1280, 640),
             (32, 44, 1920, 640), (16, 22, 1280, 1280), (16, 22, 2560, 1280), (8, 11, 1280, 1280), (8, 11, 2560, 1280)]
    for (H, W, Ci, Co) in convs:
        x = torch.randn(B, H, W, Ci, device=dev).to(BF16)
        pw = ops.pack_conv3x3(torch.randn(Co, Ci, 3, 3) / math.sqrt(9 * Ci), torch.randn(Co), dev)
        out = torch.empty(B * H * W, Co, dtype=BF16, device=dev)
        cv = dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W)
        for tile in ((0,) if args.quick else (1, 2, 3, 4)):
            if tile in (1, 4) and pw.Npad % 128:
                continue
            t = timeit(lambda: ops.gemm(x, pw, out, conv=cv, tile=tile))
            rec(f"conv3x3 {Ci}->{Co} @{H}x{W} tile{tile}", t, flops=2.0 * B * H * W * Co * 9 * Ci)
    # ---- linears (M, K, N)
    lins = [(45056, 320, 320), (45056, 320, 960), (11264, 640, 640), (11264, 640, 1920), (2816, 1280, 1280),
            (2816, 1280, 3840), (704, 1280, 1280), (45056, 1280, 320), (11264, 2560, 640), (2816, 5120, 1280),
            (2064, 1024, 640)]
    for (M, K, N) in lins:
        a = torch.randn(M, K, device=dev).to(BF16)
        pw = ops.pack_linear(torch.randn(N, K) / math.sqrt(K), torch.randn(N), dev)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        for tile in ((0,) if args.quick else (1, 2, 3, 4)):
            if tile in (1, 4) and pw.Npad % 128:
                continue
            t = timeit(lambda: ops.gemm(a, pw, out, tile=tile))
            rec(f"linear M{M} K{K} N{N} tile{tile}", t, flops=2.0 * M * K * N)
    for (M, K, D) in [(45056, 320, 1280), (11264, 640, 2560), (2816, 1280, 5120)]:
        a = torch.randn(M, K, device=dev).to(BF16)
        pw = ops.pack_geglu(torch.randn(2 * D, K) / math.sqrt(K), torch.randn(2 * D), dev)
        out = torch.empty(M, D, dtype=BF16, device=dev)
        t = timeit(lambda: ops.gemm(a, pw, out, epilogue=ops.EPI_GEGLU))
        rec(f"geglu M{M} K{K} D{D}", t, flops=2.0 * M * K * 2 * D)
    # ---- attention
    for (H, Lq, Lk) in [(5, 5632, 5632), (10, 1408, 1408), (20, 352, 352), (20, 88, 88), (5, 5632, 258), (10, 1408, 258)]:
        Cc = H * 64
        q = torch.randn(B * Lq, Cc, device=dev).to(BF16)
        k = torch.randn(B * Lk, Cc, device=dev).to(BF16)
        Lp = (Lk + 7) // 8 * 8
        vt = torch.randn(B, Cc, Lp, device=dev).to(BF16)
        out = torch.empty(B * Lq, Cc, dtype=BF16, device=dev)
        t = timeit(lambda: ops.flash_attn(q, k, vt, out, B, H, Lq, Lk))
        rec(f"attn H{H} Lq{Lq} Lk{Lk}", t, flops=4.0 * B * H * Lq * Lk * 64)
    # ---- HBM-bound
    for (HW, C) in [(5632, 320), (5632, 960), (1408, 640), (352, 1280), (88, 2560)]:
        x = torch.randn(B * HW, C, device=dev).to(BF16)
        out = torch.empty_like(x)
        ws = ops.groupnorm_ws(B, C, dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        t = timeit(lambda: ops.groupnorm(x, None, B, HW, 32, 1e-5, g, b, True, out, ws))
        rec(f"groupnorm+silu HW{HW} C{C}", t, bytes_=3.0 * x.numel() * 2)  # stats read + apply read + write
    for (rows, C) in [(45056, 320), (11264, 640), (2816, 1280)]:
        x = torch.randn(rows, C, device=dev).to(BF16)
        out = torch.empty_like(x)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        t = timeit(lambda: ops.layernorm(x, g, b, 1e-5, out))
        rec(f"layernorm rows{rows} C{C}", t, bytes_=2.0 * x.numel() * 2)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
