#!/usr/bin/env python3
"""Feasibility of a weight prefetch on a side stream (round 6; tools/ubench/prefetch_probe.hip).  (1) what a dense level-0 convolution loses
when a one-wave-per-CU kernel streams 29.5 MB (a level-3 convolution's weights) beside it; (2) what the level-3 convolution gains when its
weights were streamed that way after a cache eviction; (3) how long the prefetch itself takes alone.  MI355X; < 1 min."""
from __future__ import annotations

import ctypes as C
import statistics
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pcdms_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
_so = ROOT / "tools" / "ubench" / "libprefetch_probe.so"
if not _so.exists():   # (built on first use: hipcc cross-compiles without a GPU, the .so travels with gpurun's snapshot)
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", str(_so.with_name("prefetch_probe.hip")), "-o", str(_so)])
lib = C.CDLL(str(_so))
lib.prefetch_probe.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
side = torch.cuda.Stream()


def prefetch(t, waves, stream):
    assert lib.prefetch_probe(t.data_ptr(), t.numel() * t.element_size(), waves, sink.data_ptr(), stream.cuda_stream) == 0


def med(f, n=15):
    v = []
    for _ in range(n):
        v.append(f())
    return statistics.median(v)


def main():
    g = torch.Generator().manual_seed(0)
    # dense launch: level-0 conv (45056, 320, 2880)
    B, H, W, Cc = 8, 64, 88, 320
    x0 = (torch.randn(B, H, W, Cc, generator=g) * 0.5).to(BF).to(dev)
    pw0 = ops.pack_conv3x3(torch.randn(Cc, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5, torch.zeros(Cc), dev)
    o0 = torch.empty(B * H * W, Cc, dtype=BF, device=dev)
    dense = lambda: ops.gemm(x0, pw0, o0, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W))   # noqa: E731
    # weight-heavy launch: level-3 conv (704, 1280, 11520)
    x3 = (torch.randn(8, 8, 11, 1280, generator=g) * 0.5).to(BF).to(dev)
    pw3 = ops.pack_conv3x3(torch.randn(1280, 1280, 3, 3, generator=g) / (9 * 1280) ** 0.5, torch.zeros(1280), dev)
    o3 = torch.empty(704, 1280, dtype=BF, device=dev)
    small = lambda: ops.gemm(x3, pw3, o3, conv=dict(B=8, Hi=8, Wi=11, Ho=8, Wo=11))   # noqa: E731
    for _ in range(3):
        dense(); small()
    torch.cuda.synchronize()
    main_s = torch.cuda.current_stream()

    def t_dense(with_prefetch, waves=256):
        flush.fill_(1.0)
        dense()                                   # warm the conv's own operands again
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if with_prefetch:
            prefetch(pw3.w, waves, side)
        e0.record(); dense(); e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3

    def t_small(mode, waves=256):
        flush.fill_(1.0)
        x3.view(-1).view(torch.int64).sum()       # activations warm, as in the step
        if mode == "prefetched":
            prefetch(pw3.w, waves, main_s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); small(); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e3

    def t_prefetch(waves):
        flush.fill_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prefetch(pw3.w, waves, main_s); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e3

    print(f"dense level-0 conv alone                      {med(lambda: t_dense(False)):7.1f} us")
    for wv in (64, 256, 1024):
        print(f"  ... with a {wv:4d}-wave prefetch of 29.5 MB beside it {med(lambda: t_dense(True, wv)):7.1f} us   (the prefetch alone: {med(lambda: t_prefetch(wv)):6.1f} us)")
    print(f"level-3 conv, weights cold (activations warm)  {med(lambda: t_small('cold')):7.1f} us")
    for wv in (64, 256, 1024):
        print(f"  ... weights prefetched by {wv:4d} waves            {med(lambda: t_small('prefetched', wv)):7.1f} us")


if __name__ == "__main__":
    main()
