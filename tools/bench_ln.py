#!/usr/bin/env python3
"""LayerNorm timing on the UNet's token shapes (UNet batch 8, latent 64x88), as a replayed hipGraph of 20 calls each."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import _lib, ops  # noqa: E402

if len(sys.argv) > 1:   # A/B: an alternative build of the library
    _lib.load(sys.argv[1])

dev = torch.device("cuda:0")
for rows, C, n in [(45056, 320, 15), (11264, 640, 15), (2816, 1280, 15), (704, 1280, 3)]:
    x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    out = torch.empty_like(x)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    fn = lambda: ops.layernorm(x, g, b, 1e-5, out)  # noqa: E731
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(20):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    print(f"layernorm rows {rows:6d} C {C:5d}: {us:7.2f} us  {2.0 * rows * C * 2 / us / 1e3:7.0f} GB/s  x{n} per forward", flush=True)
