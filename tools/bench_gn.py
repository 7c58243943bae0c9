#!/usr/bin/env python3
"""GroupNorm(+SiLU) per-instance timing for every GN shape of the stage-2 UNet (UNet batch 8, latent 64x88), timed as a
replayed hipGraph of 20 back-to-back calls (what the pipeline does).  Run twice to A/B the single-pass kernel:
    python tools/bench_gn.py                       # default dispatch
    PCDM_GN_FUSED_MAX_KB=0 python tools/bench_gn.py   # two-pass (stats + apply) everywhere
Bytes: algorithmic = read + write of the tensor once each; HBM roof 8 TB/s."""
from __future__ import annotations

import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import _lib, ops  # noqa: E402

if len(sys.argv) > 1:   # A/B: an alternative build of the library
    _lib.load(sys.argv[1])

# (HW, C1, C2, instances per UNet forward)
SHAPES = [(5632, 320, 0, 9), (5632, 640, 320, 1), (5632, 320, 320, 2), (5632, 640, 0, 0),
          (1408, 320, 0, 1), (1408, 640, 0, 11), (1408, 1280, 640, 1), (1408, 640, 640, 1), (1408, 640, 320, 1),
          (352, 640, 0, 1), (352, 1280, 0, 14), (352, 1280, 1280, 2), (352, 1280, 640, 1),
          (88, 1280, 0, 13), (88, 1280, 1280, 3)]


def main():
    dev = torch.device("cuda:0")
    B, REP = 8, 20
    rows, tot = [], 0.0
    for HW, C1, C2, n in SHAPES:
        C = C1 + C2
        x1 = torch.randn(B * HW, C1, device=dev).to(torch.bfloat16)
        x2 = torch.randn(B * HW, C2, device=dev).to(torch.bfloat16) if C2 else None
        out = torch.empty(B * HW, C, dtype=torch.bfloat16, device=dev)
        ws = ops.groupnorm_ws(B, C, dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        fn = lambda: ops.groupnorm(x1, x2, B, HW, 32, 1e-5, g, b, True, out, ws)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(REP):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            graph.replay()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / (5 * REP) * 1e3
        gbs = 2.0 * B * HW * C * 2 / (us * 1e-6) / 1e9
        rows.append(dict(HW=HW, C1=C1, C2=C2, us=round(us, 2), alg_GBps=round(gbs), frac_hbm=round(gbs / 8000, 3), per_forward=n))
        tot += us * n
        print(f"HW {HW:5d} C {C1:4d}+{C2:4d}: {us:7.2f} us  {gbs:6.0f} GB/s  x{n}", flush=True)
    print(json.dumps(dict(total_us_per_forward=round(tot, 1), rows=rows)))


if __name__ == "__main__":
    main()
