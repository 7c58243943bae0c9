"""GEMM kernel ablation on the MI355X: full kernel vs no-steady-state-loads vs no-MFMA, per tile config,
plus square GEMM calibration points (4096^3, 8192^3) comparable with cdna_hip_programming.md's ladder.

    python tools/ablate_gemm.py > gpurun_out/ablate.log
"""
from __future__ import annotations

import math
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")


def run(name, fn_of_tile, flops, tiles):
    for tile in tiles:
        row = []
        for dbg in (0, 1, 2, 3, 8, 16):   # (8 / 16, round 6: steady state without the A / the B loads -- which operand's staging is exposed?)
            t = timeit(lambda: fn_of_tile(tile | (dbg << 8)), iters=10, warmup=2)
            row.append(t)
        print(f"{name:44s} tile{tile}  full {row[0]*1e6:8.1f} us ({flops/row[0]/1e12:6.1f} TF/s) | noload {row[1]*1e6:8.1f} us "
              f"({flops/row[1]/1e12:6.1f}) | nomfma {row[2]*1e6:8.1f} us | neither {row[3]*1e6:8.1f} us | no-A {row[4]*1e6:8.1f} us | no-B {row[5]*1e6:8.1f} us", flush=True)


def main():
    B = 8
    # PCDM_ABLATE_CONST=1: constant operands (every value 1.0) instead of N(0, 1) -- the same instruction stream with almost no operand
    # bit toggling: the difference is the chip's power management, not the kernel
    const = os.environ.get("PCDM_ABLATE_CONST") == "1"
    randn = (lambda *sh, **kw: torch.ones(*sh, **kw)) if const else torch.randn
    want = tuple(int(t) for t in sys.argv[1].split(",")) if len(sys.argv) > 1 else (21, 17, 1, 4)
    for (M, K, N) in [(8192, 8192, 8192), (45056, 1280, 320), (11264, 5760, 1280), (45056, 320, 1280)]:
        a = randn(M, K, device=dev).to(BF16)
        pw = ops.pack_linear(randn(N, K) / math.sqrt(K), torch.randn(N), dev)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        tiles = [t for t in want if pw.Npad % ops.TILE_SHAPES[t][1] == 0]
        run(f"linear M{M} K{K} N{N}", lambda tl: ops.gemm(a, pw, out, tile=tl), 2.0 * M * K * N, tiles)
    for (H, W, Ci, Co) in [(64, 88, 320, 320), (64, 88, 640, 320), (32, 44, 640, 1280), (32, 44, 1920, 640)]:
        x = randn(B, H, W, Ci, device=dev).to(BF16)
        pw = ops.pack_conv3x3(randn(Co, Ci, 3, 3) / math.sqrt(9 * Ci), torch.randn(Co), dev)
        out = torch.empty(B * H * W, Co, dtype=BF16, device=dev)
        cv = dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W)
        tiles = [t for t in want if pw.Npad % ops.TILE_SHAPES[t][1] == 0]
        run(f"conv3x3 {Ci}->{Co} @{H}x{W}", lambda tl: ops.gemm(x, pw, out, conv=cv, tile=tl), 2.0 * B * H * W * Co * 9 * Ci, tiles)


if __name__ == "__main__":
    main()
