"""GEMM kernel ablation on the MI355X: full kernel vs no-steady-state-loads vs no-MFMA, per tile config,
plus square GEMM calibration points (4096^3, 8192^3) comparable with cdna_hip_programming.md's ladder.

    python tools/ablate_gemm.py > gpurun_out/ablate.log
"""
from __future__ import annotations

import math
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
        for dbg in (0, 1, 2, 3):
            t = timeit(lambda: fn_of_tile(tile | (dbg << 8)), iters=10, warmup=2)
            row.append(t)
        print(f"{name:44s} tile{tile}  full {row[0]*1e6:8.1f} us ({flops/row[0]/1e12:6.1f} TF/s) | noload {row[1]*1e6:8.1f} us "
              f"({flops/row[1]/1e12:6.1f}) | nomfma {row[2]*1e6:8.1f} us | neither {row[3]*1e6:8.1f} us", flush=True)


def main():
    B = 8
    for (M, K, N) in [(4096, 4096, 4096), (8192, 8192, 8192), (11264, 2560, 640), (45056, 1280, 320)]:
        a = torch.randn(M, K, device=dev).to(BF16)
        pw = ops.pack_linear(torch.randn(N, K) / math.sqrt(K), torch.randn(N), dev)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        tiles = [t for t in (1, 11, 4, 6, 12, 3) if not (t in (1, 4, 11) and pw.Npad % 128)]
        run(f"linear M{M} K{K} N{N}", lambda tl: ops.gemm(a, pw, out, tile=tl), 2.0 * M * K * N, tiles)
    for (H, W, Ci, Co) in [(32, 44, 1920, 640), (64, 88, 640, 320), (16, 22, 1280, 1280)]:
        x = torch.randn(B, H, W, Ci, device=dev).to(BF16)
        pw = ops.pack_conv3x3(torch.randn(Co, Ci, 3, 3) / math.sqrt(9 * Ci), torch.randn(Co), dev)
        out = torch.empty(B * H * W, Co, dtype=BF16, device=dev)
        cv = dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W)
        tiles = [t for t in (1, 11, 4, 6, 12, 3) if not (t in (1, 4, 11) and pw.Npad % 128)]
        run(f"conv3x3 {Ci}->{Co} @{H}x{W}", lambda tl: ops.gemm(x, pw, out, conv=cv, tile=tl), 2.0 * B * H * W * Co * 9 * Ci, tiles)


if __name__ == "__main__":
    main()
