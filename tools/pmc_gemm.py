"""A handful of GEMM launches for rocprofv3 --pmc runs (one counter set per run, kernel-trace only):
   full / no-load / no-MFMA variants of the 8192^3 GEMM on tile 1 and 4, and two real conv shapes."""
from __future__ import annotations

import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
M = K = N = 8192
a = torch.randn(M, K, device=dev).to(BF16)
pw = ops.pack_linear(torch.randn(N, K) / math.sqrt(K), None, dev)
out = torch.empty(M, N, dtype=BF16, device=dev)
for tile in (1, 4):
    for dbg in (0, 1, 2):
        for _ in range(2):
            ops.gemm(a, pw, out, tile=tile | (dbg << 8))
B = 8
for (H, W, Ci, Co, tile) in [(32, 44, 1920, 640, 4), (64, 88, 640, 320, 5)]:
    x = torch.randn(B, H, W, Ci, device=dev).to(BF16)
    pc = ops.pack_conv3x3(torch.randn(Co, Ci, 3, 3) / math.sqrt(9 * Ci), torch.randn(Co), dev)
    o = torch.empty(B * H * W, Co, dtype=BF16, device=dev)
    for _ in range(2):
        ops.gemm(x, pc, o, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), tile=tile)
torch.cuda.synchronize()
