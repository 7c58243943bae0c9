"""Representative GEMM launches of the denoise step for L2 / memory-path counter passes (is the L2 -> LDS fill rate an L2
miss problem, an L2 queueing problem or a CU-side limit?).  Each problem uses a DIFFERENT kernel instantiation or grid so the counter CSV
can be split by (kernel name, grid):

    cd /tmp && rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d <out>/a -- python tools/pmc_fill.py
    cd /tmp && rocprofv3 --kernel-trace --pmc TCC_TAG_STALL_sum TCC_BUSY_avr TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d <out>/b -- python tools/pmc_fill.py
    python tools/pmc_fill_summary.py <out>/a <out>/b
"""
from __future__ import annotations

import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
REP = 4


def lin(M, N, K, tile, split=1, geglu=False, res=False):
    a = torch.randn(M, K, generator=g).to(BF16).to(dev)
    if geglu:
        pw = ops.pack_geglu(torch.randn(2 * N, K, generator=g) / math.sqrt(K), torch.randn(2 * N, generator=g), dev)
    else:
        pw = ops.pack_linear(torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g), dev)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    r = torch.randn(M, N, generator=g).to(BF16).to(dev) if res else None
    for _ in range(REP):
        ops.gemm(a, pw, out, tile=tile, split_k=split, epilogue=ops.EPI_GEGLU if geglu else ops.EPI_STORE, residual=r, res_mod=M if res else 0)


def conv(B, H, W, Ci, Co, tile, split=1):
    x = torch.randn(B, H, W, Ci, generator=g).to(BF16).to(dev)
    pc = ops.pack_conv3x3(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci), torch.randn(Co, generator=g), dev)
    o = torch.empty(B * H * W, Co, dtype=BF16, device=dev)
    for _ in range(REP):
        ops.gemm(x, pc, o, conv=dict(B=B, Hi=H, Wi=W, Ho=H, Wo=W), tile=tile, split_k=split)


lin(45056, 1280, 320, 31, geglu=True)      # rowgemm GEGLU (ff1 level 0): W streamed by every workgroup
lin(45056, 1280, 320, 18, geglu=True)      # the tiled kernel on the same problem
lin(11264, 2560, 640, 18, geglu=True)      # ff1 level 1   (grid differs from the line above)
lin(2816, 5120, 1280, 17, geglu=True)      # ff1 level 2
lin(2816, 1280, 1280, 6, res=True)         # level-2 N = K linear
lin(45056, 320, 320, 5, res=True)          # level-0 N = K linear
conv(8, 64, 88, 320, 320, 21)              # level-0 conv
conv(8, 32, 44, 640, 640, 11)              # level-1 conv (staggered tile)
conv(8, 16, 22, 1280, 1280, 21, 4)         # level-2 conv, split-K 4
conv(8, 8, 11, 1280, 1280, 4, 8)           # level-3 conv, split-K 8
lin(8192, 8192, 8192, 17)                  # the square reference problem
torch.cuda.synchronize()
