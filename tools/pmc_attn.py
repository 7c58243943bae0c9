"""Level-0 / level-1 self-attention launches for rocprofv3 --pmc runs."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = 8
for (H, Lq, Lk) in [(5, 5632, 5632), (10, 1408, 1408)]:
    C = H * 64
    q = torch.randn(B * Lq, C, device=dev).bfloat16()
    k = torch.randn(B * Lk, C, device=dev).bfloat16()
    vt = torch.randn(B, C, Lk, device=dev).bfloat16()
    o = torch.empty(B * Lq, C, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.flash_attn(q, k, vt, o, B, H, Lq, Lk)
torch.cuda.synchronize()
