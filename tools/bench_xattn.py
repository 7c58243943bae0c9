#!/usr/bin/env python3
"""The cross-attention of a BasicTransformerBlock (UNet batch 8 with the zero-context half skipped: 4 batch entries, 258 context tokens) as the
launches it was -- LayerNorm-folded to_q GEMM (level 0: rowgemm; levels 1-2: LayerNorm launch + GEMM or a folded tiled instance, whatever the
table says) + pcdm_flash_attn -- against ONE pcdm_flash_attn_qproj launch (round 5), each as a replayed hipGraph of 20 calls.

    python tools/bench_xattn.py > gpurun_out/bench_xattn.txt
"""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402
from tools.bench_ln_gemm import timed  # noqa: E402

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def main():
    g = torch.Generator().manual_seed(0)
    tot_a = tot_b = 0.0
    for name, B, H, Lq, nfw in [("level 0", 4, 5, 5632, 5), ("level 1", 4, 10, 1408, 5), ("level 2", 4, 20, 352, 5), ("level 3", 4, 20, 88, 1)]:
        C, Lk, M = H * 64, 258, B * Lq
        x = torch.randn(M, C, generator=g).to(BF16).to(dev)
        wq = (torch.rand(C, C, generator=g) * 2 - 1) / math.sqrt(C)
        gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
        pw, pw_ln = ops.pack_linear(wq, None, dev), ops.pack_linear_ln(wq, None, gamma, beta, dev)
        k = torch.randn(B * Lk, C, generator=g).to(BF16).to(dev)
        vt = torch.randn(B, C, 264, generator=g).to(BF16).to(dev)
        q2 = torch.empty(M, C, dtype=BF16, device=dev)
        ln_buf = torch.empty(M, C, dtype=BF16, device=dev)
        o1, o2 = torch.empty(M, C, dtype=BF16, device=dev), torch.empty(M, C, dtype=BF16, device=dev)
        ln = (gamma.to(dev), beta.to(dev), 1e-5)

        def unfused():
            ops.gemm(x, pw, q2, ln=ln, ln_buf=ln_buf, pw_ln=pw_ln)
            ops.flash_attn(q2, k, vt, o1, B, H, Lq, Lk)
        unfused()   # (tunes the LayerNorm -> to_q pair of this shape if the table lacks it)
        t_q = timed(lambda: ops.gemm(x, pw, q2, ln=ln, ln_buf=ln_buf, pw_ln=pw_ln))
        t_a = timed(lambda: ops.flash_attn(q2, k, vt, o1, B, H, Lq, Lk))
        t_u = timed(unfused)
        t_f = timed(lambda: ops.flash_attn_qproj(x, pw_ln, k, vt, o2, B, H, Lq, Lk))
        torch.cuda.synchronize()
        rel = ((o2.float() - o1.float()).norm() / o1.float().norm()).item()
        tot_a += nfw * t_u
        tot_b += nfw * t_f
        print(f"{name}: (B {B}, H {H}, Lq {Lq}, Lk {Lk}) x{nfw}: LayerNorm -> to_q {t_q:6.2f} us + attention {t_a:6.2f} us = {t_u:6.2f} us as a pair | "
              f"one launch {t_f:6.2f} us | fused vs unfused rel-L2 {rel:.1e}", flush=True)
    print(f"sum over one forward: {tot_a:.1f} -> {tot_b:.1f} us")


if __name__ == "__main__":
    main()
