#!/usr/bin/env python3
"""Debug aid: stats_out of pcdm_gemm on inputs whose outputs are known exactly (linear, W selects / scales), printing which
(32-row block, wave-column range, group, statistic) entries differ from the expected sums."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
ops.STATS_MIN_HW = 32


def run(tile, M, HW, N, G, pattern):
    K = 64
    a = torch.zeros(M, K)
    w = torch.zeros(N, K)
    if pattern == "ones":
        a[:, 0] = 1
        w[:, 0] = 1
    elif pattern == "rows":       # out[m, n] = (m % 64) / 8  (exact in bf16)
        a[:, 0] = (torch.arange(M) % 64).float() / 8
        w[:, 0] = 1
    else:                          # out[m, n] = (n % 32) / 4
        a[:, 0] = 1
        w[:, 0] = (torch.arange(N) % 32).float() / 4
    pw = ops.pack_linear(w, torch.zeros(N), dev)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    sg = ops.gemm(a.to(BF16).to(dev), pw, out, tile=tile, rows_per_batch=HW, gn_stats=G)
    torch.cuda.synchronize()
    assert isinstance(sg, ops.StatsGemm)
    ref = (a @ w.t())
    assert torch.equal(out.float().cpu(), ref), "output"
    gs, wn = N // G, sg.wn
    TW = pw.Npad // wn
    st = sg.stats[: (M // 32) * TW * G * 2].view(M // 32, TW, G, 2).cpu()
    r = ref.view(M // 32, 32, N)
    bad = []
    for g in range(G):
        for j in range((g * gs) // wn, ((g + 1) * gs - 1) // wn + 1):
            c0, c1 = max(g * gs, j * wn), min((g + 1) * gs, (j + 1) * wn)
            want = torch.stack([r[:, :, c0:c1].sum((1, 2)), (r[:, :, c0:c1] ** 2).sum((1, 2))], -1)
            got = st[:, j, g]
            d = (got - want).abs() > 1e-3 * (want.abs() + 1)
            for blk in d.any(-1).nonzero().flatten().tolist():
                bad.append((blk, j, g, got[blk].tolist(), want[blk].tolist()))
    print(f"tile {tile} M {M} HW {HW} N {N} G {G} {pattern}: {len(bad)} bad entries of {(M // 32) * G}")
    for b in bad[:12]:
        print("   blk %d (row %d) range %d group %d got %s want %s" % (b[0], b[0] * 32, b[1], b[2], b[3], b[4]))
    if bad:
        blks = sorted({b[0] for b in bad}); gsb = sorted({b[2] for b in bad})
        print("   bad blocks:", blks[:40], "... groups:", gsb[:40])


for tile, N, G in ((4, 128, 8), (21, 320, 32), (5, 64, 4)):
    for pattern in ("ones", "rows", "cols"):
        run(tile, 1024, 512, N, G, pattern)
