"""Where rowgemm.hip (the A-in-registers K = 320 GEMM) spends its cycles: per-wave cycle accounting (debug bit 2 of the tile field):
prologue (ring start + A rows + LayerNorm) | waiting at the stage barrier | DMA issue + fragment reads + MFMAs | epilogue | loop overhead.

    python tools/rowgemm_anatomy.py          (on the MI355X)"""
import ctypes as C
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
BF16 = torch.bfloat16


def run(name, M, N, tile, epi, ln, extra=0):
    K = 320
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g).to(BF16).to(dev)
    if epi == ops.EPI_GEGLU:
        pw = ops._with_wsum(ops.pack_geglu(torch.randn(2 * N, K, generator=g) / math.sqrt(K), torch.randn(2 * N, generator=g) * 0.5, dev))
        flops = 2.0 * M * 2 * N * K
    else:
        pw = ops._with_wsum(ops.pack_linear(torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g), dev))
        flops = 2.0 * M * N * K
    bm = ops.TILE_SHAPES.get(tile, (192, 128))[0]
    nwg = -(-M // bm)
    ws = torch.zeros(max(nwg, (M + 95) // 96) * 8 * 8, dtype=torch.int64, device=dev)
    p = _lib.GemmParams()
    p.a, p.lda, p.c1, p.w = a.data_ptr(), K, K, pw.w.data_ptr()
    p.M, p.N, p.K, p.Npad = M, pw.N, K, pw.Npad
    p.bias = pw.bias.data_ptr()
    p.rows_per_batch = 5632
    p.epilogue = epi
    if epi == ops.EPI_SPLIT_VT:
        Cc = N // 3
        out = torch.empty(M, 2 * Cc, dtype=BF16, device=dev)
        vt = torch.zeros(M // 5632, Cc, 5632, dtype=BF16, device=dev)
        p.out2, p.ldo2, p.vt_col0 = vt.data_ptr(), 5632, 2 * Cc
        p.ldo = 2 * Cc
    else:
        out = torch.empty(M, N, dtype=BF16, device=dev)
        p.ldo = N
    p.out = out.data_ptr()
    if ln:
        p.ln_wsum, p.ln_eps = pw.wsum.data_ptr(), 1e-5
    p.ws, p.ws_floats = ws.data_ptr(), ws.numel() * 2
    st = torch.cuda.current_stream().cuda_stream
    times = {}
    for flag in (0, 4):
        p.tile = tile | ((flag | extra) << 8)
        for _ in range(3):
            rc = lib.pcdm_gemm(C.byref(p), st)
            assert rc == 0, rc
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.pcdm_gemm(C.byref(p), st)
        e1.record()
        torch.cuda.synchronize()
        times[flag] = e0.elapsed_time(e1) / 10 * 1e3
    s = ws.view(-1, 8)[: nwg * 8].cpu().double()
    s = s[s[:, 2] > 0]
    med = [s[:, i].median().item() for i in range(6)]
    tot = sum(med)
    print(f"{name} M{M} N{N} tile {tile} ln={ln} dbg={extra}: {times[0]:.1f} us ({flops / times[0] / 1e6:.0f} TF/s); stamped {times[4]:.1f} us; per wave median cycles: "
          f"prologue {med[0]:.0f} | vmcnt wait {med[1]:.0f} | barrier {med[5]:.0f} | DMA issue + loop {med[4]:.0f} | reads+MFMA {med[2]:.0f} | epilogue {med[3]:.0f} | "
          f"sum {tot:.0f} (= {tot / times[4] / 1e3:.2f} cycles/ns)")


if __name__ == "__main__":
    if "--dev34" in sys.argv:   # experiments on the four-wave tile (a PCDM_DEV_ROWGEMM_VARIANTS build)
        for tile, extra, nm in ((34, 0, "5 stages"), (34, 128, "5 stages, NO vmcnt wait (invalid results)"), (43, 0, "4 stages"), (42, 0, "3 stages"),
                                (44, 0, "5 stages, 3-way N split"), (34, 32, "5 stages, stores dropped"), (34, 16, "5 stages, no N-tile rotation")):
            run(f"ff1 {nm}", 45056, 1280, tile, ops.EPI_GEGLU, True, extra)
        raise SystemExit(0)
    run("ff1 (4 waves, 2 WGs/CU)", 45056, 1280, 34, ops.EPI_GEGLU, True)
    run("qkv (4 waves, 2 WGs/CU)", 45056, 960, 34, ops.EPI_SPLIT_VT, True)
    run("lin (4 waves, 2 WGs/CU)", 45056, 320, 34, ops.EPI_STORE, False)
    run("ff1 ", 45056, 1280, 31, ops.EPI_GEGLU, True)
    run("ff1 dma-both-first", 45056, 1280, 31, ops.EPI_GEGLU, True, 64)
    run("lin dma-both-first", 45056, 2560, 31, ops.EPI_STORE, False, 64)
    run("ff1 ", 45056, 1280, 31, ops.EPI_GEGLU, False)
    run("qkv ", 45056, 960, 32, ops.EPI_SPLIT_VT, True)
    run("q2  ", 22528, 320, 32, ops.EPI_STORE, True)
    run("lin ", 45056, 2560, 31, ops.EPI_STORE, False)
    run("lin ", 45056, 320, 32, ops.EPI_STORE, False)
