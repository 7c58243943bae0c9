"""Where a thin GEMM launch spends its time: per-wave s_memtime stamps at five points of gemm_kernel (debug bit 2).

    python tools/gemm_anatomy.py            (on the GPU box)
stamps: 0 kernel entry, 1 prologue done (offsets computed, first tile(s) issued), 2 first K-tile landed for the whole workgroup,
3 main loop done, 4 epilogue stores issued, 5 stores drained.  Prints medians over waves of the deltas in microseconds and the
span first-entry -> last-drain over the grid."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
BF16 = torch.bfloat16
NW = {21: 8, 18: 8, 3: 8, 13: 4, 4: 4, 26: 8, 17: 8, 1: 8, 6: 8, 7: 4, 5: 4, 11: 8, 2: 4, 8: 4, 10: 4, 22: 8, 23: 8}


def run(M, N, K, tile, residual=True, reps=3, zeros=False):
    a = (torch.zeros if zeros else torch.randn)(M, K, device=dev).to(BF16)
    w = (torch.zeros if zeros else torch.randn)(N, K) / K ** 0.5
    pw = ops.pack_linear(w, torch.randn(N), dev)
    res = torch.randn(M, N, device=dev).to(BF16) if residual else None
    out = torch.empty(M, N, dtype=BF16, device=dev)
    bm, bn = ops.TILE_SHAPES[tile]
    nwg = -(-M // bm) * (pw.Npad // bn)
    ws = torch.zeros(nwg * NW[tile] * 8, dtype=torch.int64, device=dev)
    p = _lib.GemmParams()
    p.a, p.lda, p.c1, p.w = a.data_ptr(), K, K, pw.w.data_ptr()
    p.M, p.N, p.K, p.Npad = M, N, K, pw.Npad
    p.bias = pw.bias.data_ptr()
    p.rows_per_batch = M
    if res is not None:
        p.residual, p.ldr, p.res_mod = res.data_ptr(), N, M
    p.out, p.ldo = out.data_ptr(), N
    p.ws, p.ws_floats = ws.data_ptr(), ws.numel() * 2
    st = torch.cuda.current_stream().cuda_stream
    for flag in (0, 4):
        p.tile = tile | (flag << 8)
        for _ in range(reps):
            assert lib.pcdm_gemm(C.byref(p), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.pcdm_gemm(C.byref(p), st)
        e1.record()
        torch.cuda.synchronize()
        if flag == 0:
            t_plain = e0.elapsed_time(e1) / 10 * 1e3
    t_dbg = e0.elapsed_time(e1) / 10 * 1e3
    s = ws.view(nwg * NW[tile], 8).cpu().double()
    s = s[s[:, 0] > 0]
    # clock: calibrate the cycle counter against the measured kernel time
    span = (s[:, 5].max() - s[:, 0].min()).item()
    d = [(s[:, i + 1] - s[:, i]).median().item() for i in range(5)]
    tot = (s[:, 5] - s[:, 0]).median().item()
    ghz = span / (t_dbg * 1e3)   # cycles per ns if the whole launch were the span (upper bound on the real clock)
    wall = s[:, 7]
    ok = wall > 0
    mhz = ((s[ok, 5] - s[ok, 0]) / wall[ok]).median().item() * 100.0 if ok.any() else float("nan")   # shader cycles per 10 ns tick of the 100 MHz counter
    print(f"[{'zeros' if zeros else 'N(0,1)'}] effective shader clock {mhz:.0f} MHz | "
          f"M{M} N{N} K{K} tile {tile} res={residual}: {t_plain:.1f} us ({2*M*N*K/t_plain/1e6:.0f} TF/s); stamped {t_dbg:.1f} us; grid span {span:.0f} ticks; "
          f"per wave median ticks: prologue {d[0]:.0f} | first tile {d[1]:.0f} | main loop {d[2]:.0f} | epilogue {d[3]:.0f} | drain {d[4]:.0f} | total {tot:.0f} "
          f"(span/launch = {ghz:.2f} ticks/ns); entry spread {(s[:, 0].max() - s[:, 0].min()).item():.0f}")


if __name__ == "__main__":
    import os
    if os.environ.get("PCDM_ANATOMY") == "clock":   # round 6: the clock the MFMA-dense launches run at (N(0,1) against all-zero operands)
        for (M, N, K) in [(45056, 320, 2880), (45056, 320, 5760), (11264, 1280, 5760), (45056, 320, 320), (2816, 1280, 11520)]:
            for tile in (21, 22):
                for z in (False, True):
                    run(M, N, K, tile, False, zeros=z)
        raise SystemExit(0)
    if os.environ.get("PCDM_ANATOMY") == "small":   # the small-M regime (UNet levels 2 / 3: M = 2816 / 704)
        for (M, N, K) in [(2816, 1280, 1280), (2816, 1280, 5120), (704, 1280, 1280)]:
            for tile in (6, 7, 5, 4, 8, 18, 21):
                if (N % ops.TILE_SHAPES[tile][1]) == 0:
                    run(M, N, K, tile, True)
        raise SystemExit(0)
    # thin-K linears (the prologue / epilogue share) and deep-K problems (the per-K-tile cost of the loop: main loop / (K / 64))
    for (M, N, K) in [(45056, 320, 320), (45056, 960, 320), (45056, 320, 1280), (45056, 320, 5760), (11264, 1280, 5760), (11264, 640, 640),
                      (2816, 1280, 1280)]:
        for tile in (21, 17, 4, 5, 18, 6):
            if K > 1280 and tile in (5, 18, 6):
                continue
            if (N % ops.TILE_SHAPES[tile][1]) == 0:
                for res in ((True, False) if K <= 1280 else (False,)):
                    run(M, N, K, tile, res)
