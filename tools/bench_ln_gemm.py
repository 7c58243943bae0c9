#!/usr/bin/env python3
"""LayerNorm -> Linear pairs of UNet levels 1-3 (UNet batch 8, latent 64x88): the LayerNorm launch + the tuned plain GEMM against the
LNF instances of the tiled kernel (LayerNorm folded into the weights, row statistics inside the GEMM: gemm.hip dispatch_tile_ln), each as
a replayed hipGraph of 20 calls.  The bound-first measurement of round 5's LayerNorm fold.

    python tools/bench_ln_gemm.py > gpurun_out/bench_ln_gemm.txt
"""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def timed(fn):
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(20):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            graph.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    return best


def main():
    g = torch.Generator().manual_seed(0)
    # (name, M, K, N (GEGLU: hidden width), epilogue, per forward)
    shapes = [("L1 to_q|k|v", 11264, 640, 1920, ops.EPI_SPLIT_VT, 5), ("L1 attn2.to_q", 5632, 640, 640, ops.EPI_STORE, 5),
              ("L1 GEGLU ff1", 11264, 640, 2560, ops.EPI_GEGLU, 5), ("L2 to_q|k|v", 2816, 1280, 3840, ops.EPI_SPLIT_VT, 5),
              ("L2 attn2.to_q", 1408, 1280, 1280, ops.EPI_STORE, 5), ("L2 GEGLU ff1", 2816, 1280, 5120, ops.EPI_GEGLU, 5),
              ("L3 to_q|k|v", 704, 1280, 3840, ops.EPI_SPLIT_VT, 1), ("L3 attn2.to_q", 352, 1280, 1280, ops.EPI_STORE, 1),
              ("L3 GEGLU ff1", 704, 1280, 5120, ops.EPI_GEGLU, 1)]
    total_two = total_best = total_best2 = 0.0
    for name, M, K, N, epi, nfw in shapes:
        a = torch.randn(M, K, generator=g).to(BF16).to(dev)
        gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
        if epi == ops.EPI_GEGLU:
            w = (torch.rand(2 * N, K, generator=g) * 2 - 1) / math.sqrt(K)
            b = torch.randn(2 * N, generator=g) * 0.1
            pw, pw_ln = ops.pack_geglu(w, b, dev), ops.pack_geglu_ln(w, b, gamma, beta, dev)
        else:
            w = (torch.rand(N, K, generator=g) * 2 - 1) / math.sqrt(K)
            pw, pw_ln = ops.pack_linear(w, None, dev), ops.pack_linear_ln(w, None, gamma, beta, dev)
        kw = {}
        if epi == ops.EPI_SPLIT_VT:
            C = N // 3
            out = torch.empty(M, 2 * C, dtype=BF16, device=dev)
            Bq = 8 if M % 8 == 0 and (M // 8) % 32 == 0 else 1
            kw = dict(rows_per_batch=M // Bq, out2=torch.zeros(Bq, C, M // Bq + 8, dtype=BF16, device=dev), vt_col0=2 * C)
        else:
            out = torch.empty(M, N, dtype=BF16, device=dev)
        ln = (gamma.to(dev), beta.to(dev), 1e-5)
        ln_buf = torch.empty(M, K, dtype=BF16, device=dev)
        t_two = timed(lambda: ops.gemm(a, pw, out, epilogue=epi, ln=ln, ln_buf=ln_buf, pw_ln=None, **kw))      # LayerNorm launch + tuned plain GEMM
        ref = out.float().clone()
        res = {}
        for tile in ops.LN_TILED_TILES:
            try:
                out.zero_()
                ops.gemm(a, pw, out, epilogue=epi, ln=ln, ln_buf=ln_buf, pw_ln=pw_ln, tile=tile, **kw)
            except RuntimeError:
                continue
            torch.cuda.synchronize()
            err = ((out.float() - ref).norm() / ref.norm()).item()
            res[tile] = (timed(lambda: ops.gemm(a, pw, out, epilogue=epi, ln=ln, ln_buf=ln_buf, pw_ln=pw_ln, tile=tile, **kw)), err)
        # mode 2: the row statistics merged from the producer's partials (here: computed by torch, off the clock)
        stats = ops.row_stats_reference(a)
        ops._STATS_VALID[stats.untyped_storage().data_ptr()] = True
        res2 = {}
        for tile in ops.LN_TILED_TILES:
            try:
                out.zero_()
                ops.gemm(a, pw, out, epilogue=epi, ln=ln, ln_buf=ln_buf, pw_ln=pw_ln, tile=tile, row_stats=stats, **kw)
            except RuntimeError:
                continue
            torch.cuda.synchronize()
            err = ((out.float() - ref).norm() / ref.norm()).item()
            res2[tile] = (timed(lambda: ops.gemm(a, pw, out, epilogue=epi, ln=ln, ln_buf=ln_buf, pw_ln=pw_ln, tile=tile, row_stats=stats, **kw)), err)
        bt = min(res, key=lambda t: res[t][0])
        bt2 = min(res2, key=lambda t: res2[t][0])
        total_two += nfw * t_two
        total_best += nfw * min(t_two, res[bt][0])
        total_best2 += nfw * min(t_two, res[bt][0], res2[bt2][0])
        print(f"{name:16s} M {M:6d} K {K:5d} N {N:5d} x{nfw}: LayerNorm + GEMM {t_two:7.2f} us | in-loop statistics: " +
              "  ".join(f"t{t} {v[0]:6.2f}" for t, v in res.items()) + f" | best t{bt} {res[bt][0]:.2f} (rel {res[bt][1]:.1e})\n{'':52s}| producer partials:  " +
              "  ".join(f"t{t} {v[0]:6.2f}" for t, v in res2.items()) + f" | best t{bt2} {res2[bt2][0]:.2f} (rel {res2[bt2][1]:.1e})", flush=True)
    print(f"sum over one forward: LayerNorm launches {total_two:.1f} us -> in-loop statistics where they win {total_best:.1f} us -> with producer partials {total_best2:.1f} us")
    # ---- what the partials cost their producer: the residual linears of levels 1-2 with and without row_stats
    for name, M, C, tiles in [("L1 to_out + res", 11264, 640, (18, 4, 5)), ("L2 to_out + res", 2816, 1280, (6, 7, 18, 5))]:
        x = torch.randn(M, C, generator=g).to(BF16).to(dev)
        res_t = torch.randn(M, C, generator=g).to(BF16).to(dev)
        pw = ops.pack_linear((torch.rand(C, C, generator=g) * 2 - 1) / math.sqrt(C), torch.randn(C, generator=g) * 0.1, dev)
        out = torch.empty(M, C, dtype=BF16, device=dev)
        st = torch.empty(M, C // 32, 2, dtype=torch.float32, device=dev)
        line = []
        for tile in tiles:
            t0 = timed(lambda: ops.gemm(x, pw, out, residual=res_t, res_mod=M, tile=tile))
            t1 = timed(lambda: ops.gemm(x, pw, out, residual=res_t, res_mod=M, tile=tile, row_stats=st))
            assert ops.row_stats_valid(st)
            line.append(f"t{tile} {t0:6.2f} -> {t1:6.2f}")
        print(f"{name:16s} M {M:6d} C {C:5d}: plain -> with partials, us: " + "   ".join(line), flush=True)


if __name__ == "__main__":
    main()
