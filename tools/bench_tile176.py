#!/usr/bin/env python3
"""Round 6: the 176-row block tiles (22 = 176 x 320, 23 = 176 x 256) against the committed choice (mostly tile 21 / 26: 192 rows) on every
problem shape of the stage-2 denoise step that runs on a full-row tile today, plus the level-1..3 convolutions where M = 64 / 16 / 4 x 176:
each (tile, split-K) timed as a replayed hipGraph of 10 launches, N(0,1) operands, with the epilogue variant the step uses.

    python tools/bench_tile176.py [--write]     # --write: put the winners into pcdms_amd/tuning/gfx950.json
"""
from __future__ import annotations

import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import ops  # noqa: E402
from tools.bench_conv import timed  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")


def key_fields(k):
    f = k.split(",")
    return dict(M=int(f[0]), Npad=int(f[1]), K=int(f[2]), conv=int(f[3]), stride=int(f[4]), ups=int(f[5]), epi=int(f[6]), two=f[7] == "True",
                res=f[8] == "True", flag=(f[9] if len(f) > 9 else None))


def main():
    import json
    write = "--write" in sys.argv
    cand_tiles = (22, 23)
    for i, a_ in enumerate(sys.argv):   # --tiles 24,25,29,30: the 16x16x32-fragment twins of the 32x32x16 tiles (same tool, other candidates)
        if a_ == "--tiles":
            cand_tiles = tuple(int(x) for x in sys.argv[i + 1].split(","))
    uneven = set(cand_tiles) <= {22, 23}
    tab_path = ops.TUNING_FILE
    tab = json.loads(Path(tab_path).read_text())
    gem = tab["gemm"]
    B = 8
    total_old = total_new = 0.0
    changed = {}
    for k, v in sorted(gem.items()):
        if k.startswith("ln,"):
            continue
        f = key_fields(k)
        M, Npad, K = f["M"], f["Npad"], f["K"]
        if (uneven and M % 176) or M not in (45056, 22528, 11264, 5632, 2816, 1408, 704) or f["epi"] not in (0, 1):
            continue
        cands = []
        for t in cand_tiles:
            bm, bn = ops.TILE_SHAPES[t]
            if Npad % bn:
                continue
            if f["epi"] == 1 and t in (22,):
                continue
            ntiles = -(-M // bm) * (Npad // bn)
            for sk in (1, 2, 3, 4, 6, 8, 12, 16):
                if sk > 1 and (f["epi"] != 0 or ntiles * sk > 512 or (K // 64) // sk < 4):
                    continue
                if ntiles * sk < 100:
                    continue
                cands.append((t, sk))
        if not cands or v[0] in (31, 32, 33, 34, 35, 36):
            continue
        # operands
        N = Npad
        if f["conv"]:
            cin = K // 9
            hw = {45056: (64, 88), 11264: (32, 44), 2816: (16, 22), 704: (8, 11), 22528: (64, 88), 5632: (32, 44), 1408: (16, 22)}[M]
            Bc = M // (hw[0] * hw[1])
            ho, wo = hw
            if f["stride"] == 2:
                hi, wi = 2 * ho, 2 * wo
            elif f["ups"]:
                hi, wi = ho // 2, wo // 2
            else:
                hi, wi = ho, wo
            x = torch.randn(Bc, hi, wi, cin, device=dev).to(BF16)
            pw = ops.pack_conv3x3(torch.randn(N, cin, 3, 3) / math.sqrt(9 * cin), torch.randn(N), dev)
            kw = dict(conv=dict(B=Bc, Hi=hi, Wi=wi, Ho=ho, Wo=wo, stride=f["stride"] or 1, upsample=f["ups"]), rows_per_batch=ho * wo)
            a = x
            if f["res"]:
                kw.update(residual=torch.randn(M, N, device=dev).to(BF16), res_mod=M)
            else:
                kw["rowvec"] = torch.randn(Bc, N, device=dev)
        else:
            if f["epi"] == 1:
                Nout = Npad // 2
                pw = ops.pack_geglu(torch.randn(2 * Nout, K) / math.sqrt(K), torch.randn(2 * Nout), dev)
                kw = dict(epilogue=ops.EPI_GEGLU)
            else:
                Nout = N
                pw = ops.pack_linear(torch.randn(N, K) / math.sqrt(K), torch.randn(N), dev)
                kw = {}
            if f["two"]:
                a = torch.randn(M, K // 2 if (K // 2) % 64 == 0 else 320, device=dev).to(BF16)
                kw["a2"] = torch.randn(M, K - a.shape[1], device=dev).to(BF16)
            else:
                a = torch.randn(M, K, device=dev).to(BF16)
            if f["res"]:
                kw.update(residual=torch.randn(M, Nout, device=dev).to(BF16), res_mod=M)
        out = torch.empty(M, pw.N, dtype=BF16, device=dev)
        if f["flag"] not in (None, "False", "0"):
            continue   # (zero_rows / dup_rows variants keep their tile: same kernel, see the plain key)

        def run(tile, sk):
            return ops.gemm(a, pw, out, tile=tile, split_k=sk, **kw)

        try:
            t_old = min(timed(lambda: run(v[0], v[1])) for _ in range(2))
        except RuntimeError as e:
            print(f"{k}: committed {v} failed: {e}")
            continue
        best, t_best = tuple(v), t_old
        row = []
        for (t, sk) in cands:
            try:
                us = min(timed(lambda: run(t, sk)) for _ in range(2))
            except RuntimeError:
                continue
            row.append(f"{t}/{sk}:{us:.1f}")
            if us < t_best:
                best, t_best = (t, sk), us
        fl = 2.0 * M * pw.N * K * (2 if f["epi"] == 1 else 1)
        print(f"{k:46s} committed {v[0]}/{v[1]} {t_old:7.1f} us ({fl / t_old / 1e6:6.0f} TF/s) | best {best[0]}/{best[1]} {t_best:7.1f} us "
              f"({fl / t_best / 1e6:6.0f})  {100 * (t_best / t_old - 1):+5.1f} %   [{' '.join(row)}]", flush=True)
        total_old += t_old
        total_new += t_best
        if best != tuple(v) and t_best < 0.985 * t_old:
            changed[k] = list(best)
    print(f"sum over the listed shapes (one launch each): {total_old:.1f} -> {total_new:.1f} us; {len(changed)} entries would change")
    if write and changed:
        gem.update(changed)
        # the zero_rows / dup_rows variants of a changed key follow it (same kernel instance, same grid)
        for k in list(gem):
            f = k.split(",")
            if len(f) > 9 and ",".join(f[:9]) in changed:
                gem[k] = changed[",".join(f[:9])]
        tab["note"] = (tab.get("note", "") + f" | round 6: tiles {cand_tiles} where tools/bench_tile176.py measured them >= 1.5 % faster").strip()
        Path(tab_path).write_text(json.dumps(tab, indent=0))
        print(f"wrote {tab_path}")
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/r6_tile_changes_" + "_".join(map(str, cand_tiles)) + ".json").write_text(json.dumps(changed, indent=0))


if __name__ == "__main__":
    main()
