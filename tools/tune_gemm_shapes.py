"""Measure the best (tile configuration, split-K) of gemm.hip for every GEMM / conv problem shape of the stage-2
UNet on this MI355X and write pcdms_amd/tuning/gfx950.json (committed; shapes outside it are tuned online).

    python tools/tune_gemm_shapes.py [--out gpurun_out/gfx950.json]
Shapes covered: latent 64x88 (352x512 pairs) at UNet batch 8 and 16 (configs[1], configs[2]) and latent 32x64
(256x256 pairs, configs[0]) at UNet batch 2 and 8.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--cold", action="store_true", help="evict L2 / Infinity Cache before every timed launch (weights are never cached "
                    "inside the real step)")
    ap.add_argument("--only-main", action="store_true", help="only the bench configuration (UNet batch 8, latent 64x88)")
    ap.add_argument("--merge", action="store_true", help="write the committed table updated with this run's entries (prints what changed)")
    args = ap.parse_args()
    from oracle.unet import UNetConfig, synth_state_dict
    from pcdms_amd import ops
    from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel
    from tests.test_unet import _inputs, _kwargs
    ops._TUNED.clear()
    ops.TUNE_ITERS, ops.TUNE_REPEATS, ops.TUNE_COLD = 5, 3, args.cold
    dev = torch.device("cuda:0")
    cfg = UNetConfig()
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(synth_state_dict(cfg, seed=0))
    m.to(dev)
    for (B, h, w) in [(8, 64, 88)] if args.only_main else [(8, 64, 88), (16, 64, 88), (2, 32, 64), (8, 32, 64)]:
        s, e, c, p = _inputs(cfg, B, h, w, 258)
        m(s.to(dev), torch.tensor(500, device=dev), e.to(dev), class_labels=c.to(dev), my_pose_cond=p.to(dev))
        torch.cuda.synchronize()
        # the CFG-shared prefix of the fused sampler: conv_in and the first conv1 as half-batch launches with two epilogues (dup_rows)
        s2 = torch.cat([s[: B // 2]] * 2).to(dev)
        cond = m.prepare_conditioning(B, h, w, e.to(dev), c.to(dev), p.to(dev), zero_ctx_batches=B // 2, shared_cfg_input=True)
        x_in = ops.nchw_to_nhwc_bf16(s2, cpad=m._w["conv_in"].cin)
        m._forward_nhwc(x_in, B, h, w, torch.tensor([500], device=dev), cond)
        torch.cuda.synchronize()
        print(f"B={B} latent {h}x{w}: {len(ops._TUNED)} shapes tuned", flush=True)
    out = Path(args.out) if args.out else ops.TUNING_FILE
    if args.merge:   # keep the committed entries of the shapes this run did not visit
        mine = dict(ops._TUNED)
        ops._TUNED.clear()
        ops.load_tuning()
        changed = {k: (ops._TUNED.get(k), v) for k, v in mine.items() if ops._TUNED.get(k) != v}
        ops._TUNED.update(mine)
        for k, (a, b) in sorted(changed.items(), key=lambda kv: str(kv[0])):
            print("  ", ",".join(str(x) for x in k), a, "->", b)
    ops.save_tuning(out, note=f"MI355X gfx950, torch {torch.__version__}, tools/tune_gemm_shapes.py, best of 3 x 5 launches"
                                + (", caches evicted before every timed launch" if args.cold else ""))
    print("wrote", out)


if __name__ == "__main__":
    main()
