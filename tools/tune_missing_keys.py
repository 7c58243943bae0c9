#!/usr/bin/env python3
"""Tune the shapes the committed table does not have yet (new kernels add keys: round 5's LayerNorm-folded tiled instances add the
"ln,M,Npad,K,epilogue" keys of UNet levels 1-3): load the committed table, run the forwards tools/tune_gemm_shapes.py runs (the online
tuner measures only the missing keys), print the new entries and write the merged table.

    python tools/tune_missing_keys.py [--out gpurun_out/gfx950_merged.json] [--all-configs]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/gfx950_merged.json")
    ap.add_argument("--all-configs", action="store_true", help="UNet batch 16 and the 32x64 latent too (default: the bench configuration only)")
    args = ap.parse_args()
    from oracle.unet import UNetConfig, synth_state_dict
    from pcdms_amd import ops
    from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel
    from tests.test_unet import _inputs, _kwargs
    before = dict(ops._TUNED)
    ops.TUNE_ITERS, ops.TUNE_REPEATS = 5, 3
    dev = torch.device("cuda:0")
    cfg = UNetConfig()
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(synth_state_dict(cfg, seed=0))
    m.to(dev)
    for (B, h, w) in [(8, 64, 88), (16, 64, 88), (2, 32, 64), (8, 32, 64)] if args.all_configs else [(8, 64, 88)]:
        s, e, c, p = _inputs(cfg, B, h, w, 258)
        m(s.to(dev), torch.tensor(500, device=dev), e.to(dev), class_labels=c.to(dev), my_pose_cond=p.to(dev))
        cond = m.prepare_conditioning(B, h, w, e.to(dev), c.to(dev), p.to(dev), zero_ctx_batches=B // 2, shared_cfg_input=True)
        x_in = ops.nchw_to_nhwc_bf16(torch.cat([s[: B // 2]] * 2).to(dev), cpad=m._w["conv_in"].cin)
        m._forward_nhwc(x_in, B, h, w, torch.tensor([500], device=dev), cond)
        torch.cuda.synchronize()
    new = {k: v for k, v in ops._TUNED.items() if before.get(k) != v}
    for k, v in sorted(new.items(), key=lambda kv: str(kv[0])):
        print("new", ",".join(str(x) for x in k), "->", v)
    ops.save_tuning(Path(args.out), note=f"MI355X gfx950, torch {torch.__version__}: committed table + tools/tune_missing_keys.py")
    print("wrote", args.out, f"({len(new)} new entries)")


if __name__ == "__main__":
    main()
