#!/usr/bin/env python3
"""Re-tune single entries of the committed GEMM tuning table on the GPU: drop the keys that start with one of the given prefixes, run one
full-size UNet forward (the online tuner measures every candidate tile for the dropped shapes), print old -> new.

    python tools/retune_keys.py 45056,320,576 45056,64,2880      (on the MI355X; prints, does not write)
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    from oracle.unet import UNetConfig, synth_state_dict
    from pcdms_amd import ops
    from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel
    from tests.test_unet import _inputs, _kwargs
    write = "--write" in sys.argv           # put the re-tuned entries into the committed table
    prefixes = [tuple(int(x) if x.lstrip("-").isdigit() else x for x in a.split(",")) for a in sys.argv[1:] if a != "--write"]   # ("ln,11264,5120": the folded-LayerNorm keys)
    old = {k: v for k, v in ops._TUNED.items() if any(k[: len(p)] == p for p in prefixes)}
    for k in old:
        del ops._TUNED[k]
    ops.TUNE_ITERS, ops.TUNE_REPEATS = 5, 3
    dev = torch.device("cuda:0")
    cfg = UNetConfig()
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(synth_state_dict(cfg, seed=0))
    m.to(dev)
    s, e, c, p = _inputs(cfg, 8, 64, 88, 258)
    m(s.to(dev), torch.tensor(500, device=dev), e.to(dev), class_labels=c.to(dev), my_pose_cond=p.to(dev))
    torch.cuda.synchronize()
    for k, v in old.items():
        print(",".join(str(x) for x in k), "old", v, "new", ops._TUNED.get(k))
    if write:
        import json
        tab = json.loads(Path(ops.TUNING_FILE).read_text())
        for k, v in old.items():
            nv = ops._TUNED.get(k)
            if nv is not None and tuple(nv) != tuple(v):
                tab["gemm"][",".join(str(x) for x in k)] = list(nv)
        Path(ops.TUNING_FILE).write_text(json.dumps(tab, indent=0))
        print("wrote", ops.TUNING_FILE)


if __name__ == "__main__":
    main()
