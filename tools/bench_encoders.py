#!/usr/bin/env python3
"""DINOv2-giant (the stage-2 driver's ``image_encoder_p``, 1.14 B parameters, random weights) on one MI355X: one 224x224 image ->
[1, 257, 1536], as in stage2_batchtest_inpaint_model.py:165-166.  Weight-streaming bound (2.27 GB of bf16 weights per call)."""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import CLIPVisionModelWithProjection, Dinov2Model, ImageProjModel_p  # noqa: E402


def main():
    m = Dinov2Model()
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in m.expected_shapes().items():
        if len(shp) >= 2 and "embeddings.cls" not in k and "position" not in k and "mask_token" not in k:
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) / (shp[1] * (shp[2] * shp[3] if len(shp) == 4 else 1)) ** 0.5
        elif k.endswith("lambda1") or (k.endswith(".weight") and len(shp) == 1):
            sd[k] = torch.ones(shp)
        else:
            sd[k] = torch.randn(shp, generator=g) * 0.02
    m.load_state_dict(sd)
    m.to("cuda")
    proj = ImageProjModel_p(1536, 768, 1024)
    proj.load_state_dict({k: torch.randn(s, generator=g) * 0.02 for k, s in proj.expected_shapes().items()})
    proj.to("cuda")
    x = torch.randn(1, 3, 224, 224, generator=g).cuda()
    t0 = time.time()
    for _ in range(2):
        y = proj(m(x).last_hidden_state)
    torch.cuda.synchronize()
    setup = time.time() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = proj(m(x).last_hidden_state)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    wb = 2 * sum(v.numel() for k, v in sd.items() if "encoder.layer" in k)
    print(json.dumps(dict(metric="dinov2_giant_plus_image_proj_ms", value=ms, weight_GB=wb / 1e9, weight_stream_TBps=wb / (ms * 1e-3) / 1e12,
                          out_shape=list(y.shape), warm_s=setup)))
    # ---- OpenCLIP ViT-H/14 vision tower + projection (image_encoder_g / stage-1 image_encoder), 632 M parameters
    c = CLIPVisionModelWithProjection()
    sd = {}
    for k, shp in c.expected_shapes().items():
        if len(shp) >= 2 and "position" not in k:
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) / (shp[1] * (shp[2] * shp[3] if len(shp) == 4 else 1)) ** 0.5
        elif k.endswith(".weight") and len(shp) == 1:
            sd[k] = torch.ones(shp)
        else:
            sd[k] = torch.randn(shp, generator=g) * 0.02
    c.load_state_dict(sd)
    c.to("cuda")
    t0 = time.time()
    for _ in range(2):
        e = c(x).image_embeds
    torch.cuda.synchronize()
    setup = time.time() - t0
    e0.record()
    for _ in range(10):
        e = c(x).image_embeds
    e1.record()
    e1.synchronize()
    print(json.dumps(dict(metric="clip_vit_h14_image_embeds_ms", value=e0.elapsed_time(e1) / 10, out_shape=list(e.shape), warm_s=setup)))


if __name__ == "__main__":
    main()
