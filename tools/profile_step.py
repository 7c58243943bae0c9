"""Per-shape time breakdown of one eager denoise step (UNet batch 2*N, latent 64x88) on the MI355X.

    python tools/profile_step.py [--batch 4] > gpurun_out/step_breakdown.txt
Groups the HIP-event timed launches of ops.LAUNCH_LOG by (kernel, problem shape).
"""
from __future__ import annotations

import argparse
import sys
from collections import defaultdict
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--flops-json", default="", help="write the algorithmic FLOPs of one step per family (conv3x3 / Linear / attention) here: "
                    "tools/kernel_trace_summary.py --flops turns the rocprofv3 family times into FLOP-weighted TF/s")
    args = ap.parse_args()
    from oracle.pipeline import synth_inputs
    from oracle.unet import UNetConfig, synth_state_dict
    from pcdms_amd import ops
    from pcdms_amd.pipeline import Stage2_InpaintDiffusionPipeline
    from pcdms_amd.schedulers import DDIMScheduler
    from tests.test_unet import _kwargs
    from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel
    dev = torch.device("cuda:0")
    cfg = UNetConfig.tiny() if args.tiny else UNetConfig()
    sd = synth_state_dict(cfg, seed=0)
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(sd)
    m.to(dev)
    pipe = Stage2_InpaintDiffusionPipeline(m, DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                                                            beta_schedule="scaled_linear", clip_sample=False,
                                                            set_alpha_to_one=False, steps_offset=1))
    h, w, N = 64, 88, args.batch
    inp = {k: v.to(dev) for k, v in synth_inputs(cfg, h, w, N).items()}
    pipe(height=h * 8, width=w * 8, num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=2,
         output_type="latent", use_graph=False, **inp)
    st = pipe._st
    st["step"].zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.LAUNCH_LOG = []
    e0.record()
    pipe._step_eager(st)
    e1.record()
    torch.cuda.synchronize()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for name, flops, a, b, info in (e[:5] for e in log):
        k = (name, info)
        agg[k][0] += 1
        agg[k][1] += flops
        agg[k][2] += a.elapsed_time(b) * 1e-3
    tot = sum(v[2] for v in agg.values())
    fam = defaultdict(lambda: [0, 0.0, 0.0])
    for (name, info), (n, fl, t) in agg.items():
        f = "attention" if name.startswith("flash_attn") else ("conv3x3" if name == "gemm_kernel" and info[3] else ("linear" if name == "gemm_kernel" else None))
        if f:
            fam[f][0] += n
            fam[f][1] += fl
            fam[f][2] += t
    if args.flops_json:
        import json
        Path(args.flops_json).write_text(json.dumps({k: {"launches": v[0], "flops": v[1], "event_ms": v[2] * 1e3} for k, v in fam.items()}, indent=1))
    print(f"eager step wall {e0.elapsed_time(e1):.2f} ms; timed launches (GEMM / attention / GroupNorm / LayerNorm) {tot*1e3:.2f} ms in {len(log)} launches")
    print(f"{'kernel':18s} {'shape (M,N,K,conv,tile) / (B,H,Lq,Lk)':46s} {'n':>3s} {'ms':>8s} {'%':>6s} {'TF/s|TB/s':>10s}")
    for (name, info), (n, fl, t) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        print(f"{name:18s} {str(info):46s} {n:3d} {t*1e3:8.3f} {100*t/tot:6.1f} {fl/t/1e12:10.2f}")
    for k, (n, fl, t) in sorted(fam.items()):
        print(f"# family {k:10s}: {n:3d} launches, {fl / 1e12:6.3f} TFLOP, {t * 1e3:7.3f} ms (HIP events) => {fl / t / 1e12:7.1f} TF/s FLOP-weighted")


if __name__ == "__main__":
    main()
