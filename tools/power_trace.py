#!/usr/bin/env python3
"""Round 6: is the denoise step POWER-bound?  Samples the GPU's socket power and shader clock (hwmon sysfs, every ~2 ms, from a thread)
while the GPU runs, in turn: nothing; the replayed hipGraph of the whole stage-2 denoise step; single kernels back to back (the level-0
3x3 convolution on tiles 21 / 22 with N(0,1) and with all-zero operands, the level-0 self-attention, the level-0 GroupNorm, the GEGLU
projection).  Prints per phase: mean / p95 power against the cap, mean shader clock, and the phase's throughput.

The 176-row tiles (round 6) are 2-8 % faster than the 192-row ones launch for launch and change NOTHING end to end; if the chip sits at
its power cap during the MFMA-dense kernels, work moved from 235 to 256 CUs is paid for in clock.  This tool measures that.

    python tools/power_trace.py > gpurun_out/power_trace.txt       (on the MI355X)
"""
from __future__ import annotations

import glob
import math
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

BF16 = torch.bfloat16


class Sampler:
    def __init__(self):
        self.power_f = self.cap_f = self.clk_f = None
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            for name in ("power1_average", "power1_input"):
                if Path(hw, name).exists():
                    self.power_f = Path(hw, name)
                    break
            if self.power_f:
                self.cap_f = Path(hw, "power1_cap") if Path(hw, "power1_cap").exists() else None
                self.clk_f = Path(hw, "freq1_input") if Path(hw, "freq1_input").exists() else None
                break
        self.samples = []
        self._stop = False
        self._t = None

    def ok(self):
        return self.power_f is not None

    def cap_w(self):
        try:
            return int(self.cap_f.read_text()) / 1e6
        except Exception:
            return None

    def _read(self):
        try:
            p = int(self.power_f.read_text()) / 1e6
        except Exception:
            p = float("nan")
        try:
            c = int(self.clk_f.read_text()) / 1e6 if self.clk_f else float("nan")
        except Exception:
            c = float("nan")
        return p, c

    def start(self):
        self.samples, self._stop = [], False

        def loop():
            while not self._stop:
                self.samples.append((time.perf_counter(), *self._read()))
                time.sleep(0.002)
        self._t = threading.Thread(target=loop, daemon=True)
        self._t.start()

    def stop(self):
        self._stop = True
        self._t.join()
        return self.samples


def smi_snapshot():
    for cmd in (["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], ["amd-smi", "metric", "-p", "-c"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=30)
            return " ".join(cmd) + "\n" + r.stdout[-3000:]
        except Exception as e:   # noqa: BLE001
            last = repr(e)
    return "no SMI tool: " + last


def phase(name, sampler, fn, seconds, work_per_call=0.0, unit=""):
    """Run fn() back to back for ~seconds (enqueue in bursts, keep the queue shallow), sampling power / clock."""
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    per = max(time.perf_counter() - t0, 1e-5)
    burst = max(1, int(0.05 / per))
    n = 0
    # warm the power state for half a second before sampling
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.5:
        for _ in range(burst):
            fn()
        torch.cuda.synchronize()
    if sampler.ok():
        sampler.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(burst):
            fn()
        torch.cuda.synchronize()
        n += burst
    el = time.perf_counter() - t0
    s = sampler.stop() if sampler.ok() else []
    pw = sorted(x[1] for x in s if not math.isnan(x[1]))
    ck = [x[2] for x in s if not math.isnan(x[2])]
    line = f"{name:58s} {el / max(n, 1) * 1e6:10.1f} us/call"
    if work_per_call:
        line += f"  {work_per_call * n / el:9.1f} {unit}"
    if pw:
        line += f"  | power mean {sum(pw) / len(pw):7.1f} W  p95 {pw[int(0.95 * (len(pw) - 1))]:7.1f} W  max {pw[-1]:7.1f} W"
    if ck:
        line += f"  | sclk mean {sum(ck) / len(ck):6.0f} MHz  min {min(ck):6.0f}"
    line += f"  ({len(s)} samples)"
    print(line, flush=True)


def main():
    from oracle.pipeline import synth_inputs
    from oracle.unet import UNetConfig, synth_state_dict
    from pcdms_amd import ops
    from pcdms_amd.pipeline import Stage2_InpaintDiffusionPipeline
    from pcdms_amd.schedulers import DDIMScheduler
    from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel
    from tests.test_unet import _kwargs
    dev = torch.device("cuda:0")
    smp = Sampler()
    print("hwmon:", smp.power_f, "cap", smp.cap_w(), "W", "clk file", smp.clk_f, flush=True)
    print(smi_snapshot(), flush=True)
    if smp.ok():
        smp.start()
        time.sleep(1.0)
        s = smp.stop()
        pw = [x[1] for x in s]
        print(f"idle: power mean {sum(pw) / len(pw):.1f} W over {len(s)} samples; distinct readings {len(set(pw))} (sensor refresh granularity)", flush=True)

    B, h, w = 8, 64, 88
    # ---- single kernels
    for tile in (21, 22):
        for data in ("N(0,1)", "zeros"):
            gen = torch.randn if data == "N(0,1)" else (lambda *s, **k: torch.zeros(*s, **k))
            x = gen(B, h, w, 320, device=dev).to(BF16)
            pw = ops.pack_conv3x3(gen(320, 320, 3, 3) / math.sqrt(2880), torch.randn(320), dev)
            out = torch.empty(B * h * w, 320, dtype=BF16, device=dev)
            rv = torch.randn(B, 320, device=dev)
            fl = 2.0 * B * h * w * 320 * 2880
            phase(f"conv3x3 320->320 @64x88 tile {tile} {data}", smp,
                  lambda: ops.gemm(x, pw, out, conv=dict(B=B, Hi=h, Wi=w, Ho=h, Wo=w), rowvec=rv, rows_per_batch=h * w, tile=tile), 2.0, fl / 1e12, "TF/s")
    for data in ("N(0,1)", "zeros"):
        gen = torch.randn if data == "N(0,1)" else (lambda *s, **k: torch.zeros(*s, **k))
        L, H = h * w, 5
        q = gen(B * L, H * 64, device=dev).to(BF16)
        k = gen(B * L, H * 64, device=dev).to(BF16)
        vt = gen(B, H * 64, L, device=dev).to(BF16)
        o = torch.empty(B * L, H * 64, dtype=BF16, device=dev)
        phase(f"flash attention self N=5632 {data}", smp, lambda: ops.flash_attn(q, k, vt, o, B, H, L, L), 2.0, 4.0 * B * H * L * L * 64 / 1e12, "TF/s")
    xg = torch.randn(B * h * w, 320, device=dev).to(BF16)
    og = torch.empty_like(xg)
    gam, bet = torch.ones(320, device=dev), torch.zeros(320, device=dev)
    gws = ops.groupnorm_ws(B, 320, dev)
    phase("GroupNorm+SiLU level 0 (8 x 5632 x 320)", smp, lambda: ops.groupnorm(xg, None, B, h * w, 32, 1e-5, gam, bet, True, og, gws), 2.0,
          2.0 * xg.numel() * 2 / 1e12, "TB/s")

    # ---- the whole denoise step, replayed hipGraph (the bench's hot loop)
    cfg = UNetConfig()
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(synth_state_dict(cfg, seed=0))
    m.to(dev)
    pipe = Stage2_InpaintDiffusionPipeline(m, DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                                            clip_sample=False, set_alpha_to_one=False, steps_offset=1))
    N = 4
    inp = {k_: v.to(dev) for k_, v in synth_inputs(cfg, h, w, N).items()}

    def call():
        return pipe(height=h * 8, width=w * 8, num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=50, output_type="latent", **inp).latents
    call()
    torch.cuda.synchronize()
    phase("stage-2 sampling call (50 DDIM steps, batch 4, hipGraph)", smp, call, 6.0, N, "images/s")


if __name__ == "__main__":
    main()
