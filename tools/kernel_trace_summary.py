#!/usr/bin/env python3
"""Per-kernel summary of the TIMED region of a rocprofv3 --kernel-trace run of bench.py: the hipGraph replays of the denoise step.

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o bench -- python bench.py --no-cpu-baseline --no-roofline
    python tools/kernel_trace_summary.py <dir> > profiles/r2_kernel_step_summary.txt

`--stats` aggregates the whole process (weight packing, online tuning of the VAE shapes, warm-up); this script keeps only the last
`--steps` denoise steps (delimited by cfg_step_kernel dispatches) and reports, per kernel instantiation, launches per step, average
duration and time per step -- the numbers bench.py's `roofline.avg_launch_us` (HIP events) must agree with."""
from __future__ import annotations

import argparse
import csv
import glob
import re
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:72]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--flops", default="", help="JSON of tools/profile_step.py --flops-json: algorithmic FLOPs of one step per family")
    a = ap.parse_args()
    f = glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "cfg_step_kernel" in r["Kernel_Name"]]
    n = min(a.steps, len(marks) - 1)
    lo, hi = marks[-n - 1] + 1, marks[-1] + 1
    sel = rows[lo:hi]
    agg = defaultdict(lambda: [0, 0])
    for r in sel:
        d = agg[short(r["Kernel_Name"])]
        d[0] += 1
        d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    busy = sum(v[1] for v in agg.values())
    print(f"# last {n} denoise steps of the trace: {len(sel) / n:.1f} dispatches per step, {span / n / 1e6:.3f} ms per step wall "
          f"(first start -> last end; a step that straddles two graph launches includes the host gap), {busy / n / 1e6:.3f} ms kernel-busy")
    print(f"{'kernel':72s} {'per step':>8s} {'avg us':>9s} {'ms/step':>8s} {'%busy':>6s}")
    fam = defaultdict(lambda: [0, 0])
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:72s} {c / n:8.1f} {t / c / 1e3:9.2f} {t / n / 1e6:8.3f} {100 * t / busy:6.1f}")
        key = "gemm_kernel (all instantiations)" if k.startswith(("gemm_kernel", "rowgemm_kernel")) else ("flash_attn" if "flash_attn" in k else None)
        if key:
            fam[key][0] += c
            fam[key][1] += t
    for k, (c, t) in fam.items():
        print(f"{k:72s} {c / n:8.1f} {t / c / 1e3:9.2f} {t / n / 1e6:8.3f} {100 * t / busy:6.1f}")
    # FLOP-weighted TF/s per family: conv3x3 = the CONV instantiations of gemm_kernel (6th template argument), Linear = the
    # other gemm_kernel instantiations + rowgemm_kernel, attention = flash_attn*; FLOPs = the un-padded algorithmic 2 M N K / 4 B H Lq Lk 64
    # of one step (tools/profile_step.py), time = this trace's kernel durations
    if a.flops:
        import json
        fl = json.load(open(a.flops))
        t_f = defaultdict(lambda: [0, 0])
        for k, (c, t) in agg.items():
            if k.startswith("gemm_kernel"):
                args_ = k[k.index("<") + 1:].split(",")
                key = "conv3x3" if len(args_) > 5 and args_[5].strip() == "true" else "linear"
            elif k.startswith("rowgemm_kernel"):
                key = "linear"
            elif "flash_attn" in k:
                key = "attention"
            else:
                continue
            t_f[key][0] += c
            t_f[key][1] += t
        for key in ("conv3x3", "linear", "attention"):
            if key in fl and t_f[key][1]:
                ms = t_f[key][1] / n / 1e6
                print(f"# family {key:10s}: {t_f[key][0] / n:6.1f} launches, {ms:7.3f} ms per step, {fl[key]['flops'] / 1e12:6.3f} TFLOP per step => "
                      f"{fl[key]['flops'] / (ms * 1e-3) / 1e12:7.1f} TF/s FLOP-weighted ({fl[key]['flops'] / (ms * 1e-3) / 2.5e15:.3f} of the 2.5 PF/s roof)")


if __name__ == "__main__":
    main()
