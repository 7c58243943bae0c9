#!/usr/bin/env python3
"""HBM-side traffic per kernel family of one denoise step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md §HBM prescribes (gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> doubled; values are KB).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <out>/pmc_fetch -- python tools/profile_step.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <out>/pmc_write -- python tools/profile_step.py
    python tools/pmc_traffic.py <out>/pmc_fetch <out>/pmc_write > profiles/r1_kernel_traffic.json

Only the LAST eager denoise step of profile_step.py is counted (dispatches after the second-to-last cfg_step_kernel).
Infinity-Cache hits are included in these fabric-side counters; WRITE_SIZE is uncalibrated (guide) -- ratios are what to read."""
from __future__ import annotations

import csv
import glob
import json
import sys


def family(name: str) -> str:
    for key, fam in (("gemm_kernel", "gemm_kernel"), ("flash_attn", "flash_attn_kernel"), ("gn_fused", "groupnorm (single pass)"), ("gn_cluster", "groupnorm (cluster single pass)"),
                     ("gn_stats", "groupnorm (stats pass)"), ("gn_apply", "groupnorm (apply pass)"), ("layernorm", "layernorm"),
                     ("splitk_reduce", "splitk_reduce")):
        if key in name:
            return fam
    return "other"


def last_step(path: str, counter: str):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "cfg_step_kernel" in r["Kernel_Name"]]
    lo = marks[-2] + 1 if len(marks) >= 2 else 0
    out = {}
    for r in rows[lo:marks[-1] + 1]:
        fam = family(r["Kernel_Name"])
        d = out.setdefault(fam, [0, 0.0])
        d[0] += 1
        d[1] += float(r["Counter_Value"])
    return out


def main():
    fetch, write = last_step(sys.argv[1], "FETCH_SIZE"), last_step(sys.argv[2], "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate --kernel-trace passes) over tools/profile_step.py; last eager denoise "
                     "step (UNet batch 8, latent 64x88); FETCH_SIZE x2 (gfx950 correction), KB -> bytes; Infinity-Cache hits included",
           "families": {}}
    for fam in sorted(set(fetch) | set(write)):
        n = fetch.get(fam, write.get(fam))[0]
        fb = fetch.get(fam, [0, 0.0])[1] * 2 * 1024
        wb = write.get(fam, [0, 0.0])[1] * 1024
        res["families"][fam] = {"launches": n, "fetch_MB_per_step": round(fb / 1e6, 1), "write_MB_per_step": round(wb / 1e6, 1),
                                "hbm_bytes_per_launch": round((fb + wb) / max(n, 1))}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 3:   # third argument: path of the gemm_kernel summary bench.py quotes as roofline.traffic (stamped with the sources' hash)
        import importlib.util
        from pathlib import Path
        root = Path(__file__).resolve().parent.parent
        spec = importlib.util.spec_from_file_location("bench_mod", root / "bench.py")
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        g = res["families"]["gemm_kernel"]
        n = g["launches"]
        doc = {"source": res["source"], "measured": "tools/pmc_traffic.py", "kernel_src_sha256": bench.kernel_source_hash(),
               "fetch_bytes_per_launch": round(g["fetch_MB_per_step"] * 1e6 / n), "write_bytes_per_launch": round(g["write_MB_per_step"] * 1e6 / n),
               "hbm_bytes_per_launch": g["hbm_bytes_per_launch"], "launches": n,
               "note": "includes Infinity-Cache hits (memory-side fabric counters); MFMA-bound kernel, reported for information"}
        Path(sys.argv[3]).write_text(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
