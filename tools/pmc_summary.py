#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (counter_collection CSVs) into one table per kernel family: per-dispatch means of every counter and
the ratios that say where the wave cycles go (MI355X_MICROARCH.md "rocprofv3 PMC slots": WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY
~ WAVE_CYCLES, in quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles = 32 x MFMAs for 32x32x16, 16 x for 16x16x32).

    python tools/pmc_summary.py <dir-of-pass-a> <dir-of-pass-b> ... [--match flash_attn] > profiles/r2_pmc_*.json
MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) / (dispatch duration x 1024 SIMDs x clock); the
duration comes from the dispatch's own Start / End timestamps in the counter CSV (serialised dispatches), the clock is taken as the
2.4 GHz maximum, so the figure is a LOWER bound (under MFMA load the chip runs at 1.9-2.1 GHz: divide by ~0.85)."""
from __future__ import annotations

import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"pcdm_gemm_detail::GemmArgs", "GemmArgs", name)
    return re.sub(r"^void ", "", name)[:90]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
    if match:
        args = [a for a in args if a != match]
    acc = defaultdict(lambda: defaultdict(list))
    for d in args:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if match and match not in r["Kernel_Name"]:
                    continue
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if "Start_Timestamp" in r and "End_Timestamp" in r and r["Counter_Name"] == "SQ_WAVE_CYCLES":
                    acc[short(r["Kernel_Name"])]["duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {}
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        row = {"dispatches": max(len(v) for v in cs.values()), **{c: round(v, 1) for c, v in sorted(m.items())}}
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c, lbl in (("SQ_WAIT_ANY", "frac_wave_cycles_waiting (s_waitcnt / barrier)"), ("SQ_WAIT_INST_ANY", "frac_wave_cycles_issue_stalled"),
                           ("SQ_ACTIVE_INST_ANY", "frac_wave_cycles_issuing"), ("SQ_WAIT_INST_LDS", "frac_wave_cycles_lds_issue_stalled")):
                if c in m:
                    row[lbl] = round(m[c] / wc, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("duration_ns"):
            row["mfma_busy_frac (>=, at 2.4 GHz)"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["duration_ns"] * 2.4 * 1024), 3)
        if "SQ_INSTS_MFMA" in m and m["SQ_INSTS_MFMA"]:
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
                if c in m:
                    row[c.replace("SQ_INSTS_", "").lower() + "_per_mfma"] = round(m[c] / m["SQ_INSTS_MFMA"], 2)
        out[k] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
