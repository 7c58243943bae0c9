#!/usr/bin/env python3
"""FETCH_SIZE per flash_attn launch from a rocprofv3 --pmc FETCH_SIZE pass over tools/pmc_attn.py (x2: the gfx950 correction of
MI355X_MICROARCH.md, KB -> bytes; Infinity-Cache hits included): the fabric-side bytes the K / V^T re-reads of an attention launch cost.
usage: python tools/pmc_attn_fetch.py <rocprof output dir>"""
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "flash_attn" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
print(json.dumps({"fetch_MB_per_launch": [round(float(r["Counter_Value"]) * 2 * 1024 / 1e6, 1) for r in rows],
                  "order": "3 x level-0 self-attention (8, 5, 5632, 5632), then 3 x level-1 (8, 10, 1408, 1408)"}))
