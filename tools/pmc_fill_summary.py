"""Per (kernel instantiation, grid) means of the counters collected by tools/pmc_fill.py passes + derived L2 hit rate and request rates."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::|pcdm_gemm_detail::|void ", "", r["Kernel_Name"])
            if "gemm_kernel" not in name:
                continue
            key = name.split("(")[0][:70] + " grid=" + r.get("Grid_Size", "?")
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r:
                acc[key]["duration_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = {}
for k, cs in acc.items():
    m = {c: sum(v[1:]) / max(len(v) - 1, 1) for c, v in cs.items()}     # (first launch of each problem dropped: cold)
    row = {c: round(v, 1) for c, v in sorted(m.items())}
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        row["l2_hit_rate"] = round(m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1), 4)
    if "TCC_REQ_sum" in m and m.get("duration_us"):
        row["l2_req_per_us"] = round(m["TCC_REQ_sum"] / m["duration_us"], 1)
        row["l2_GBps_at_128B_per_req"] = round(m["TCC_REQ_sum"] * 128 / m["duration_us"] / 1e3, 1)
    if "TCC_EA0_RDREQ_sum" in m and m.get("duration_us"):
        row["fabric_read_GBps_at_64B_per_req(x2 if 128B)"] = round(m["TCC_EA0_RDREQ_sum"] * 64 / m["duration_us"] / 1e3, 1)
    out[k] = row
print(json.dumps(out, indent=1))
