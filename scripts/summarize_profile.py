#!/usr/bin/env python3
"""Turn rocprofv3 output (gpurun_out/prof/{trace,pmc_fetch,pmc_write}) into the small committed
summaries under profiles/:  <tag>_kernel_stats.csv, <tag>_pmc.json and roofline_traffic.json.

HBM bytes from PMC as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE / WRITE_SIZE are KB, collected
in separate passes; on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced
streaming read, so the read side is doubled; WRITE_SIZE is exact for 16-B stores.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

src = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof")
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
out = Path("profiles")
out.mkdir(exist_ok=True)


def short(name):
    m = re.search(r"(\w+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


stats = glob.glob(str(src / "trace" / "*" / "*_kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(out / f"{tag}_kernel_stats.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
    print("wrote", out / f"{tag}_kernel_stats.csv")

pmc = {}
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(str(src / sub / "*" / "*_counter_collection.csv"))
    if not files:
        continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc.setdefault(k, {})[counter + "_KB_avg"] = sum(v) / len(v)
        pmc[k][counter + "_launches"] = len(v)
if pmc:
    for k, d in pmc.items():
        f = d.get("FETCH_SIZE_KB_avg")
        w = d.get("WRITE_SIZE_KB_avg")
        if f is not None and w is not None:
            d["hbm_bytes_per_launch_corrected"] = 2.0 * f * 1024 + w * 1024
    (out / f"{tag}_pmc.json").write_text(json.dumps(pmc, indent=1, sort_keys=True))
    print("wrote", out / f"{tag}_pmc.json")
    def corrected(prefix):
        n = next((n for n in pmc if n.startswith(prefix) and "hbm_bytes_per_launch_corrected" in pmc[n]), None)
        return None if n is None else pmc[n]["hbm_bytes_per_launch_corrected"]

    traffic = {"source": f"profiles/{tag}_pmc.json",
               "nnconv_hbm_bytes_per_launch": corrected("nnconv64_row_kernel"),
               "gemm_per_source_split_kernel_hbm_bytes_per_launch": corrected("gemm_per_source_split_kernel"),
               "gemm_per_source_kernel_hbm_bytes_per_launch": corrected("gemm_per_source_kernel"),
               "note": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 read-side correction), separate --pmc passes"}
    (out / "roofline_traffic.json").write_text(json.dumps(traffic, indent=1))
    print("wrote profiles/roofline_traffic.json")
