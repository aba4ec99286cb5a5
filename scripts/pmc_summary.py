"""Per-kernel averages of the counters collected by scripts/pmc_passes.sh (gpurun_out/pmc/pass*)."""
import csv, glob, re, sys
from collections import defaultdict

want = sys.argv[1] if len(sys.argv) > 1 else ""
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmc/pass*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+)(<[^(]*>)?\(", r["Kernel_Name"])
        name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:50]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if want and want not in k:
        continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:32s} avg {sum(v)/len(v):16.1f}  n={len(v)}")
