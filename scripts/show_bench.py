"""Print the headline and the per-kernel breakdown of a bench.py JSON line (file argument)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{d['value']:.1f} {d['unit']}  {d['ms_per_step']:.4f} ms/step  n_gpus={d['n_gpus']}")
for k, v in (d.get("kernels") or {}).items():
    print(f"  {k:18s} {v['ms_per_step']:.4f} ms/step  avg {1e3 * v['avg_ms']:.1f} us x {v['launches'] // max(d['steps'], 1)}")
r = d.get("roofline") or {}
print("  roofline:", {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "frac", "traffic")})
