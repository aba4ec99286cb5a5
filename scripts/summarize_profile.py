#!/usr/bin/env python3
"""Turn rocprofv3 output (gpurun_out/prof/{trace,pmc_fetch,pmc_write}_m{1,8}, train_trace — written by
scripts/collect_profiles.sh) into the small committed summaries under profiles/:
  <tag>_m<M>_kernel_stats.csv, <tag>_m<M>_bench_under_rocprof.json, <tag>_m<M>_pmc.json,
  <tag>_train_kernel_stats.csv and roofline_traffic.json.

HBM bytes from PMC as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE / WRITE_SIZE are KB, collected
in separate passes; on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced
streaming read, so the read side is doubled; WRITE_SIZE is exact for 16-B stores.

roofline_traffic.json holds one entry per (kernel, atoms, members, conv_mode, gemm_mode): bench.py prints
`traffic` only for the configuration an entry was measured on.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

src = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof")
tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
out = Path("profiles")
out.mkdir(exist_ok=True)


def short(name):
    m = re.search(r"(\w+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def newest(paths):
    # gpurun merges every call's files into gpurun_out/: an earlier collection's run directory can sit next
    # to this one's
    return max(paths, key=lambda p: Path(p).stat().st_mtime)


def kernel_stats(sub, dest):
    stats = glob.glob(str(src / sub / "*" / "*_kernel_stats.csv"))
    if not stats:
        return
    rows = list(csv.DictReader(open(newest(stats))))
    with open(dest, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
    print("wrote", dest)


def trace_avg_us(sub, kernel):
    """average duration (us) of the kernel whose short name starts with `kernel` in that trace's kernel_stats, or None"""
    stats = glob.glob(str(src / sub / "*" / "*_kernel_stats.csv"))
    if not stats:
        return None
    rows = [r for r in csv.DictReader(open(newest(stats))) if short(r["Name"]).startswith(kernel)]
    if not rows:
        return None
    r = max(rows, key=lambda r_: float(r_["TotalDurationNs"]))      # (template variants: the one the run spent its time in)
    return float(r["AverageNs"]) / 1e3


def last_json_line(path):
    try:
        return json.loads(Path(path).read_text().strip().splitlines()[-1])
    except Exception:
        return None


traffic = []
for M in (1, 8):
    kernel_stats(f"trace_m{M}", out / f"{tag}_m{M}_kernel_stats.csv")
    line = last_json_line(src / f"bench_m{M}.json")
    if line is not None:
        (out / f"{tag}_m{M}_bench_under_rocprof.json").write_text(json.dumps(line, indent=1))
    pmc = {}
    cfg = last_json_line(src / f"pmc_fetch_m{M}.json")
    for counter, sub in (("FETCH_SIZE", f"pmc_fetch_m{M}"), ("WRITE_SIZE", f"pmc_write_m{M}")):
        files = glob.glob(str(src / sub / "*" / "*_counter_collection.csv"))
        if not files:
            continue
        agg = defaultdict(list)
        for r in csv.DictReader(open(newest(files))):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc.setdefault(k, {})[counter + "_KB_avg"] = sum(v) / len(v)
            pmc[k][counter + "_launches"] = len(v)
    if not pmc:
        continue
    for k, d in pmc.items():
        f, w = d.get("FETCH_SIZE_KB_avg"), d.get("WRITE_SIZE_KB_avg")
        if f is not None and w is not None:
            d["hbm_bytes_per_launch_corrected"] = 2.0 * f * 1024 + w * 1024
    (out / f"{tag}_m{M}_pmc.json").write_text(json.dumps(pmc, indent=1, sort_keys=True))
    print("wrote", out / f"{tag}_m{M}_pmc.json")
    if cfg is None:
        continue
    c = cfg["config"]
    # both conv formulations run in a bench (timed path + comparison leg at M <= 8)
    for kernel, conv_mode in (("moment_kernel", "factored"), ("nnconv64_row_kernel", "materialized")):
        name = next((n for n in pmc if n.startswith(kernel) and "hbm_bytes_per_launch_corrected" in pmc[n]), None)
        if name is None:
            continue
        per_launch = pmc[name]["hbm_bytes_per_launch_corrected"]
        # the factored conv runs chunk by chunk over the sources: report per conv application
        launches_per_app = 1
        if kernel == "moment_kernel":
            launches_per_app = -(-c.get("members_in_the_profiled_group", c["members_this_rank"]) * c["atoms"] // 512)
        ent = {"kernel": kernel, "atoms": c["atoms"], "members": c.get("members_in_the_profiled_group", c["members_this_rank"]), "conv_mode": conv_mode,
               "gemm_mode": c["edge_mlp_gemm"], "hbm_bytes_per_launch": per_launch * launches_per_app,
               "launches_per_application": launches_per_app, "source": f"profiles/{tag}_m{M}_pmc.json",
               "rocprof_avg_us": trace_avg_us(f"trace_m{M}", kernel), "rocprof_source": f"profiles/{tag}_m{M}_kernel_stats.csv"}
        if kernel == "moment_kernel":      # the whole conv application: K1 + K2 + K3 (S and the partials are its intermediates)
            tot, parts = 0.0, {}
            for kn in ("moment_kernel", "project_kernel", "finish_kernel"):
                # (K2 is project_f16_kernel in gemm_mode split_f16, project_kernel<256> in split_bf16)
                pref = ("project_f16_kernel", "project_kernel") if kn == "project_kernel" else (kn,)
                nm = next((n for pf in pref for n in pmc if n.startswith(pf) and "hbm_bytes_per_launch_corrected" in pmc[n]), None)
                if nm is not None:
                    parts[kn] = pmc[nm]["hbm_bytes_per_launch_corrected"] * launches_per_app
                    tot += parts[kn]
            if len(parts) == 3:
                ent["application_hbm_bytes"], ent["application_hbm_bytes_by_kernel"] = tot, parts
        traffic.append(ent)


# shape C (N = 50,000): the factored conv's K1 (csrc/moment.hip) per conv application (one launch per 512 destinations) and the materialised conv
# kernel on the 2.0M-edge slice (scripts/run_shape_c.py), same counters and corrections
pmc_c = {}
for counter, sub in (("FETCH_SIZE", "pmc_fetch_shape_c"), ("WRITE_SIZE", "pmc_write_shape_c")):
    files = glob.glob(str(src / sub / "*" / "*_counter_collection.csv"))
    if not files:
        continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(newest(files))):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc_c.setdefault(k, {})[counter + "_KB_avg"] = sum(v) / len(v)
        pmc_c[k][counter + "_launches"] = len(v)
if pmc_c:
    for k, d in pmc_c.items():
        f, w = d.get("FETCH_SIZE_KB_avg"), d.get("WRITE_SIZE_KB_avg")
        if f is not None and w is not None:
            d["hbm_bytes_per_launch_corrected"] = 2.0 * f * 1024 + w * 1024
    (out / f"{tag}_shapeC_pmc.json").write_text(json.dumps(pmc_c, indent=1, sort_keys=True))
    print("wrote", out / f"{tag}_shapeC_pmc.json")
    atoms_c = 50000
    name = next((n for n in pmc_c if n.startswith("moment_kernel") and "hbm_bytes_per_launch_corrected" in pmc_c[n]), None)
    if name:
        lpa = -(-atoms_c // 512)
        traffic.append({"kernel": "moment_kernel", "atoms": atoms_c, "members": 1, "conv_mode": "factored",
                        "gemm_mode": "split_f16", "hbm_bytes_per_launch": pmc_c[name]["hbm_bytes_per_launch_corrected"] * lpa,
                        "launches_per_application": lpa, "source": f"profiles/{tag}_shapeC_pmc.json"})
    name = next((n for n in pmc_c if n.startswith("nnconv64_row_kernel") and "hbm_bytes_per_launch_corrected" in pmc_c[n]), None)
    if name:
        traffic.append({"kernel": "nnconv64_row_kernel", "atoms": atoms_c, "members": 1, "conv_mode": "materialized",
                        "gemm_mode": "slice", "hbm_bytes_per_launch": pmc_c[name]["hbm_bytes_per_launch_corrected"],
                        "launches_per_application": 1, "source": f"profiles/{tag}_shapeC_pmc.json",
                        "note": "first rows of the 50k-atom graph holding 2.0M edges (scripts/run_shape_c.py, bench.py cfg5 leg)"})


def mfma_summary(sub, dest, what, keep):
    """GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES per launch -> kernel cycles and matrix-pipe occupancy"""
    files = glob.glob(str(src / sub / "*" / "*_counter_collection.csv"))
    if not files:
        return
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(newest(files))):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ks = {}
    for k, d in agg.items():
        if not any(k.startswith(p_) for p_ in keep) or "GRBM_GUI_ACTIVE" not in d:
            continue
        gui = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"])
        mf = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0])) / max(len(d.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), 1)
        sq = sum(d.get("SQ_BUSY_CYCLES", [0.0])) / max(len(d.get("SQ_BUSY_CYCLES", [])), 1)
        ks[k] = {"launches": len(d["GRBM_GUI_ACTIVE"]), "GRBM_GUI_ACTIVE": gui, "kernel_cycles": gui / 8,
                 "SQ_VALU_MFMA_BUSY_CYCLES": mf, "SQ_BUSY_CYCLES": sq, "mfma_busy_frac": mf / (1024 * gui / 8) if gui else None}
    dest.write_text(json.dumps({
        "note": f"rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- {what}; averages per launch. "
                "GRBM_GUI_ACTIVE is summed over the 8 XCDs: kernel cycles = GUI/8. SQ_VALU_MFMA_BUSY_CYCLES is summed over the "
                "1,024 SIMDs and equals 32 cycles x the number of v_mfma_f32_32x32x16 issued. mfma_busy_frac = MFMA_BUSY / "
                "(1024 * GUI/8).", "kernels": ks}, indent=1))
    print("wrote", dest)


mfma_summary("pmc_mfma_m1", out / f"{tag}_m1_pmc_mfma.json",
             "python3 bench.py --skip-cpu-baseline --skip-ensemble-leg --single-mode --steps 3 --warmup 1 --no-graph "
             "(1 member, N=504, split_f16)", ("gemm_split_f16_kernel", "moment_kernel", "project_f16_kernel", "project_kernel"))
mfma_summary("pmc_mfma_train", out / f"{tag}_train_pmc_mfma.json",
             "python3 scripts/train_synthetic.py --frames 600 (cfg4 batch: 43.7k edges, k=1024, bf16)", ("gemm_pp_kernel",))
kernel_stats("train_trace", out / f"{tag}_train_kernel_stats.csv")
tj = last_json_line(src / "train.json")
if tj is not None:
    (out / f"{tag}_train_under_rocprof.json").write_text(json.dumps(tj, indent=1))
if traffic:
    (out / "roofline_traffic.json").write_text(json.dumps(
        {"note": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 read-side correction), separate --pmc passes; bytes per "
                 "conv application (all chunk launches of K1, csrc/moment.hip); rocprof_avg_us = the kernel's average in the same "
                 "collection's kernel trace", "configs": traffic}, indent=1))
    print("wrote profiles/roofline_traffic.json")
for leg, name in (("train_fp32", "train_fp32"), ("shape_a", "shapeA_m1"), ("shape_c", "shapeC")):
    if (src / f"{leg}_trace").exists():
        kernel_stats(f"{leg}_trace", out / f"{tag}_{name}_kernel_stats.csv")
        text = (src / f"{leg}.json").read_text() if (src / f"{leg}.json").exists() else ""
        if text.strip():
            (out / f"{tag}_{name}_under_rocprof.json").write_text(text)
