#!/usr/bin/env python3
"""Headline benchmark: rolled-out MD frames/sec (BASELINE.json), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md §8d shape B): BBA all-atom stand-in, N=504 atoms
uniform at 0.1 atoms/A^3, 8 A cutoff (E ~ 60k directed edges incl. self-loops), window 10, the
reference's CLI-default model (width 64, kernel_width 1024, depth 6 -> 12 conv applications), fp32,
free-running autoregressive rollout entirely on the device.  One "step" = one new frame for every
member on every rank (graph rebuild + forward + window slide).  Weak scaling: every rank runs
`--members-per-gpu` independent trajectories (ensemble members, different perturbations of the
start window); no collective while stepping, one RCCL all-gather of the produced frames at the end,
inside the timed region.  value = frames produced by all ranks / max-over-ranks wall time.

Weights: synthetic near-identity set (weights.py) — no trained checkpoint exists offline and
random-init weights collapse the cloud to one point, which would change E (the cost driver) after
one step.  The architecture, arithmetic and update rule (next frame = model output) are unchanged.

Extra objects on the JSON line:
  roofline      the dominant kernel of the timed path against HBM — the per-source GEMM of the
                factored conv (what conv_mode "auto" runs at this size; algorithmic bytes per launch
                E*k*4 + R*C*k*4 + 2*E*C*4) or, in materialized mode, the conv (gather ->
                per-edge matvec -> scatter-mean) kernel (16,388*E + 516*R + 4):
                bytes / average launch duration, measured with HIP events on the launching stream over
                K more steps of the same rollout issued as plain launches (events cannot sit inside a
                hipGraph replay); traffic = PMC bytes per launch from profiles/roofline_traffic.json.
  rooflines     the same for every leg: both conv formulations (the other one is run as a comparison
                leg on the same start window) and the two wide split-bf16 GEMMs against the bf16 MFMA peak.
  cpu_baseline  the oracle (CPU restatement of the reference: edge-MLP re-evaluated in all 12 conv
                applications + scipy graph rebuild per step) timed on this box's host cores on a
                bounded sample; rank 0, N=1 only.  A reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA = vector fp32 peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--members-per-gpu", type=int, default=1)
    ap.add_argument("--atoms", type=int, default=504)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--kernel-width", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--window", type=int, default=10)
    ap.add_argument("--threshold", type=float, default=8.0)
    ap.add_argument("--no-graph", action="store_true", help="plain launches instead of hipGraph replay")
    ap.add_argument("--gemm-mode", choices=["split_bf16", "f32"], default="split_bf16",
                    help="edge-MLP GEMMs: exact 3-way bf16 split (6 products, fp32 accumulate) or fp32-input MFMA")
    ap.add_argument("--single-mode", action="store_true", help="skip the comparison leg in the other conv mode")
    ap.add_argument("--conv-mode", choices=["auto", "materialized", "factored"], default="auto",
                    help="materialized = W_e written once and streamed by every conv application (the reference's "
                         "formulation); factored = same sums reassociated per node, W_e never formed")
    ap.add_argument("--variant", choices=["intree", "notebook"], default="intree",
                    help="notebook = the model the reference's notebook ran (no LSTM, conv1 only; use with "
                         "--atoms 28 --window 1 --kernel-width 512 --chain for the nb:370 shape)")
    ap.add_argument("--chain", action="store_true", help="random-walk C-alpha chain frame instead of the uniform box")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-roofline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=45.0)
    return ap.parse_args()


def cpu_baseline(sd, depth, window, aa, threshold, budget_s):
    """Reference-faithful CPU step on this host: forward with the edge-MLP evaluated in every conv
    application (hoist=False) + scipy graph rebuild (graph_kernel.py:396-413)."""
    from oracle import graph_kernel_oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
    aa = aa.cpu()
    t0 = time.perf_counter()
    s = O.construct_pairdata(window, aa, threshold)
    t_graph = time.perf_counter() - t0
    E = int(s["edge_index"].shape[1])
    # one conv application (edge-MLP + gather/matvec/scatter) to size the sample
    x = torch.randn(window.shape[1], sd_cpu["fc1.weight"].shape[0])
    O.edge_mlp(s["edge_attr"][:2048], sd_cpu, "conv1.net.")  # warm the thread pool
    t0 = time.perf_counter()
    w_e = O.edge_mlp(s["edge_attr"], sd_cpu, "conv1.net.")
    O.nnconv_apply(x, s["edge_index"], w_e, sd_cpu["conv1.root"], sd_cpu["conv1.bias"], "mean")
    t_conv = time.perf_counter() - t0
    del w_e
    est = 2 * depth * t_conv + t_graph
    if est <= budget_s:
        t0 = time.perf_counter()
        O.recursive_propagation(sd_cpu, depth, s, 1, threshold, hoist=False)
        t_step = time.perf_counter() - t0
        sample = (f"1 full rollout step: forward with the edge-MLP evaluated {2 * depth}x as the reference does + "
                  f"scipy graph rebuild, N={window.shape[1]}, E={E}")
    else:
        t_step = est
        sample = (f"1 of the {2 * depth} conv applications (edge-MLP + conv, {t_conv:.2f}s) x {2 * depth} + measured "
                  f"scipy graph rebuild ({t_graph:.3f}s); full step estimated, N={window.shape[1]}, E={E}")
    return {"value": 1.0 / t_step, "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample,
            "seconds_per_frame": t_step}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)     # rehearsal on a 1-GPU box: ranks share the card
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("MDNO_BENCH_BACKEND", "nccl")   # "gloo" only to rehearse N>1 on one GPU
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, KernelNNNotebook
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine, default_edge_cap, gather_trajectories
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict

    N, W, M = a.atoms, a.window, a.members_per_gpu
    total_members = M * world
    sd = near_identity_state_dict(a.width, a.kernel_width, seed=0, kernel_gain=1e-3, feature_gain=0.1)
    if a.variant == "notebook":
        sd = {k: v for k, v in sd.items() if not k.startswith(("lstm", "conv2"))}
        model = KernelNNNotebook(a.width, a.kernel_width, a.depth, 6, 7, 3, 20, 4)
    else:
        model = KernelNN(a.width, a.kernel_width, a.depth, 6, 7, 3, 20, 4)
    model.load_state_dict(sd)
    model.eval().to(dev)
    model.gemm_mode = a.gemm_mode
    model.conv_mode = a.conv_mode

    frame0 = syn.chain_frame(N, seed=1) if a.chain else syn.box_frame(N, seed=1)
    base = syn.jitter_window(frame0, W, seed=1)                                    # [W,N,3]
    wins = np.stack([base if (total_members == 1) else
                     syn.ensemble_windows(base, 1, sigma=0.1, seed0=100 + rank + world * m)[0]
                     for m in range(M)], axis=1)                                   # [W,M,N,3] member = rank + world*m
    aa = torch.from_numpy(syn.amino_acids(N, seed=1))
    max_steps = a.warmup + a.steps + (0 if a.skip_roofline else a.steps)
    cap = default_edge_cap(M, N, a.threshold)
    eng = RolloutEngine(model, M, N, W, a.threshold, max_steps=max_steps, edge_cap=cap, device=dev,
                        use_graph=not a.no_graph)
    eng.reset(torch.from_numpy(wins), aa)
    mode = eng.conv_mode            # what "auto" resolved to at this edge capacity

    # ---- warm-up (untimed): also captures nothing new — the step graph was captured in reset()
    eng.step(a.warmup)
    eng.synchronize()
    if world > 1:    # the collective of the timed region, once, untimed: communicator set-up and buffers
        gather_trajectories(torch.zeros((a.steps, M, N, 3), dtype=torch.float32, device=dev), total_members)
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps + trajectory collection
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(a.steps)
    eng.stream.synchronize()
    produced = eng.traj[W + a.warmup:W + a.warmup + a.steps]                       # [K,M,N,3]
    if world > 1:
        full = gather_trajectories(produced, total_members)
    else:
        full = produced
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    eng.synchronize()   # raises on edge overflow / bad input
    assert full.shape == (a.steps, total_members, N, 3) and bool(torch.isfinite(full).all())
    eps = eng.edges_per_step[a.warmup:a.warmup + a.steps].double()
    e_mean = float(eps.mean().item())
    frames = a.steps * total_members
    value = frames / elapsed

    # ---- roofline leg: K more steps of the SAME rollout, plain launches bracketed by HIP events on
    # the launch stream (events cannot sit inside a hipGraph replay)
    roofs = {}
    kernels = {}
    other_mode = None
    R, C, KW = M * N, a.width, a.kernel_width

    def timed_leg(engine, first_step):
        engine.attach_timer(a.steps * (6 * a.depth + 20))
        engine.step(a.steps)
        tm = engine.read_timer()
        engine.detach_timer()
        engine.synchronize()
        e = float(engine.edges_per_step[first_step:first_step + a.steps].double().mean().item())
        ks = {k: {"avg_ms": ms / n, "launches": int(n), "ms_per_step": ms / a.steps} for k, (ms, n) in tm.items() if n}
        return ks, e

    def conv_roofline(ks, e):      # the metric's kernel: gather -> per-edge matvec -> scatter-mean, HBM-bound
        avg_s = ks["nnconv"]["avg_ms"] * 1e-3
        alg = e * (C * C * 4 + 4) + (R + 1) * 4 + 2 * R * C * 4
        r = {"bound": "hbm", "kernel": "nnconv64_row_kernel", "conv_mode": "materialized", "achieved": alg / avg_s / 1e9,
             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / avg_s / 1e9 / HBM_PEAK_GBS,
             "frac_of_measured_copy_peak": alg / avg_s / 1e9 / HBM_COPY_GBS, "traffic": None,
             "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_s * 1e3, "edges_per_launch": e,
             "rows_per_launch": R}
        tf = REPO / "profiles" / "roofline_traffic.json"
        if tf.exists():
            try:
                r["traffic"] = json.loads(tf.read_text()).get("nnconv_hbm_bytes_per_launch")
            except Exception:
                pass
        return r

    def gemm_roofline(ks, e, which, n_out):
        step_s = ks[which]["ms_per_step"] * 1e-3     # all launches of a step (capacity-sized chunks past E exit at once)
        flops32 = 2.0 * e * KW * n_out                # fp32-equivalent work
        if a.gemm_mode == "f32":
            return {"bound": "mfma", "kernel": "gemm_tn_mfma_kernel", "achieved": flops32 / step_s / 1e12,
                    "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops32 / step_s / 1e12 / MFMA_F32_PEAK_TFLOPS,
                    "ms_per_step": step_s * 1e3}
        ach = 6.0 * flops32 / step_s / 1e12           # 6 bf16 plane products executed per fp32 product
        return {"bound": "mfma", "kernel": "gemm_split_bf16_kernel", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS, "fp32_equivalent_tflops": flops32 / step_s / 1e12,
                "ms_per_step": step_s * 1e3,
                "note": "executed bf16 MFMA flops (6 plane products per fp32 product) vs dense bf16 peak"}

    def per_source_roofline(ks, e):   # factored path: M_j = H_j . Y_j^T, one launch per conv application
        avg_s = ks["nnconv"]["avg_ms"] * 1e-3
        alg = e * KW * 4 + R * C * KW * 4 + 2 * e * C * 4 + (R + 1) * 4        # H once + Y once + 2 k-slice partials out
        flops = 2.0 * e * KW * C                                                # fp32-equivalent
        split = a.gemm_mode == "split_bf16"
        # matrix-pipe work as executed: 6 bf16 plane products per fp32 product, or the fp32 MFMA itself
        mfma_exec, mfma_peak = (6.0 * flops, MFMA_BF16_PEAK_TFLOPS) if split else (flops, MFMA_F32_PEAK_TFLOPS)
        t_hbm, t_mfma = alg / (HBM_PEAK_GBS * 1e9), mfma_exec / (mfma_peak * 1e12)
        name = "gemm_per_source_split_kernel" if split else "gemm_per_source_kernel"
        r = {"kernel": name, "conv_mode": "factored", "avg_launch_ms": avg_s * 1e3,
             "algorithmic_bytes_per_launch": alg, "flops_per_launch": flops, "traffic": None,
             "hbm_GBps": alg / avg_s / 1e9, "hbm_frac": alg / avg_s / 1e9 / HBM_PEAK_GBS,
             "frac_of_measured_copy_peak": alg / avg_s / 1e9 / HBM_COPY_GBS,
             "mfma_TFLOPs": mfma_exec / avg_s / 1e12, "mfma_frac": mfma_exec / avg_s / 1e12 / mfma_peak}
        tf = REPO / "profiles" / "roofline_traffic.json"
        if tf.exists():
            try:
                r["traffic"] = json.loads(tf.read_text()).get(name + "_hbm_bytes_per_launch")
            except Exception:
                pass
        if t_hbm >= t_mfma:
            r.update(bound="hbm", achieved=r["hbm_GBps"], peak=HBM_PEAK_GBS, unit="GB/s", frac=r["hbm_frac"])
        else:
            r.update(bound="mfma", achieved=r["mfma_TFLOPs"], peak=mfma_peak, unit="TFLOP/s", frac=r["mfma_frac"])
        return r

    if not a.skip_roofline:
        kernels, e2 = timed_leg(eng, a.warmup + a.steps)
        if mode == "materialized":
            roofs["conv_materialized"] = conv_roofline(kernels, e2)
            roofs["edge_mlp_last_gemm"] = gemm_roofline(kernels, e2, "edge_mlp_gemm2", C * C)
        else:
            roofs["conv_factored_per_source_gemm"] = per_source_roofline(kernels, e2)
        roofs["edge_mlp_hidden_gemm"] = gemm_roofline(kernels, e2, "edge_mlp_gemm1", KW)
        # ---- the other conv formulation on the same start window: frames/s and, for the materialised
        # one, the HBM roofline of the gather/matvec/scatter kernel BASELINE.json's target is stated on
        if a.variant == "intree" and not a.single_mode:
            om = "materialized" if mode == "factored" else "factored"
            model.conv_mode = om
            eng2 = RolloutEngine(model, M, N, W, a.threshold, max_steps=a.warmup + 2 * a.steps, edge_cap=cap, device=dev,
                                 use_graph=not a.no_graph)
            eng2.reset(torch.from_numpy(wins), aa)
            eng2.step(a.warmup)
            eng2.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng2.step(a.steps)
            eng2.stream.synchronize()
            dt = time.perf_counter() - t0
            k2, e3 = timed_leg(eng2, a.warmup + a.steps)
            other_mode = {"conv_mode": om, "frames_per_s_this_rank": a.steps * M / dt, "ms_per_step": dt / a.steps * 1e3,
                          "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in k2.items()}}
            if om == "materialized":
                roofs["conv_materialized"] = conv_roofline(k2, e3)
                roofs["edge_mlp_last_gemm"] = gemm_roofline(k2, e3, "edge_mlp_gemm2", C * C)
            else:
                roofs["conv_factored_per_source_gemm"] = per_source_roofline(k2, e3)
            eng2.close()
            model.conv_mode = a.conv_mode
    # "roofline" = the dominant kernel of the TIMED path
    dominant = None
    if kernels:
        name = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        dominant = {"nnconv": roofs.get("conv_materialized" if mode == "materialized"
                                        else "conv_factored_per_source_gemm"),
                    "edge_mlp_gemm2": roofs.get("edge_mlp_last_gemm"),
                    "edge_mlp_gemm1": roofs.get("edge_mlp_hidden_gemm")}.get(name)

    cpu = None
    if rank == 0 and world == 1 and not a.skip_cpu_baseline and a.variant == "intree":
        cpu = cpu_baseline(sd, a.depth, base, aa, a.threshold, a.cpu_budget_s)

    if rank == 0:
        line = {
            "metric": "rolled-out MD frames/sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (uniform-box frames, seeded; near-identity synthetic weights, see weights.py)",
            "config": {"workload": f"BBA all-atom stand-in N={N} r={a.threshold}A free-running autoregressive rollout "
                                   f"(BASELINE configs[1]), {M} member(s) per GPU",
                       "atoms": N, "window": W, "width": a.width, "kernel_width": a.kernel_width, "depth": a.depth,
                       "members_per_gpu": M, "total_members": total_members, "mean_edges_per_member": e_mean / M,
                       "edges_first_last": [int(eps[0].item()), int(eps[-1].item())], "edge_cap": cap,
                       "parallelism": f"ensemble-sharded x{world}, one all-gather of trajectories",
                       "launch": "plain" if a.no_graph else "hipGraph replay", "edge_mlp_gemm": a.gemm_mode,
                       "variant": a.variant, "conv_mode": mode, "conv_mode_requested": a.conv_mode},
            "roofline": dominant, "rooflines": roofs, "other_conv_mode": other_mode, "cpu_baseline": cpu,
            "kernels": kernels,
        }
        print(json.dumps(line))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
