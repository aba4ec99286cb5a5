#!/usr/bin/env python3
"""Headline benchmark: rolled-out MD frames/sec (BASELINE.json), one process per GPU.

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts N fresh worker processes itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W        (or under an external launcher)

Workload.  N=504 atoms uniform at 0.1 atoms/A^3 (BBA all-atom stand-in), 8 A cutoff (E ~ 60k directed
edges incl. self-loops per member), window 10, the reference's CLI-default model (width 64,
kernel_width 1024, depth 6 -> 12 conv applications), fp32, free-running autoregressive rollout entirely
on the device.  One "step" = one new frame for every ensemble member (graph rebuild + forward + window
slide).
  --gpus 1 (default)   BASELINE configs[1]: ONE trajectory, 1 GPU.
  --gpus N > 1         BASELINE configs[2]: the 64-member ensemble (`--total-members`, default 64 when
                       N > 1), member m on rank m mod N, no collective while stepping, one RCCL
                       all-gather of the produced frames at the end, inside the timed region.  The total
                       work is the same for every N > 1 ("scaling": "strong"); the like-for-like
                       single-GPU figure (64 members on one GPU) is `python bench.py --gpus 1
                       --total-members 64` and is also reported by the default N=1 run as
                       `ensemble64_single_gpu`.
  --members-per-gpu M  weak-scaling mode instead: every rank runs M members.
value = frames produced by all ranks / max-over-ranks wall time of exactly K steps (+ the gather).

Weights: synthetic near-identity set (weights.py) — no trained checkpoint exists offline and
random-init weights collapse the cloud to one point, which would change E (the cost driver) after
one step.  The architecture, arithmetic and update rule (next frame = model output) are unchanged.

Extra objects on the JSON line:
  roofline      the dominant kernel of the timed path against HBM.  Factored conv (what conv_mode "auto" runs at
                this size; csrc/moment.hip): K1 `moment_kernel`, S_t = sum_(e->t) x_src (x) h_e — algorithmic bytes
                per launch E*k*4 (H in) + R*64*k*4 (S out) + E*4 (src) + (R+1)*4 (row_ptr), DESIGN.md §4.5.
                Materialized conv: the gather -> per-edge matvec -> scatter-mean kernel (SURVEY.md §8d:
                16,388*E + 516*R + 4).  bytes / average launch duration, measured with HIP events on the
                launching stream over K more steps of the same rollout issued as plain launches (events cannot
                sit inside a hipGraph replay): `avg_launch_ms_events`; `avg_launch_us_rocprof` = the same kernel's
                average in the committed rocprofv3 trace (profiles/roofline_traffic.json), and traffic = PMC bytes
                per launch from there, both only when this run's (atoms, members, conv mode, GEMM mode) is the
                profiled one.
  rooflines     the same for every leg: `conv_factored_moment` (K1) with a `per_application` object — the whole conv
                application K1 + K2 + K3 against its COMPULSORY bytes (H + W3R + x + y: the S image and the K-slice
                partials are intermediates), i.e. the composite and not only the best kernel —, `conv_materialized`
                (the other formulation, run as a comparison leg on the same start window) and the two wide
                split-plane GEMMs against the 16-bit MFMA peak.
  cpu_baseline  the oracle (CPU restatement of the reference: edge-MLP re-evaluated in all 12 conv
                applications + scipy graph rebuild per step) timed on this box's host cores on a
                bounded sample; rank 0, N=1 only.  A reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
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
# what the chip SUSTAINS on v_mfma_f32_32x32x16_f16 beside one 1-KiB LDS-DMA piece per 4 MFMAs and wave from beyond L2 —
# the hidden GEMM's data-movement intensity — measured by scripts/micro/mfma_sustained.hip (EXPERIMENTS.md §00.6;
# MFMAs alone: 1.5-1.7 PF): context beside the nominal peak, as HBM_COPY_GBS is beside HBM_PEAK_GBS
MFMA_16BIT_SUSTAINED_TFLOPS = 1480.0
ENSEMBLE_MEMBERS = 64         # BASELINE.json configs[2]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--total-members", type=int, default=None,
                    help="ensemble size shared by all ranks (member m on rank m mod N); default 1 for --gpus 1, "
                         f"{ENSEMBLE_MEMBERS} for --gpus > 1")
    ap.add_argument("--members-per-gpu", type=int, default=None, help="weak scaling: every rank runs this many members")
    ap.add_argument("--atoms", type=int, default=504)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--kernel-width", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--window", type=int, default=10)
    ap.add_argument("--threshold", type=float, default=8.0)
    ap.add_argument("--no-graph", action="store_true", help="plain launches instead of hipGraph replay")
    ap.add_argument("--member-groups", type=int, default=0,
                    help="a rank's members as this many independent engines on their own streams, stepped together "
                         "(rollout.GroupedRolloutEngine; frames bitwise unchanged); 0 = groups of four members (at least 2 "
                         "groups, at most 16) from 2 members of >= 256 atoms on, else 1")
    ap.add_argument("--gemm-mode", choices=["split_bf16", "split_f16", "f32"], default="split_f16",
                    help="edge-MLP GEMMs: exact 3-way bf16 split (6 products, fp32 accumulate); split_f16 = the same "
                         "with the hidden layer of the factored path on 2 fp16 planes (3 products, device-side "
                         "fallback to bf16 out of fp16 range); f32 = fp32-input MFMA")
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
    ap.add_argument("--skip-ensemble-leg", action="store_true",
                    help="N=1 default run: skip the 64-member single-GPU leg (cfg3's like-for-like baseline)")
    ap.add_argument("--skip-rank-share-leg", action="store_true",
                    help="N=1 default run: skip the leg that runs 8 members (one rank's share of configs[2] at 8 GPUs) in a "
                         "child process through a world-size-1 RCCL group")
    ap.add_argument("--skip-config-legs", action="store_true",
                    help="N=1 default run: skip the cfg4 (training), cfg5 (50k-atom box) and shape-A (N=28) legs")
    ap.add_argument("--cpu-budget-s", type=float, default=120.0,
                    help="bound on the full reference-faithful CPU step; estimated first from one conv application")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launcher
def visible_gpus() -> int:
    """GPUs the workers will see, found WITHOUT a HIP call in this (launcher) process — torch.cuda.device_count()
    falls back to hipGetDeviceCount on builds without amdsmi, which opens the runtime here.  The visibility
    variables if one is set; else a short-lived child asks the runtime (a container can expose fewer devices
    than the KFD topology lists, so sysfs alone is not trusted); the topology count only if that child fails."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            return int(r.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        pass
    n = 0
    for prop in Path("/sys/class/kfd/kfd/topology/nodes").glob("*/properties"):
        try:
            for line in prop.read_text().splitlines():
                k, _, val = line.partition(" ")
                if k == "simd_count" and int(val) > 0:
                    n += 1
        except (OSError, ValueError):
            pass
    return n


RANK_GRACE_S = 30.0           # after the first rank fails, the others get this long before they are stopped
INIT_TIMEOUT_S = 300          # rendezvous + every collective (the process group's timeout); the driver allows 600 s


def launch_workers(a, script=None, argv=None) -> int:
    """`python bench.py --gpus N` with no launcher around it: start N fresh worker processes (this
    process has made no GPU call and makes none), relay rank 0's JSON line, return the worst exit code.

    Every rank is polled from the start while a thread drains rank 0's pipe: the first rank that exits
    non-zero starts a RANK_GRACE_S clock, after which exactly the children started here are stopped
    (SIGTERM, then SIGKILL) — a dead rank costs half a minute, not the rendezvous timeout.  A failed run
    prints ONE JSON line {"error": ..., "rank_exit_codes": [...]} on stdout and returns non-zero.
    (`script` / `argv`: what each rank runs — this file with this command line; tests substitute a stub.)"""
    import threading
    script = str(Path(__file__).resolve()) if script is None else str(script)
    argv = sys.argv[1:] if argv is None else list(argv)
    n = a.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL needs it)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    ndev = visible_gpus()                                   # no HIP call in this process (see there)
    if ndev < n and "MDNO_BENCH_BACKEND" not in env:
        print(f"bench.py: {n} ranks on {ndev} visible GPU(s): ranks share cards, collective over gloo "
              "(rehearsal only — RCCL needs one GPU per rank)", file=sys.stderr)
        env["MDNO_BENCH_BACKEND"] = "gloo"
    procs = []
    try:
        err_fd = sys.stderr.fileno()
    except (OSError, ValueError, AttributeError):           # a replaced sys.stderr (test capture): the process's fd 2
        err_fd = 2
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else err_fd, text=(r == 0)))
    out0: list = []
    drain = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    drain.start()
    codes = [None] * n
    deadline, first_bad, stopped = None, None, []
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
                if codes[i] not in (None, 0) and first_bad is None:
                    first_bad = i
                    deadline = time.time() + RANK_GRACE_S
                    print(f"bench.py: rank {i} exited with code {codes[i]}; the other ranks get {RANK_GRACE_S:.0f} s",
                          file=sys.stderr, flush=True)
        if deadline is not None and time.time() > deadline:
            live = [i for i in range(n) if codes[i] is None]
            for i in live:
                procs[i].terminate()                        # exactly the children started above
            t_kill = time.time() + 5.0
            for i in live:
                try:
                    codes[i] = procs[i].wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    procs[i].kill()
                    codes[i] = procs[i].wait()
            stopped = live
        time.sleep(0.05)
    drain.join(timeout=10.0)
    text = out0[0] if out0 else ""
    bad = [c for c in codes if c != 0]
    if bad:
        # rank 0's own line (if it got that far) is not relayed as a result: a failed run has no value
        err = {"error": f"rank {first_bad} exited with code {codes[first_bad]}"
                        + (f"; ranks {stopped} were stopped after {RANK_GRACE_S:.0f} s" if stopped else ""),
               "rank_exit_codes": codes, "n_gpus": n}
        for ln in (text or "").splitlines():               # rank 0's own error line, if it printed one
            try:
                d = json.loads(ln)
            except ValueError:
                continue
            if isinstance(d, dict) and "error" in d:
                err["rank0_error"] = d["error"]
        print(json.dumps(err), flush=True)
        return max(bad) if max(bad) > 0 else 1
    sys.stdout.write(text or "")
    sys.stdout.flush()
    return 0


# ----------------------------------------------------------------------------------------------- CPU leg
def note(msg: str) -> None:
    """Progress line on stderr (the JSON line is the only thing on stdout)."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _cores() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box
    shows all 256 host threads in the mask but grants 16 CPUs of quota — 256 torch threads on that
    quota are throttled to a crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())             # cgroup v1
            p = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(sd, depth, window, aa, threshold, budget_s, first_frame_gpu=None):
    """Reference-faithful CPU step on this host (graph_kernel.py:396-413): forward with the edge-MLP
    evaluated in every conv application (hoist=False) + scipy graph rebuild.  Legs:
      value      shape B (this workload), all host cores: ONE full rollout step, timed (estimated from one
                 conv application only if that estimate exceeds the budget — stated in `sample`)
      one_thread shape B at torch.set_num_threads(1) — the reference's own setting when
                 num_data_workers == 0 (graph_kernel.py:501) — extrapolated from one conv application on an
                 edge slice (labelled so)
      shape_A    N=28 C-alpha chain (the reference's BBA), median of 20 steps, all cores, and of 5 steps
                 at one thread"""
    from oracle import graph_kernel_oracle as O
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    cores = _cores()
    torch.set_num_threads(cores)
    note(f"cpu baseline: {cores} threads")
    sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
    aa = aa.cpu()
    N = window.shape[1]
    t0 = time.perf_counter()
    s = O.construct_pairdata(window, aa, threshold)
    t_graph = time.perf_counter() - t0
    E = int(s["edge_index"].shape[1])
    x = torch.randn(N, sd_cpu["fc1.weight"].shape[0])
    O.edge_mlp(s["edge_attr"][:2048], sd_cpu, "conv1.net.")  # warm the thread pool
    t0 = time.perf_counter()
    w_e = O.edge_mlp(s["edge_attr"], sd_cpu, "conv1.net.")
    O.nnconv_apply(x, s["edge_index"], w_e, sd_cpu["conv1.root"], sd_cpu["conv1.bias"], "mean")
    t_conv = time.perf_counter() - t0
    del w_e
    est = 2 * depth * t_conv + t_graph
    note(f"cpu baseline: one conv application {t_conv:.2f}s, graph {t_graph:.3f}s -> full step ~{est:.0f}s")
    first_frame_cpu = None
    if est <= budget_s:
        t0 = time.perf_counter()
        first_frame_cpu = O.recursive_propagation(sd_cpu, depth, s, 1, threshold, hoist=False)[0]["x_position"][-1].numpy()
        t_step = time.perf_counter() - t0
        sample = (f"1 full rollout step, timed: forward with the edge-MLP evaluated {2 * depth}x as the reference does "
                  f"+ scipy graph rebuild, N={N}, E={E} ({cores} threads)")
        measured = True
    else:
        t_step = est
        sample = (f"ESTIMATE (a full step would exceed --cpu-budget-s {budget_s:.0f}): 1 of the {2 * depth} conv "
                  f"applications (edge-MLP + conv, {t_conv:.2f}s) x {2 * depth} + measured scipy graph rebuild "
                  f"({t_graph:.3f}s), N={N}, E={E} ({cores} threads)")
        measured = False
    out = {"value": 1.0 / t_step, "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample,
           "seconds_per_frame": t_step, "full_step_measured": measured,
           "one_conv_application_s": t_conv, "graph_rebuild_s": t_graph}
    # parity at the bench's own weights and start window: the oracle's first frame (the step just timed; or,
    # when that was only estimated, one step with the shared edge-MLP evaluated once — the same values) against
    # frame W of the timed engine's trajectory
    if first_frame_gpu is not None:
        if first_frame_cpu is None and est / (2 * depth) * 1.5 <= budget_s:
            first_frame_cpu = O.recursive_propagation(sd_cpu, depth, s, 1, threshold, hoist=True)[0]["x_position"][-1].numpy()
        if first_frame_cpu is not None:
            d = first_frame_gpu.astype(np.float64) - first_frame_cpu.astype(np.float64)
            # what the model computes is the displacement from the last window frame: the error relative to
            # THAT (a near-identity model would make any error look small against |coordinates| ~ 10 A)
            disp = first_frame_cpu.astype(np.float64) - np.asarray(window[-1], dtype=np.float64)
            out["parity_first_frame"] = {
                "rel_l2": float(np.linalg.norm(d) / np.linalg.norm(first_frame_cpu)),
                "max_abs_A": float(np.abs(d).max()),
                "rel_l2_of_displacement": float(np.linalg.norm(d) / max(np.linalg.norm(disp), 1e-300)),
                "mean_displacement_A": float(np.linalg.norm(disp, axis=1).mean()),
                "tolerance_rel_l2": 1e-5,
                "what": "frame W of the timed rollout vs the oracle's first step from the same window and weights "
                        "(reference-faithful CPU forward + scipy graph)"}
            out["parity_first_frame_rel_l2"] = out["parity_first_frame"]["rel_l2"]

    # ---- one thread, shape B: one conv application on an edge slice, scaled
    note(f"cpu baseline: full step {t_step:.1f}s ({'timed' if measured else 'estimated'}); one-thread leg")
    torch.set_num_threads(1)
    e_slice = min(E, 3072)
    ei, ea = s["edge_index"][:, :e_slice], s["edge_attr"][:e_slice]
    t0 = time.perf_counter()
    w_e = O.edge_mlp(ea, sd_cpu, "conv1.net.")
    O.nnconv_apply(x, ei, w_e, sd_cpu["conv1.root"], sd_cpu["conv1.bias"], "mean")
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.construct_pairdata(window, aa, threshold)
    t_graph1 = time.perf_counter() - t0
    t_step1 = 2 * depth * t1 * (E / e_slice) + t_graph1
    out["one_thread"] = {"value": 1.0 / t_step1, "unit": "frames/s", "cores": 1, "seconds_per_frame": t_step1,
                         "sample": f"EXTRAPOLATED: one conv application on the first {e_slice} of {E} edges ({t1:.2f}s) "
                                   f"x {E / e_slice:.1f} x {2 * depth} + measured graph rebuild ({t_graph1:.3f}s); "
                                   "torch.set_num_threads(1) as graph_kernel.py:501"}

    # ---- shape A: N=28 chain, the reference's own BBA size (bba_analysis.ipynb:1034), timed steps
    note(f"cpu baseline: one thread ~{t_step1:.0f}s/frame; shape A legs")
    wa = syn.jitter_window(syn.chain_frame(28, seed=0), window.shape[0], seed=0)
    aa_a = torch.from_numpy(syn.amino_acids(28, seed=0))

    def steps_a(n):
        sa = O.construct_pairdata(wa, aa_a, threshold)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            sa = O.recursive_propagation(sd_cpu, depth, sa, 1, threshold, hoist=False)[0]
            ts.append(time.perf_counter() - t0)
        return ts

    ta1 = steps_a(5)
    note(f"cpu baseline: shape A one thread {statistics.median(ta1):.2f}s/frame; {cores} threads, 3 + 20 steps")
    torch.set_num_threads(cores)
    steps_a(3)
    ta = steps_a(20)
    out["shape_A"] = {"atoms": 28, "median_s_per_frame": statistics.median(ta), "frames_per_s": 1.0 / statistics.median(ta),
                      "steps": 20, "warmup": 3, "cores": cores,
                      "one_thread_median_s_per_frame": statistics.median(ta1), "one_thread_steps": 5}
    return out


# ----------------------------------------------------------------------------------------------- extra legs
def _event_ms(fn, reps, warm=2):
    """Mean milliseconds of fn() over `reps` calls, HIP events on torch's current stream (the stream the
    training ops launch on)."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def leg_cfg4_training(dev, frames=10000, batch=128, kernel_width=1024, depth=6, cpu_samples=8):
    """BASELINE configs[3]: operator training on a synthetic preprocessed-BBA-layout trajectory (N=28 C-alpha
    chain, Ornstein-Uhlenbeck jitter, contact maps at 8 A; SURVEY.md §8d cfg4: T=10,000 frames, batch 128), the
    reference's model / optimiser / loss (graph_kernel.py:528-547), ONE timed epoch per precision after a
    two-batch warm-up, batches built on the device from the resident trajectory.  Also: the dominant GEMM and
    conv kernels of the bf16 step timed alone at the batch's shapes (HIP events), and the same step in torch
    autograd over the oracle's formulas on the host as the CPU baseline."""
    import tempfile
    from molecular_dynamics_neural_operator_amd import ops, synthetic as syn
    from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset, write_trajectory_npz
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, LpLoss
    from molecular_dynamics_neural_operator_amd.training import Adam, DeviceTrajectory, train_epoch
    N, W = 28, 10
    traj = syn.ou_trajectory(syn.chain_frame(N, seed=0), frames, sigma=0.3, theta=0.1, seed=2)
    cms = [syn.contact_map(f, 8.0) for f in traj]
    with tempfile.TemporaryDirectory() as td:
        path = Path(td) / "synthetic_bba.npz"
        write_trajectory_npz(path, traj, cms, syn.amino_acids(N, seed=0))
        dset = ContactMapDataset(str(path), window_size=W, horizon=1)
    dtraj = DeviceTrajectory(dset, dev)
    n_train = int(len(dset) * 0.8)                                      # partition split 0.8 (graph_kernel.py:509-520)
    idx = [list(range(s, s + batch)) for s in range(0, n_train - batch + 1, batch)]      # drop_last
    out = {"frames": frames, "atoms": N, "window": W, "batch_size": batch, "train_batches": len(idx),
           "kernel_width": kernel_width, "depth": depth, "collate": "device (mdno_collate_samples)",
           "optimizer": "training.Adam(lr=1e-4, weight_decay=5e-4): torch.optim.Adam's update, one libmdno launch (mdno_adam_step)"}
    E0 = None
    for precision in ("bf16", "fp32"):
        torch.manual_seed(0)
        model = KernelNN(64, kernel_width, depth, 6, 7, 3, 20, 4)
        with torch.no_grad():     # the reference's init explodes through 12 layers at k=1024: damp the kernel's last layer
            for p_ in model.conv1.net.layers[4].parameters():
                p_.mul_(0.05)
        model.to(dev)
        model.train_precision = precision
        opt = Adam(model.parameters(), lr=1e-4, weight_decay=5e-4)      # torch.optim.Adam's update as one libmdno launch
        loss_fn = LpLoss(size_average=False)
        train_epoch(model, (dtraj.batch(i) for i in idx[:2]), opt, loss_fn)              # warm-up
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        tl, mse = train_epoch(model, (dtraj.batch(i) for i in idx), opt, loss_fn)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[precision] = {"samples_per_s": len(idx) * batch / dt, "ms_per_batch": dt / len(idx) * 1e3, "epoch_s": dt,
                          "train_loss": tl, "train_mse": mse, "peak_memory_MiB": torch.cuda.max_memory_allocated() / 2**20}
        note(f"cfg4 {precision}: {out[precision]['samples_per_s']:.0f} samples/s, {out[precision]['ms_per_batch']:.2f} ms/batch")
        if precision == "bf16":
            b0 = dtraj.batch(idx[0])
            E0 = int(b0.edge_index.shape[1])
            # the step's dominant kernels, alone, at this batch's shapes — on the step's OWN operands: h2 is the model's
            # hidden activation of this batch (post-ReLU, half of it zeros; the chip holds a higher clock on it than on
            # dense random bits), dW_e a dense random stand-in (its values are dense in the real step too); the same
            # products on all-random operands alongside
            net = model.conv1.net
            with torch.no_grad():
                h1 = ops.linear_smallk_bf16(b0.edge_attr, net.layers[0].weight.detach(), net.layers[0].bias.detach(), relu=True)
                h2 = ops.linear_bf16(h1, net.layers[2].weight.detach(), net.layers[2].bias.detach(), relu=True, out_bf16=True)
            h2r = torch.randn(E0, kernel_width, device=dev).to(torch.bfloat16)
            w2, bb2 = net.layers[4].weight.detach(), net.layers[4].bias.detach()
            w2b = ops.cast_bf16(w2)
            fl = 2.0 * E0 * kernel_width * 4096

            # these entry points take the fp32 MASTER weight and cast it per call (a 4.2M-element kernel in front of the
            # GEMM): timed alone and taken off, so that "ms" is the GEMM kernel (+ its slab reduction for A^T.B)
            ms_cast = _event_ms(lambda: ops.cast_bf16(w2), 20)
            w2t = ops.transpose(w2)

            def gemm_entry(what, fn, fn_random, cast):
                ms_all, msr_all = _event_ms(fn, 10), _event_ms(fn_random, 10)
                ms, msr = ms_all - cast, msr_all - cast
                return {"bound": "mfma", "kernel": what, "ms": ms, "ms_incl_weight_cast": ms_all, "achieved": fl / ms / 1e9,
                        "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS,
                        "operands": "h2 = this batch's hidden activations (post-ReLU), dW_e dense random",
                        "all_random_operands": {"ms": msr, "achieved": fl / msr / 1e9, "frac": fl / msr / 1e9 / MFMA_BF16_PEAK_TFLOPS}}
            roofs = {"gemm_last_layer_fwd": gemm_entry(
                "gemm_pp_kernel<NT, bias>: bf16 A.W^T [E,k]x[k,4096] -> bf16",
                lambda: ops.linear_bf16(h2, w2, bb2, relu=False, out_bf16=True),
                lambda: ops.linear_bf16(h2r, w2, bb2, relu=False, out_bf16=True), ms_cast)}
            dwe = torch.randn(E0, 4096, device=dev).to(torch.bfloat16)
            roofs["gemm_weight_grad"] = gemm_entry("gemm_pp_kernel<TN, slab>: bf16 A^T.B [E,4096]^T x [E,k]",
                                                   lambda: ops.gemm_atb_bf16(dwe, h2), lambda: ops.gemm_atb_bf16(dwe, h2r), 0.0)
            roofs["gemm_input_grad_masked"] = gemm_entry(
                "gemm_pp_kernel<NT, mask>: bf16 (h2 > 0) * (dW_e . W2) [E,4096]x[4096,k] -> bf16",
                lambda: ops.linear_bf16_relu_bwd(dwe, w2t, h2), lambda: ops.linear_bf16_relu_bwd(dwe, w2t, h2r), ms_cast)
            out["weight_cast_ms"] = ms_cast
            del h1, h2r, w2b, w2t
            g = ops.coo_to_csr(b0.edge_index, batch * N, validate=False)
            x = torch.randn(batch * N, 64, device=dev)
            root, cb = model.conv1.root.detach(), model.conv1.bias.detach()
            ms = _event_ms(lambda: ops.nnconv_bf16w(x, g, dwe, root, cb, "mean", relu=True), 20)
            byts = E0 * (64 * 64 * 2 + 4) + (batch * N + 1) * 4 + 2 * batch * N * 64 * 4
            roofs["conv_fwd_bf16_We"] = {"bound": "hbm", "kernel": "nnconv64_bf16w_kernel", "ms": ms, "achieved": byts / ms / 1e6,
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byts / ms / 1e6 / HBM_PEAK_GBS,
                                         "algorithmic_bytes_per_launch": byts}
            out["rooflines"] = roofs
            out["edges_per_batch"] = E0
            del h2, dwe, x
        del model, opt
        torch.cuda.empty_cache()
    # CPU baseline: the same step (forward, LpLoss, backward) in torch autograd over the oracle's formulas
    from oracle import graph_kernel_oracle as O
    torch.set_num_threads(_cores())
    torch.manual_seed(0)
    ref = KernelNN(64, kernel_width, depth, 6, 7, 3, 20, 4)
    sd = {k: v.detach() for k, v in ref.state_dict().items()}
    samples = [dset[i] for i in range(cpu_samples)]
    dicts = [dict(x_position=s_.x_position, x_aminoacid=s_.x_aminoacid, y=s_.y, edge_index=s_.edge_index,
                  edge_attr=s_.edge_attr) for s_ in samples]
    O.train_step(sd, dicts[:1], depth, dtype=torch.float32)
    t0 = time.perf_counter()
    O.train_step(sd, dicts, depth, dtype=torch.float32)
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": cpu_samples / dt, "unit": "samples/s", "cores": _cores(), "kind": "port",
                           "sample": f"forward + LpLoss + backward of {cpu_samples} samples, torch autograd over the oracle's "
                                     "formulas (edge-MLP evaluated once per sample, not 12x), no optimizer step"}
    # ---- "loss after 1 epoch vs the CPU restatement" (SURVEY.md §8d cfg4): a 16-batch mini-epoch of the reference's
    # loop (graph_kernel.py:445-474: forward, LpLoss(size_average=False), backward, Adam step per batch; the epoch's
    # average loss is what train() returns), same start parameters and batches, on the device in both precisions and
    # on the host in fp64 over the oracle's train step
    try:
        mb, mbs = 16, 4
        ep_idx = [list(range(200 + b_ * mbs, 200 + (b_ + 1) * mbs)) for b_ in range(mb)]
        sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
        with torch.no_grad():
            for k_ in ("conv1.net.layers.4.weight", "conv1.net.layers.4.bias"):
                sd0[k_] = sd0[k_] * 0.05
                sd0[k_.replace("conv1", "conv2")] = sd0[k_]
        ep = {"batches": mb, "batch_size": mbs, "optimizer": "Adam(lr=1e-4, weight_decay=5e-4)"}
        for precision in ("fp32", "bf16"):
            m_ = KernelNN(64, kernel_width, depth, 6, 7, 3, 20, 4)
            m_.load_state_dict(sd0)
            m_.to(dev)
            m_.train_precision = precision
            o_ = Adam(m_.parameters(), lr=1e-4, weight_decay=5e-4)
            ep[f"hip_{precision}"], _ = train_epoch(m_, (dtraj.batch(i) for i in ep_idx), o_, LpLoss(size_average=False))
            del m_, o_
        t0 = time.perf_counter()
        pp = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd0.items() if not k.startswith("conv2.net.")}
        o_ = torch.optim.Adam(list(pp.values()), lr=1e-4, weight_decay=5e-4)
        tot = 0.0
        for i in ep_idx:
            dd = [dict(x_position=s_.x_position, x_aminoacid=s_.x_aminoacid, y=s_.y, edge_index=s_.edge_index,
                       edge_attr=s_.edge_attr) for s_ in (dset[j] for j in i)]
            l_, _, g_ = O.train_step({k: v.detach() for k, v in pp.items()}, dd, depth)
            o_.zero_grad()
            for k, v in pp.items():
                v.grad = g_[k].detach().double().clone()
            o_.step()
            tot += l_
        ep["oracle_fp64"] = tot / mb
        ep["oracle_seconds"] = time.perf_counter() - t0
        ep["rel_diff_fp32"] = abs(ep["hip_fp32"] - ep["oracle_fp64"]) / abs(ep["oracle_fp64"])
        ep["rel_diff_bf16"] = abs(ep["hip_bf16"] - ep["oracle_fp64"]) / abs(ep["oracle_fp64"])
        out["loss_after_1_epoch"] = ep
        note(f"cfg4 mini-epoch loss: hip fp32 {ep['hip_fp32']:.6f} bf16 {ep['hip_bf16']:.6f} oracle {ep['oracle_fp64']:.6f} "
             f"({ep['oracle_seconds']:.0f} s on the host)")
    except Exception as e:     # noqa: BLE001 — recorded in the line
        out["loss_after_1_epoch"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def leg_cfg5_shape_c(dev, atoms=50000, cutoff=10.0, steps=2, slice_edges=2_000_000):
    """BASELINE configs[4] (SURVEY.md §8 shape C): synthetic 50k-atom box, 10 A cutoff, the full model, factored
    conv (the materialised W_e would be 298 GB): `steps` timed rollout steps issued as plain launches with the
    per-kernel HIP-event timer attached (graph build on the device included)."""
    from molecular_dynamics_neural_operator_amd import ops, synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    N, W, KW, C = atoms, 10, 1024, 64
    frame = syn.box_frame(N, seed=3)
    win = syn.jitter_window(frame, W, sigma=0.01, seed=3)
    aa = torch.from_numpy(syn.amino_acids(N, seed=3))
    g = ops.radius_graph(torch.from_numpy(frame).to(dev), N, cutoff, edge_cap=int(N * 500))
    E = g.edge_count()
    deg_max = int((g.row_ptr[1:] - g.row_ptr[:-1]).max().item())
    # ---- the stress BASELINE configs[4] names: the gather / per-edge matvec / scatter-mean kernel (materialised conv,
    # SURVEY.md §8d's kernel) alone on the first rows of this graph holding ~2.0M edges, fp32 W_e (33 GB of the 298 GB
    # the whole box would need): HIP events around 10 launches on torch's current stream (the one it is launched on)
    conv_slice = None
    try:
        from molecular_dynamics_neural_operator_amd import _lib
        lib = _lib.load()
        rows = int(torch.searchsorted(g.row_ptr.long(), slice_edges).item())
        Es = int(g.row_ptr[rows].item())
        x = torch.randn(N, 64, device=dev)
        w_e = torch.empty(Es, 4096, device=dev).normal_(0.0, 0.02)
        root, cbias, y = torch.randn(64, 64, device=dev) * 0.1, torch.randn(64, device=dev), torch.empty(rows, 64, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def conv():
            _lib.check(lib.mdno_nnconv_fwd(x.data_ptr(), g.row_ptr.data_ptr(), g.src.data_ptr(), rows, w_e.data_ptr(),
                                           root.data_ptr(), cbias.data_ptr(), 64, 64, 1, 1, y.data_ptr(), st), "mdno_nnconv_fwd")
        ms = _event_ms(conv, 10, warm=3)
        byts = Es * (C * C * 4 + 4) + (rows + 1) * 4 + 2 * rows * C * 4
        conv_slice = {"bound": "hbm", "kernel": "nnconv64_row_kernel", "conv_mode": "materialized", "rows": rows, "edges": Es,
                      "avg_launch_ms": ms, "algorithmic_bytes_per_launch": byts, "achieved": byts / ms / 1e6, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": byts / ms / 1e6 / HBM_PEAK_GBS,
                      "frac_of_measured_copy_peak": byts / ms / 1e6 / HBM_COPY_GBS,
                      "traffic": profiled_traffic("nnconv64_row_kernel", N, 1, "materialized", "slice")}
        del x, w_e, y
    except Exception as e:     # noqa: BLE001 — recorded in the line
        conv_slice = {"error": f"{type(e).__name__}: {e}"}
    del g
    torch.cuda.empty_cache()
    sd = near_identity_state_dict(64, KW, seed=0, kernel_gain=1e-3, feature_gain=0.1)
    model = KernelNN(64, KW, 6, 6, 7, 3, 20, 4)
    model.load_state_dict(sd)
    model.eval().to(dev)
    model.conv_mode = "factored"
    eng = RolloutEngine(model, 1, N, W, cutoff, max_steps=steps + 1, edge_cap=int(E * 1.05), device=dev)
    ws_gib = eng.workspace.numel() / 2**30
    eng.reset(torch.from_numpy(win), aa)
    eng.step(1)
    eng.synchronize()
    eng.attach_timer(steps * 6000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(steps)
    eng.stream.synchronize()
    dt = time.perf_counter() - t0
    tm = eng.read_timer()
    eng.detach_timer()
    eng.synchronize()
    e_mean = float(eng.edges_per_step[1:1 + steps].double().mean().item())
    ks = {k: {"ms_per_step": ms / steps, "launches": int(n)} for k, (ms, n) in tm.items() if n}
    app_s = ks["nnconv"]["ms_per_step"] * 1e-3 / 12            # all launches of one conv application
    alg = e_mean * KW * 4 + N * C * KW * 4 + e_mean * 4 + (N + 1) * 4          # K1 of csrc/moment.hip: H in, S out, src, row_ptr
    out = {"atoms": N, "cutoff_A": cutoff, "edges": E, "mean_degree": E / N, "max_degree": deg_max, "steps": steps,
           "ms_per_step": dt / steps * 1e3, "frames_per_s": steps / dt, "workspace_GiB": ws_gib, "conv_mode": eng.conv_mode,
           "kernels_ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in ks.items()},
           "roofline": {"bound": "hbm", "kernel": "moment_kernel", "achieved": alg / app_s / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / app_s / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_application": alg, "ms_per_application": app_s * 1e3,
                        "traffic": profiled_traffic("moment_kernel", N, 1, "factored", "split_f16")},
           "conv_materialized_slice": conv_slice, "fallbacks": eng.fallback_counts(),
           "launch": "plain launches (event timer attached)"}
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return out


def leg_cfg2_1000_steps(dev, a, steps=1000):
    """BASELINE configs[1] as worded — "BBA 1000-step autoregressive rollout, fp32, 1xMI355X": the headline's
    workload (same start window, weights, capacity and launch mode) run for 1,000 steps in one call, after a
    10-step warm-up; the window slides through the trajectory buffer the whole time."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine, default_edge_cap
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    N, W, warm = a.atoms, a.window, 10
    model = KernelNN(a.width, a.kernel_width, a.depth, 6, 7, 3, 20, 4)
    model.load_state_dict(near_identity_state_dict(a.width, a.kernel_width, seed=0, kernel_gain=1e-3, feature_gain=0.1))
    model.eval().to(dev)
    model.gemm_mode, model.conv_mode = a.gemm_mode, a.conv_mode
    win = torch.from_numpy(syn.jitter_window(syn.box_frame(N, seed=1), W, seed=1))
    aa = torch.from_numpy(syn.amino_acids(N, seed=1))
    eng = RolloutEngine(model, 1, N, W, a.threshold, max_steps=warm + steps, edge_cap=default_edge_cap(1, N, a.threshold),
                        device=dev, use_graph=not a.no_graph)
    eng.reset(win, aa)
    eng.step(warm)
    eng.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(steps)
    eng.stream.synchronize()
    dt = time.perf_counter() - t0
    eng.synchronize()                      # raises on overflow / bad input
    fr = eng.frames()
    e = eng.edges_per_step[warm:warm + steps]
    out = {"steps": steps, "warmup": warm, "atoms": N, "frames_per_s": steps / dt, "ms_per_step": dt / steps * 1e3,
           "seconds": dt, "finite": bool(torch.isfinite(fr).all()), "conv_mode": eng.conv_mode,
           "edges_first_last": [int(e[0].item()), int(e[-1].item())], "edges_min_max": [int(e.min().item()), int(e.max().item())],
           "max_displacement_from_start_A": float((fr[-1, 0] - fr[0, 0]).abs().max().item()),
           "launch": "plain" if a.no_graph else "hipGraph replay", "fallbacks": eng.fallback_counts()}
    eng.close()
    if not out["finite"]:
        raise RuntimeError("cfg2 1000-step rollout produced a non-finite frame")
    return out


def leg_shape_a(dev, a):
    """SURVEY.md §8 shape A — the reference's actual BBA (N=28 C-alpha, bba_analysis.ipynb:1034): the in-tree model
    with 1 and with 64 members, and the notebook-era model (window 1, conv1 only, kernel_width 512) whose rollout
    the notebook timed at 80.56 it/s on unknown hardware (nb:370)."""
    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, KernelNNNotebook
    from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
    N = 28
    frame0 = syn.chain_frame(N, seed=1)
    aa = torch.from_numpy(syn.amino_acids(N, seed=1))
    out = {"atoms": N, "reference_notebook_it_per_s": 80.56,
           "reference_note": "bba_analysis.ipynb:370 — notebook-era model, device unknown, incl. host graph rebuild"}

    def run(model, M, W, steps, warm, reps=2):
        base = syn.jitter_window(frame0, W, seed=1)
        wins = np.stack([syn.ensemble_windows(base, 1, sigma=0.1, seed0=100 + m)[0] if M > 1 else base for m in range(M)], axis=1)
        eng = RolloutEngine(model, M, N, W, a.threshold, max_steps=warm + reps * steps, device=dev)
        eng.reset(torch.from_numpy(wins), aa)
        eng.step(warm)
        eng.synchronize()
        torch.cuda.synchronize()
        # a step is ~0.1 ms of ~16 tiny kernels: the timed region is repeated and the faster repetition reported
        # (one in three runs of a single 50 ms region caught the GPU still ramping its clocks: half the rate)
        dts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            eng.step(steps)
            eng.stream.synchronize()
            dts.append(time.perf_counter() - t0)
        eng.synchronize()
        dt = min(dts)
        r = {"members": M, "steps": steps, "frames_per_s": steps * M / dt, "ms_per_step": dt / steps * 1e3, "conv_mode": eng.conv_mode,
             "repetitions_ms_per_step": [round(t / steps * 1e3, 4) for t in dts], "fallbacks": eng.fallback_counts(),
             "edges_per_member": float(eng.edges_per_step[warm:warm + steps].double().mean().item()) / M}
        eng.close()
        return r

    sd = near_identity_state_dict(64, 1024, seed=0, kernel_gain=1e-3, feature_gain=0.1)
    model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
    model.load_state_dict(sd)
    model.eval().to(dev)
    model.gemm_mode = a.gemm_mode
    out["intree_1_member"] = run(model, 1, 10, 2000, 1000)
    out["intree_64_members"] = run(model, 64, 10, 100, 10)
    sdn = {k: v for k, v in near_identity_state_dict(64, 512, seed=0, kernel_gain=1e-3, feature_gain=0.1).items()
           if not k.startswith(("lstm", "conv2"))}
    nb = KernelNNNotebook(64, 512, 6, 6, 7, 3, 20, 4)
    nb.load_state_dict(sdn)
    nb.eval().to(dev)
    nb.gemm_mode = a.gemm_mode
    out["notebook_era_k512_window1"] = run(nb, 1, 1, 2000, 1000)
    out["notebook_era_k512_window1"]["vs_reference_notebook"] = out["notebook_era_k512_window1"]["frames_per_s"] / 80.56
    return out


def leg_rank_share(a, ensemble_leg, members=8, timeout_s=240):
    """What ONE rank of the 8-GPU run of BASELINE configs[2] does — 8 of the 64 members, K steps, then the trajectory
    all-gather — run as a child `python bench.py --gpus 1 --total-members 8` with MDNO_BENCH_FORCE_DIST=1: a
    world-size-1 nccl (= RCCL) process group on this GPU, so the collective path of the timed region (barrier,
    all_gather_into_tensor, all_reduce MAX) executes through librccl.  A failure is recorded, not fatal."""
    env = dict(os.environ, MDNO_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MDNO_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, str(Path(__file__).resolve()), "--gpus", "1", "--total-members", str(members),
           "--steps", str(a.steps), "--warmup", str(a.warmup), "--atoms", str(a.atoms), "--window", str(a.window),
           "--kernel-width", str(a.kernel_width), "--depth", str(a.depth), "--gemm-mode", a.gemm_mode,
           "--conv-mode", a.conv_mode, "--skip-roofline", "--skip-cpu-baseline", "--single-mode"]
    if a.no_graph:
        cmd.append("--no-graph")
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"child did not finish in {timeout_s} s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"child exit code {r.returncode}", "stderr_tail": r.stderr[-1500:]}
    d = json.loads(lines[-1])
    if "value" not in d:
        return {"error": d.get("error", "no value on the child's line"), "stderr_tail": r.stderr[-1500:]}
    mg = d.get("multi_gpu_timing") or {}
    out = {"members": members, "steps": d["steps"], "warmup": d["warmup"], "frames_per_s": d["value"], "ms_per_step": d["ms_per_step"],
           "member_groups": d["config"]["member_groups_this_rank"], "conv_mode": d["config"]["conv_mode"],
           "fallbacks": {k: v for k, v in (d.get("fallbacks") or {}).items() if k != "what"},
           "collective": {"backend": mg.get("backend"), "world_size": mg.get("world_size"),
                          "init_process_group_s": mg.get("init_process_group_s"), "first_gather_s": mg.get("first_gather_s"),
                          "timed_gather_ms": (mg.get("per_rank_gather_ms") or [None])[0],
                          "gathered_bytes": mg.get("gathered_bytes_per_rank"),
                          "what": "barrier + all_gather_into_tensor + all_reduce(MAX) of the timed region through a "
                                  "world-size-1 nccl (RCCL) group in a child process"},
           "child_seconds": time.perf_counter() - t0,
           "note": "one rank's share of BASELINE configs[2] at 8 GPUs (8 of 64 members), inside the same timed region "
                   "an N > 1 rank runs"}
    if ensemble_leg and ensemble_leg.get("frames_per_s"):
        out["baseline_1gpu_same_workload"] = ensemble_leg["frames_per_s"]
        out["projected_speedup_8gpu"] = 8.0 * d["value"] / ensemble_leg["frames_per_s"]
        out["projected_value_8gpu"] = 8.0 * d["value"]
    note(f"per-rank share: {d['value']:.1f} frames/s with {members} members"
         + (f", projected 8-GPU speed-up {out['projected_speedup_8gpu']:.2f}x" if "projected_speedup_8gpu" in out else ""))
    return out


# ----------------------------------------------------------------------------------------------- worker
def profiled_entry(kernel: str, atoms: int, members: int, conv_mode: str, gemm_mode: str) -> dict:
    """The entry of profiles/roofline_traffic.json for this kernel and configuration ({} if none): PMC HBM bytes per
    launch, the rocprofv3 trace's average duration, the conv application's summed traffic — only for the configuration
    the counters were collected on (anything else would print a number that belongs to another run)."""
    tf = REPO / "profiles" / "roofline_traffic.json"
    if not tf.exists():
        return {}
    try:
        for ent in json.loads(tf.read_text()).get("configs", []):
            if (ent.get("kernel") == kernel and ent.get("atoms") == atoms and ent.get("members") == members
                    and ent.get("conv_mode") == conv_mode and ent.get("gemm_mode") == gemm_mode):
                return ent
    except Exception:
        return {}
    return {}


def profiled_traffic(kernel: str, atoms: int, members: int, conv_mode: str, gemm_mode: str):
    return profiled_entry(kernel, atoms, members, conv_mode, gemm_mode).get("hbm_bytes_per_launch")


def worker(a):
    # Before the first HIP call of this process (HSA reads it when the runtime opens): the host driver of this pool
    # only supports dmabuf IPC, and RCCL's / torch's cross-process device-memory handles fail with
    # `hipIpcGetMemHandle: invalid argument` without it (the environment's own instructions export it; a launcher
    # that does not pass it on — torch.distributed.run under a scrubbed environment — must not lose it).
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)     # rehearsal on a 1-GPU box: ranks share the card
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if os.environ.get("MDNO_BENCH_FAIL_RANK") == str(rank):  # injected failure (tests: a rank that dies at start-up)
        raise RuntimeError(f"MDNO_BENCH_FAIL_RANK={rank}: injected failure after set_device")
    backend = os.environ.get("MDNO_BENCH_BACKEND", "nccl")   # "gloo" only to rehearse N>1 on one GPU
    # MDNO_BENCH_FORCE_DIST=1 on a one-rank run: a world-size-1 process group, so that the timed region's barrier,
    # all-gather and max-over-ranks run through the backend (RCCL: communicator init + all_gather_into_tensor through
    # librccl on this GPU) exactly as a rank of an N > 1 run issues them
    force_dist = world == 1 and os.environ.get("MDNO_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    t_init = None
    if use_dist:
        from datetime import timedelta
        # a rank that never arrives (or dies in a collective) fails the others after INIT_TIMEOUT_S, not after the
        # backends' 10-30 minute defaults
        tmo = timedelta(seconds=float(os.environ.get("MDNO_BENCH_INIT_TIMEOUT_S", INIT_TIMEOUT_S)))
        kw = {}
        if force_dist and "MASTER_PORT" not in os.environ:   # no launcher around a forced one-rank group
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                kw["init_method"] = f"tcp://127.0.0.1:{sk.getsockname()[1]}"
        t0 = time.perf_counter()
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo, **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo, **kw)
        t_init = time.perf_counter() - t0

    from molecular_dynamics_neural_operator_amd import synthetic as syn
    from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, KernelNNNotebook
    from molecular_dynamics_neural_operator_amd.rollout import (GroupedRolloutEngine, RolloutEngine, default_edge_cap,
                                                                 gather_trajectories, shard_members)
    from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict

    N, W = a.atoms, a.window
    if a.members_per_gpu is not None and a.total_members is not None:
        raise SystemExit("give --total-members (strong scaling) or --members-per-gpu (weak scaling), not both")
    if a.members_per_gpu is not None:
        total_members, scaling = a.members_per_gpu * world, "weak"
    elif a.total_members is not None:
        total_members, scaling = a.total_members, "strong"
    else:
        # (the default one-GPU line is configs[1], one member — not a point of either series; the N > 1 lines are the
        # 64-member ensemble split over the ranks: fixed total work.  `scaling_note` on the line says so.)
        total_members, scaling = (1, "strong") if world == 1 else (ENSEMBLE_MEMBERS, "strong")
    my_members = shard_members(total_members, rank, world)
    M = len(my_members)
    if M == 0:
        raise SystemExit(f"rank {rank}: no members ({total_members} members over {world} ranks)")
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

    def member_windows(ids, perturbed):
        """[W,len(ids),N,3]: member m = base window + N(0, 0.1^2), seed 100+m; a
        one-member run uses the base window itself (cfg2)."""
        return np.stack([syn.ensemble_windows(base, 1, sigma=0.1, seed0=100 + m)[0] if perturbed else base
                         for m in ids], axis=1)

    wins = member_windows(my_members, perturbed=total_members > 1)
    aa = torch.from_numpy(syn.amino_acids(N, seed=1))
    max_steps = a.warmup + a.steps + (0 if a.skip_roofline else a.steps)
    cap = default_edge_cap(M, N, a.threshold)
    def groups_for(members):
        # four members per group (8 members: 2 groups 582 vs 4 groups 577 frames/s; 64 members: 4 / 8 / 16 groups
        # 577 / 589 / 594 on one box)
        return a.member_groups if a.member_groups > 0 else (max(2, min(16, members // 4)) if members >= 2 and N >= 256 else 1)

    def make_engine(members, steps_cap, edge_cap):
        """one engine, or the members as groups on concurrent streams (same frames, another schedule: EXPERIMENTS §0.2b)"""
        g = min(groups_for(members), members)
        if g > 1:
            return GroupedRolloutEngine(model, members, N, W, a.threshold, max_steps=steps_cap, edge_cap=edge_cap, device=dev,
                                        use_graph=not a.no_graph, groups=g)
        return RolloutEngine(model, members, N, W, a.threshold, max_steps=steps_cap, edge_cap=edge_cap, device=dev,
                             use_graph=not a.no_graph)

    def wait(engine):       # the engine's stream(s) drained, no status read
        engine.wait() if isinstance(engine, GroupedRolloutEngine) else engine.stream.synchronize()

    def ws_bytes(engine):
        return engine.workspace_bytes if isinstance(engine, GroupedRolloutEngine) else engine.workspace.numel()

    eng = make_engine(M, max_steps, cap)
    n_groups = len(eng.engines) if isinstance(eng, GroupedRolloutEngine) else 1
    eng.reset(torch.from_numpy(wins), aa)
    mode = eng.conv_mode            # what "auto" resolved to at this edge capacity
    if rank == 0:
        note(f"rank 0: {M} of {total_members} members in {n_groups} group(s), conv_mode {mode}, workspace {ws_bytes(eng) / 2**30:.1f} GiB")

    # ---- warm-up (untimed): the step graph was captured in reset()
    eng.step(a.warmup)
    eng.synchronize()
    # first produced frame of member 0 (kept for the cpu_baseline leg's parity figure)
    first_frame_gpu = eng.traj[W, 0].detach().cpu().numpy() if (a.warmup > 0 and total_members == 1) else None
    m_max = -(-total_members // world)
    t_first_gather = None
    if use_dist:     # the collective of the timed region, once, untimed, on the same path (the groups' frames concatenated,
        # gathered, re-ordered): communicator set-up, buffers, and the first-use load of every kernel involved — a first
        # torch.cat alone is ~10 ms
        t0 = time.perf_counter()
        gather_trajectories(eng.produced(0, a.steps), total_members)
        torch.cuda.synchronize()
        t_first_gather = time.perf_counter() - t0

    # ---- timed region: exactly K steps + trajectory collection
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(a.steps)
    wait(eng)
    t_steps = time.perf_counter() - t0                                             # this rank's K steps alone
    produced = eng.produced(a.warmup, a.steps)                                     # [K,M,N,3]
    if use_dist:
        full = gather_trajectories(produced, total_members)
    else:
        full = produced
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - t0 - t_steps                                  # incl. waiting for the slowest rank
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    mg_timing = None
    if use_dist:     # per-rank step time and the collective, separately (outside the timed region)
        tt = torch.tensor([t_steps, t_gather], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        mg_timing = {"per_rank_steps_ms": [float(x[0]) * 1e3 for x in allt],
                     "per_rank_gather_ms": [float(x[1]) * 1e3 for x in allt],
                     "note": "gather_ms of a rank includes its wait for the slowest rank's steps; the collective "
                             "itself is the smallest entry", "gathered_bytes_per_rank": int(produced.numel() * 4),
                     "backend": backend, "world_size": world, "init_process_group_s": t_init,
                     "first_gather_s": t_first_gather}
    eng.synchronize()   # raises on edge overflow / bad input
    assert full.shape == (a.steps, total_members, N, 3) and bool(torch.isfinite(full).all())
    # which path the split_f16 products of the TIMED steps took (device counters, zeroed per step() call): all zero = two
    # fp16 planes everywhere, the path the headline is quoted on; anything else = pieces redone on bf16 planes
    fallbacks = eng.fallback_counts()
    eps = eng.edges_per_step[a.warmup:a.warmup + a.steps].double()
    e_mean = float(eps.mean().item())
    frames = a.steps * total_members
    value = frames / elapsed
    if rank == 0:
        note(f"timed region: {value:.1f} frames/s ({elapsed / a.steps * 1e3:.3f} ms/step)")

    # ---- roofline leg: K more steps of the SAME rollout, plain launches bracketed by HIP events on
    # the launch stream (events cannot sit inside a hipGraph replay).  Rank 0 only.
    roofs = {}
    kernels = {}
    other_mode = None
    eng_r = eng.engines[0] if isinstance(eng, GroupedRolloutEngine) else eng      # the engine the kernel timers sit on
    M_r = eng_r.M
    R, C, KW = M_r * N, a.width, a.kernel_width

    def timed_leg(engine, first_step, members):
        engine.attach_timer(a.steps * (6 * a.depth * max(1, members) + 20))
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
        prof = profiled_entry("nnconv64_row_kernel", N, M_r, "materialized", a.gemm_mode)
        return {"bound": "hbm", "kernel": "nnconv64_row_kernel", "conv_mode": "materialized", "achieved": alg / avg_s / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / avg_s / 1e9 / HBM_PEAK_GBS,
                "frac_of_measured_copy_peak": alg / avg_s / 1e9 / HBM_COPY_GBS,
                "traffic": prof.get("hbm_bytes_per_launch"),
                "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_s * 1e3, "avg_launch_ms_events": avg_s * 1e3,
                "avg_launch_us_rocprof": prof.get("rocprof_avg_us"), "edges_per_launch": e, "rows_per_launch": R}

    def gemm_roofline(ks, e, which, n_out):
        step_s = ks[which]["ms_per_step"] * 1e-3     # all launches of a step (capacity-sized chunks past E exit at once)
        flops32 = 2.0 * e * KW * n_out                # fp32-equivalent work
        if a.gemm_mode == "f32":
            return {"bound": "mfma", "kernel": "gemm_tn_mfma_kernel", "achieved": flops32 / step_s / 1e12,
                    "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops32 / step_s / 1e12 / MFMA_F32_PEAK_TFLOPS,
                    "ms_per_step": step_s * 1e3}
        f16 = a.gemm_mode == "split_f16"              # both edge-MLP GEMMs run on two fp16 planes in that mode
        products = 3.0 if f16 else 6.0                # plane products executed per fp32 product
        ach = products * flops32 / step_s / 1e12
        return {"bound": "mfma", "kernel": "gemm_split_f16_kernel" if f16 else "gemm_split_bf16_kernel", "achieved": ach,
                "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
                "fp32_equivalent_tflops": flops32 / step_s / 1e12, "ms_per_step": step_s * 1e3,
                "frac_of_measured_sustained_rate": ach / MFMA_16BIT_SUSTAINED_TFLOPS,
                "note": f"executed 16-bit MFMA flops ({products:.0f} plane products per fp32 product) vs the dense "
                        "bf16/fp16 peak; frac_of_measured_sustained_rate: vs the 1.48 PF the chip sustains on fp16 MFMAs at "
                        "this kernel's LDS-DMA intensity (scripts/micro/mfma_sustained.hip)"}

    def moment_roofline(ks, e):   # factored path (csrc/moment.hip): K1 over all launches of one conv application
        launches_per_app = max(1, round(ks["nnconv"]["launches"] / (a.steps * 2 * a.depth)))
        avg_s = ks["nnconv"]["avg_ms"] * 1e-3 * launches_per_app      # one application over all R rows
        split = a.gemm_mode != "f32"
        flops = 2.0 * e * KW * C                                                # fp32-equivalent
        # K1: S_t = sum_{e->t} x_src (x) h_e — H read once, S written once, the source index per edge, row_ptr (the
        # neighbours' 256-B feature rows are gathered from L2)
        name = "moment_kernel" if split else "moment_f32_kernel"
        alg = e * KW * 4 + R * C * KW * 4 + e * 4 + (R + 1) * 4
        # matrix-pipe work as executed: 3 fp16 (split_f16) or 6 bf16 (split_bf16) plane products per fp32 product, or the
        # fp32 MFMA itself
        products = {"split_f16": 3.0, "split_bf16": 6.0}.get(a.gemm_mode, 1.0)
        mfma_exec, mfma_peak = (products * flops, MFMA_BF16_PEAK_TFLOPS) if split else (flops, MFMA_F32_PEAK_TFLOPS)
        t_hbm, t_mfma = alg / (HBM_PEAK_GBS * 1e9), mfma_exec / (mfma_peak * 1e12)
        prof = profiled_entry(name, N, M_r, "factored", a.gemm_mode)
        r = {"kernel": name, "conv_mode": "factored", "avg_launch_ms": avg_s * 1e3, "avg_launch_ms_events": avg_s * 1e3,
             "avg_launch_us_rocprof": prof.get("rocprof_avg_us"), "launches_per_application": launches_per_app,
             "algorithmic_bytes_per_launch": alg, "flops_per_launch": flops,
             "traffic": prof.get("hbm_bytes_per_launch"),
             "hbm_GBps": alg / avg_s / 1e9, "hbm_frac": alg / avg_s / 1e9 / HBM_PEAK_GBS,
             "frac_of_measured_copy_peak": alg / avg_s / 1e9 / HBM_COPY_GBS,
             "mfma_TFLOPs": mfma_exec / avg_s / 1e12, "mfma_frac": mfma_exec / avg_s / 1e12 / mfma_peak}
        if t_hbm >= t_mfma:
            r.update(bound="hbm", achieved=r["hbm_GBps"], peak=HBM_PEAK_GBS, unit="GB/s", frac=r["hbm_frac"])
        else:
            r.update(bound="mfma", achieved=r["mfma_TFLOPs"], peak=mfma_peak, unit="TFLOP/s", frac=r["mfma_frac"])
        # the whole conv application (K1 + K2 + K3) against what it MUST move: H in, W3R in (+ the B3 rows), x in, y out.
        # S (R*64*k*4 out of K1, in again to K2) and the K-slice partials are intermediates of this formulation.
        k2 = ks.get("factored_y", {"ms_per_step": 0.0})["ms_per_step"]
        k3 = ks.get("nnconv_combine", {"ms_per_step": 0.0})["ms_per_step"]
        app_s = (ks["nnconv"]["ms_per_step"] + k2 + k3) * 1e-3 / (2 * a.depth)
        compulsory = e * KW * 4 + (C * KW + C) * C * 4 + 2 * R * C * 4 + e * 4 + (R + 1) * 4
        moved = alg + (R * C * KW * 4 + (C * KW + C) * C * 4) + 2 * 128 * R * C * 4 + R * C * 4      # + K2 in, partials out/in, y
        r["per_application"] = {
            "kernels": {"split_f16": "moment_kernel<true> (K1) + project_f16_kernel (K2) + finish_kernel (K3)",
                        "split_bf16": "moment_kernel<false> (K1) + project_kernel<256> (K2) + finish_kernel (K3)"}.get(
                            a.gemm_mode, "moment_f32_kernel (K1) + project_f32_kernel (K2) + finish_kernel (K3)"),
            "ms_events": app_s * 1e3, "k1_k2_k3_ms": [ks["nnconv"]["ms_per_step"] / (2 * a.depth), k2 / (2 * a.depth), k3 / (2 * a.depth)],
            "compulsory_bytes": compulsory, "algorithmic_bytes_moved_by_the_three_kernels": moved,
            "traffic": prof.get("application_hbm_bytes"),
            "achieved": compulsory / app_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": compulsory / app_s / 1e9 / HBM_PEAK_GBS,
            "frac_of_bytes_moved": moved / app_s / 1e9 / HBM_PEAK_GBS}
        return r

    # rank 0 only: the other ranks idle at the final barrier through the measurement legs
    if not a.skip_roofline and rank == 0:
        kernels, e2 = timed_leg(eng_r, a.warmup + a.steps, M_r)
        if mode == "materialized":
            roofs["conv_materialized"] = conv_roofline(kernels, e2)
            roofs["edge_mlp_last_gemm"] = gemm_roofline(kernels, e2, "edge_mlp_gemm2", C * C)
        else:
            roofs["conv_factored_moment"] = moment_roofline(kernels, e2)
        roofs["edge_mlp_hidden_gemm"] = gemm_roofline(kernels, e2, "edge_mlp_gemm1", KW)
        # ---- the other conv formulation on the same start window: frames/s and, for the materialised
        # one, the HBM roofline of the gather/matvec/scatter kernel BASELINE.json's target is stated on
        if a.variant == "intree" and not a.single_mode and world == 1 and M <= 8:
            om = "materialized" if mode == "factored" else "factored"
            model.conv_mode = om
            # (one engine with the members of the profiled group: the rooflines below are per launch of that size)
            eng2 = RolloutEngine(model, M_r, N, W, a.threshold, max_steps=a.warmup + 2 * a.steps,
                                 edge_cap=default_edge_cap(M_r, N, a.threshold), device=dev, use_graph=not a.no_graph)
            eng2.reset(torch.from_numpy(np.ascontiguousarray(wins[:, :M_r])), aa)
            eng2.step(a.warmup)
            eng2.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng2.step(a.steps)
            eng2.stream.synchronize()
            dt = time.perf_counter() - t0
            k2, e3 = timed_leg(eng2, a.warmup + a.steps, M_r)
            note(f"comparison leg ({om}, {M_r} member(s)): {a.steps * M_r / dt:.1f} frames/s")
            other_mode = {"conv_mode": om, "members": M_r, "frames_per_s_this_rank": a.steps * M_r / dt, "ms_per_step": dt / a.steps * 1e3,
                          "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in k2.items()}}
            if om == "materialized":
                roofs["conv_materialized"] = conv_roofline(k2, e3)
                roofs["edge_mlp_last_gemm"] = gemm_roofline(k2, e3, "edge_mlp_gemm2", C * C)
            else:
                roofs["conv_factored_moment"] = moment_roofline(k2, e3)
            eng2.close()
            del eng2
            model.conv_mode = a.conv_mode
    # "roofline" = the dominant kernel of the TIMED path
    dominant = None
    if kernels:
        have = {"nnconv": roofs.get("conv_materialized" if mode == "materialized" else "conv_factored_moment"),
                "edge_mlp_gemm2": roofs.get("edge_mlp_last_gemm"), "edge_mlp_gemm1": roofs.get("edge_mlp_hidden_gemm")}
        have = {k: v for k, v in have.items() if v is not None and k in kernels}
        if have:      # (tiny shapes: a latency-bound helper can top the list; the roofline is quoted on a roofline-bound kernel)
            dominant = have[max(have, key=lambda k: kernels[k]["ms_per_step"])]

    # ---- cfg3's like-for-like single-GPU figure: the SAME 64-member ensemble the N > 1 runs shard, on
    # this one GPU (default N=1 run only; `--gpus 1 --total-members 64` makes it the headline instead)
    ensemble_leg = None
    default_single = (world == 1 and a.total_members is None and a.members_per_gpu is None
                      and a.variant == "intree" and not a.chain)
    if default_single and not a.skip_ensemble_leg:
        eng.close()
        del eng
        torch.cuda.empty_cache()
        me = ENSEMBLE_MEMBERS
        es, ew = max(2, min(a.steps, 10)), 2
        we = member_windows(list(range(me)), perturbed=True)     # seeds 100..163
        note(f"ensemble leg: {me} members x {N} atoms")
        enge = make_engine(me, ew + es, default_edge_cap(me, N, a.threshold))
        enge.reset(torch.from_numpy(we), aa)
        enge.step(ew)
        enge.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enge.step(es)
        wait(enge)
        dte = time.perf_counter() - t0
        enge.synchronize()
        note(f"ensemble leg: {me} members on this GPU, {es * me / dte:.1f} frames/s")
        ensemble_leg = {"members": me, "steps": es, "warmup": ew, "frames_per_s": es * me / dte, "fallbacks": enge.fallback_counts(),
                        "ms_per_step": dte / es * 1e3, "ms_per_member_step": dte / es / me * 1e3,
                        "conv_mode": enge.conv_mode, "member_groups": len(enge.engines) if isinstance(enge, GroupedRolloutEngine) else 1,
                        "note": "BASELINE configs[2] (64-member ensemble) on ONE GPU: the 1-GPU point of the "
                                "strong-scaling series that `--gpus N` (N > 1) runs"}
        enge.close()
        del enge
        eng = None

    # ---- the other BASELINE configurations on the same clock (default N=1 run only; each leg is bounded and a
    # failure in one is recorded, not fatal to the headline)
    config_legs = {}
    if default_single and not a.skip_config_legs:
        if eng is not None:
            eng.close()
            eng = None
        torch.cuda.empty_cache()
        for name, fn in (("cfg2_1000_steps", lambda: leg_cfg2_1000_steps(dev, a)),
                         ("shape_A", lambda: leg_shape_a(dev, a)), ("cfg4_training", lambda: leg_cfg4_training(dev)),
                         ("cfg5_shape_c", lambda: leg_cfg5_shape_c(dev))):
            t0 = time.perf_counter()
            note(f"{name} leg")
            try:
                config_legs[name] = fn()
                config_legs[name]["leg_seconds"] = time.perf_counter() - t0
            except Exception as e:     # noqa: BLE001 — recorded in the line
                config_legs[name] = {"error": f"{type(e).__name__}: {e}"}
                note(f"{name} leg FAILED: {config_legs[name]['error']}")
            torch.cuda.empty_cache()

    # ---- one rank's share of the 8-GPU run of configs[2] (8 of the 64 members), as that rank runs it: a fresh child
    # of this file with a world-size-1 nccl group (MDNO_BENCH_FORCE_DIST=1), so that barrier, all-gather and
    # max-over-ranks of the timed region go through RCCL on this GPU.  8 x this / the 64-member single-GPU figure =
    # what the 8-GPU point should show if nothing but the shard size changes.
    share_leg = None
    if default_single and not a.skip_ensemble_leg and not a.skip_rank_share_leg:
        if eng is not None:
            eng.close()
            eng = None
        torch.cuda.empty_cache()
        note("per-rank share leg: 8 members through a world-size-1 RCCL group (child process)")
        share_leg = leg_rank_share(a, ensemble_leg)

    cpu = None
    if rank == 0 and world == 1 and not a.skip_cpu_baseline and a.variant == "intree":
        cpu = cpu_baseline(sd, a.depth, base, aa, a.threshold, a.cpu_budget_s, first_frame_gpu)
        if cpu.get("parity_first_frame"):
            pf = cpu["parity_first_frame"]
            note(f"parity, first frame vs the oracle: rel L2 {pf['rel_l2']:.2e} (of the displacement "
                 f"{pf['rel_l2_of_displacement']:.2e}), max |err| {pf['max_abs_A']:.2e} A")
            if not pf["rel_l2"] <= pf["tolerance_rel_l2"]:
                raise SystemExit(f"bench.py: first-frame parity {pf['rel_l2']:.3e} exceeds {pf['tolerance_rel_l2']:.0e}")

    if rank == 0:
        cfg = "configs[1]" if total_members == 1 else ("configs[2]" if total_members == ENSEMBLE_MEMBERS else "ensemble")
        line = {
            "metric": "rolled-out MD frames/sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "scaling_note": ("this line is BASELINE configs[1] (ONE member on one GPU), not a point of the N > 1 series; "
                             "the series splits the 64-member ensemble of configs[2] over the ranks (fixed total work = "
                             "strong scaling) and its 1-GPU point is `baseline_1gpu_same_workload`"
                             if world == 1 and total_members == 1 else
                             "fixed total work split over the ranks" if scaling == "strong" else "fixed work per rank"),
            "data": "synthetic (uniform-box frames, seeded; near-identity synthetic weights, see weights.py)",
            "config": {"workload": f"BBA all-atom stand-in N={N} r={a.threshold}A free-running autoregressive rollout, "
                                   f"{total_members}-member ensemble (BASELINE {cfg}), member m on rank m mod {world}",
                       "atoms": N, "window": W, "width": a.width, "kernel_width": a.kernel_width, "depth": a.depth,
                       "members_this_rank": M, "members_per_gpu_max": m_max, "total_members": total_members,
                       "member_groups_this_rank": n_groups, "members_in_the_profiled_group": M_r,
                       "mean_edges_per_member": e_mean / M,
                       "edges_first_last": [int(eps[0].item()), int(eps[-1].item())], "edge_cap": cap,
                       "parallelism": f"ensemble-sharded x{world}, no collective while stepping, one all-gather of "
                                      f"trajectories ({(backend + (' world 1, forced' if force_dist else '')) if use_dist else 'none'})",
                       "launch": "plain" if a.no_graph else "hipGraph replay", "edge_mlp_gemm": a.gemm_mode,
                       "variant": a.variant, "conv_mode": mode, "conv_mode_requested": a.conv_mode},
            "fallbacks": dict(fallbacks, what="pieces of the timed steps that left the two-fp16-plane path (this rank): K1 "
                              "workgroups rerun on bf16 planes, destinations with unscaled operands, edge-MLP products on "
                              "bf16 planes; expected all zero on the benchmark's weights"),
            "roofline": dominant, "rooflines": roofs, "other_conv_mode": other_mode,
            "ensemble64_single_gpu": ensemble_leg,
            # the like-for-like 1-GPU point of the N > 1 series (same 64-member workload): divide an N-GPU
            # `value` by THIS, not by the 1-member headline above
            "baseline_1gpu_same_workload": ({"workload": "BASELINE configs[2]: 64-member ensemble on one GPU",
                                             "value": ensemble_leg["frames_per_s"], "unit": "frames/s"}
                                            if ensemble_leg else None),
            "per_rank_share_8gpu": share_leg,
            "projected_speedup_8gpu": (share_leg or {}).get("projected_speedup_8gpu"),
            "cfg2_1000_steps": config_legs.get("cfg2_1000_steps"),
            "cfg4_training": config_legs.get("cfg4_training"), "cfg5_shape_c": config_legs.get("cfg5_shape_c"),
            "shape_A": config_legs.get("shape_A"),
            "cpu_baseline": cpu, "kernels": kernels, "multi_gpu_timing": mg_timing,
        }
        print(json.dumps(line), flush=True)
    if eng is not None:
        eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # no launcher around us: become one.  Nothing above has touched the GPU (torch is imported, HIP
        # is not initialised), and this process never does.
        sys.exit(launch_workers(a))
    try:
        worker(a)
    except BaseException as e:     # noqa: BLE001 — reported, then re-raised
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        rank = os.environ.get("RANK", "0")
        print(f"bench.py: rank {rank} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        if rank == "0":             # rank 0 owns stdout: the failure as one JSON line, then the non-zero exit
            print(json.dumps({"error": f"{type(e).__name__}: {e}", "rank": 0, "n_gpus": a.gpus}), flush=True)
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            # no interpreter teardown: a process group whose peers are still waiting can block its destructor, and
            # the launcher (ours or torch.distributed.run) is waiting for this exit code to stop the other ranks
            sys.stderr.flush()
            os._exit(e.code if isinstance(e, SystemExit) and isinstance(e.code, int) and e.code else 1)
        raise


if __name__ == "__main__":
    main()
