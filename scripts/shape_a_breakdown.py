#!/usr/bin/env python3
"""Where a step of SURVEY.md §8 shape A (N = 28, the reference's BBA) goes: graph-replayed ms/step, then the same
steps as plain launches with the per-kernel event timer attached.  `--members M` for the ensemble shapes."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from molecular_dynamics_neural_operator_amd import synthetic as syn  # noqa: E402
from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN  # noqa: E402
from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine  # noqa: E402
from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--atoms", type=int, default=28)
ap.add_argument("--members", type=int, default=1)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--kernel-width", type=int, default=1024)
ap.add_argument("--depth", type=int, default=6)
ap.add_argument("--gemm-mode", default="split_f16")
ap.add_argument("--conv-mode", default="auto")
ap.add_argument("--edge-cap", type=int, default=None, help="edge capacity of the engine (default: M * N * N)")
a = ap.parse_args()

dev = torch.device("cuda:0")
N, M, W = a.atoms, a.members, 10
frame0 = syn.chain_frame(N, seed=1)
aa = torch.from_numpy(syn.amino_acids(N, seed=1))
base = syn.jitter_window(frame0, W, seed=1)
wins = np.stack([syn.ensemble_windows(base, 1, sigma=0.1, seed0=100 + m)[0] if M > 1 else base for m in range(M)], axis=1)
model = KernelNN(64, a.kernel_width, a.depth, 6, 7, 3, 20, 4)
model.load_state_dict(near_identity_state_dict(64, a.kernel_width, seed=0, kernel_gain=1e-3, feature_gain=0.1))
model.eval().to(dev)
model.gemm_mode, model.conv_mode = a.gemm_mode, a.conv_mode

warm = 20
eng = RolloutEngine(model, M, N, W, 8.0, max_steps=warm + 2 * a.steps, device=dev, edge_cap=a.edge_cap)
eng.reset(torch.from_numpy(wins), aa)
eng.step(warm)
eng.synchronize()
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.step(a.steps)
eng.stream.synchronize()
dt = time.perf_counter() - t0
eng.synchronize()
out = {"atoms": N, "members": M, "conv_mode": eng.conv_mode, "graph_ms_per_step": dt / a.steps * 1e3,
       "frames_per_s": a.steps * M / dt,
       "edges_per_member": float(eng.edges_per_step[warm:warm + a.steps].double().mean().item()) / M}
eng.attach_timer(a.steps * (6 * a.depth * max(1, M) + 20))
eng.step(a.steps)
tm = eng.read_timer()
eng.detach_timer()
eng.synchronize()
out["kernels"] = {k: {"avg_us": round(ms / n * 1e3, 2), "launches_per_step": n / a.steps, "us_per_step": round(ms / a.steps * 1e3, 2)}
                  for k, (ms, n) in tm.items() if n}
out["sum_us_per_step"] = round(sum(v["us_per_step"] for v in out["kernels"].values()), 1)
eng.close()
print(json.dumps(out, indent=1))
