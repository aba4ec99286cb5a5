"""Per-kernel time of the FIRST rollout step after a reset (cfg2 shape), repeated — for experimental library builds
whose results are garbage on purpose (MDNO_LIB=scripts/micro/exp/...so): only the first step's graph is the real one."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from molecular_dynamics_neural_operator_amd import synthetic as syn  # noqa: E402
from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN  # noqa: E402
from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine, default_edge_cap  # noqa: E402
from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict  # noqa: E402

dev = torch.device("cuda:0")
N, W, reps = 504, 10, 12
model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
model.load_state_dict(near_identity_state_dict(64, 1024, seed=0, kernel_gain=1e-3, feature_gain=0.1))
model.eval().to(dev)
model.conv_mode = "factored"
win = torch.from_numpy(syn.jitter_window(syn.box_frame(N, seed=1), W, seed=1))
aa = torch.from_numpy(syn.amino_acids(N, seed=1))
eng = RolloutEngine(model, 1, N, W, 8.0, max_steps=4, edge_cap=default_edge_cap(1, N, 8.0), device=dev, use_graph=False)
tot = {}
for r in range(reps + 2):
    eng.reset(win, aa)
    eng.attach_timer(512)
    eng.step(1)
    tm = eng.read_timer()
    eng.detach_timer()
    eng.stream.synchronize()
    if r >= 2:
        for k, (ms, n) in tm.items():
            if n:
                tot[k] = tot.get(k, 0.0) + ms
print({k: round(v / reps * 1e3, 1) for k, v in tot.items()}, "us per step, edges", int(eng.edges_per_step[0]))
