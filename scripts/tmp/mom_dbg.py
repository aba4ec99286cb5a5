import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from molecular_dynamics_neural_operator_amd import synthetic as syn
from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
from molecular_dynamics_neural_operator_amd.rollout import RolloutEngine, default_edge_cap
from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
dev = torch.device('cuda:0')
model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
model.load_state_dict(near_identity_state_dict(64, 1024, seed=0, kernel_gain=1e-3, feature_gain=0.1))
model.eval().to(dev)
def timed(M, N, W, cutoff, win, aa, cap, steps, maxdeg=0):
    eng = RolloutEngine(model, M, N, W, cutoff, max_steps=steps + 6, edge_cap=cap, device=dev, max_degree=maxdeg)
    eng.reset(win, aa); eng.step(3); eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.step(steps); eng.stream.synchronize(); dt = time.perf_counter() - t0
    eng.synchronize()
    eng.attach_timer(40000); eng.step(2); tm = eng.read_timer(); eng.detach_timer(); eng.synchronize()
    napp = 12 * 2
    print(f"M={M} N={N}: {dt/steps*1e3:.3f} ms/step {steps*M/dt:.1f} frames/s | per application (us):",
          {k: round(ms / napp * 1e3, 1) for k, (ms, n) in tm.items() if n and k in ("nnconv", "factored_y", "nnconv_combine")}, flush=True)
    eng.close()
N, W = 504, 10
base = syn.jitter_window(syn.box_frame(N, seed=1), W, seed=1)
aa = torch.from_numpy(syn.amino_acids(N, seed=1))
for M, steps in ((1, 40), (8, 20), (64, 5)):
    wins = np.stack([syn.ensemble_windows(base, 1, sigma=0.1, seed0=100 + m)[0] if M > 1 else base for m in range(M)], axis=1)
    timed(M, N, W, 8.0, torch.from_numpy(wins), aa, default_edge_cap(M, N, 8.0), steps)
if len(sys.argv) > 1 and sys.argv[1] == "C":
    N = 50000
    frame = syn.box_frame(N, seed=3)
    win = syn.jitter_window(frame, W, sigma=0.01, seed=3)
    aa = torch.from_numpy(syn.amino_acids(N, seed=3))
    model.conv_mode = "factored"
    timed(1, N, W, 10.0, torch.from_numpy(win), aa, int(18123866 * 1.05), 2, 640)
