import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from molecular_dynamics_neural_operator_amd import synthetic as syn
from molecular_dynamics_neural_operator_amd.dataset import ContactMapDataset, write_trajectory_npz
from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN, recursive_propagation, construct_pairdata
from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
import tempfile, os
N, W, T = 28, 10, 40
frames = np.stack([syn.jitter_window(syn.chain_frame(N, seed=1), T, seed=1)[t] for t in range(T)])
d = tempfile.mkdtemp(); path = os.path.join(d, 'traj.npz')
from scipy.spatial import distance_matrix
cms = []
for t in range(T):
    dm = distance_matrix(frames[t], frames[t]); r, c = np.nonzero(dm < 8.0); cms.append(np.concatenate([r, c]).astype(np.int16))
write_trajectory_npz(path, frames, cms, syn.amino_acids(N, seed=1), np.zeros(T, dtype=np.float32))
ds = ContactMapDataset(path, window_size=W, horizon=1)
model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4)
model.load_state_dict(near_identity_state_dict(64, 1024, seed=0, kernel_gain=1e-3, feature_gain=0.1))
model.eval().to('cuda')
for steps in (100, 1000):
    recursive_propagation(model, ds, 'cuda', 20, [0], 8.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = recursive_propagation(model, ds, 'cuda', steps, [0], 8.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"recursive_propagation {steps} steps: {dt*1e3:.1f} ms = {steps/dt:.0f} it/s; last E = {out[-1].edge_index.shape[1]}")

# KernelNN.forward(data) with the sample's own edge list, called in a loop as the notebook does
s = ds[0].to('cuda')
with torch.no_grad():
    for _ in range(20):
        model(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        out = model(s)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"model(data), N={N}: {dt/300*1e3:.3f} ms per call = {300/dt:.0f} calls/s")
