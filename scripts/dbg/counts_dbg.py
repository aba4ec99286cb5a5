import sys, torch
sys.path.insert(0, '.')
from molecular_dynamics_neural_operator_amd import ops, synthetic as syn, _lib
from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
dev = torch.device('cuda')
N, W = 150, 4
sd = near_identity_state_dict(64, 384, seed=4, kernel_gain=3e-2, feature_gain=0.3, kernel_to_coords=1.0)
model = KernelNN(64, 384, 2, 6, 7, 3, 20, 4); model.load_state_dict(sd); model.eval().to(dev)
model.gemm_mode = "split_f16"
win = torch.from_numpy(syn.jitter_window(syn.box_frame(N, seed=3), W, seed=3)).to(dev)
aa = torch.from_numpy(syn.amino_acids(N, seed=3))
g = ops.radius_graph(win[-1], N, 8.0)
pack = model.param_pack(dev, conv_mode="factored")
lib = _lib.load()
for M in (1, 2):
    fr = win.unsqueeze(1).repeat(1, M, 1, 1).contiguous()
    gg = ops.radius_graph(fr[-1].reshape(M * N, 3), N, 8.0)
    nbytes = lib.mdno_kernelnn_workspace_bytes(pack.ref, M, N, gg.edge_cap)
    ws = torch.full((nbytes,), 0x55, dtype=torch.uint8, device=dev)
    for rep in range(2):
        c = {}
        ops.kernelnn_forward(pack, fr, aa, gg, edge_pos=fr[-1].reshape(M * N, 3), workspace=ws, fallback_counts=c)
        print(M, rep, nbytes, c)
