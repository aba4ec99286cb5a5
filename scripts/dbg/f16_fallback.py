import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from molecular_dynamics_neural_operator_amd import ops, synthetic as syn
from molecular_dynamics_neural_operator_amd.graph_kernel import KernelNN
from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict
dev = torch.device('cuda:0')
N, W = 504, 10
sd = near_identity_state_dict(64, 1024, seed=0, kernel_gain=0.02, feature_gain=0.1, kernel_to_coords=1.0)
model = KernelNN(64, 1024, 6, 6, 7, 3, 20, 4); model.load_state_dict(sd); model.eval().to(dev)
win = torch.from_numpy(syn.jitter_window(syn.box_frame(N, seed=1), W, seed=1)).to(dev)
aa = torch.from_numpy(syn.amino_acids(N, seed=1)).to(dev)
def lat(mode, scale):
    model.gemm_mode = mode
    fr = win * scale
    g = ops.radius_graph(fr[-1].contiguous(), N, 8.0 * scale)
    return ops.kernelnn_forward(model.param_pack(dev, conv_mode="factored"), fr.unsqueeze(1), aa, g, edge_pos=fr[-1].contiguous(), return_latent=True)[1]
for scale in (1.0, 3000.0, 3.0e5):
    a1, a2 = lat("split_bf16", scale), lat("split_bf16", scale)
    b1, b2 = lat("split_f16", scale), lat("split_f16", scale)
    print(f"scale {scale}: bf16 repeat equal {torch.equal(a1,a2)}  f16 repeat equal {torch.equal(b1,b2)}  f16==bf16 {torch.equal(a1,b1)}  "
          f"rel diff {float((a1.double()-b1.double()).norm()/a1.double().norm()):.2e}  max {float(a1.abs().max()):.2e}  finite {bool(torch.isfinite(a1).all())}")
