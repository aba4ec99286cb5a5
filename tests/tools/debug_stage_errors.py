"""Dev tool: per-stage error of the HIP path vs the oracle for a given weight set (GPU box)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from molecular_dynamics_neural_operator_amd import _lib, ops, synthetic as syn  # noqa: E402
from molecular_dynamics_neural_operator_amd.weights import near_identity_state_dict  # noqa: E402
from oracle import graph_kernel_oracle as O  # noqa: E402
import torch.nn.functional as F  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
N, W = 28, 10
win = syn.jitter_window(syn.chain_frame(N, seed=0), W, seed=0)
aa = torch.from_numpy(syn.amino_acids(N, seed=0))
for width, k, depth in ((8, 16, 2), (64, 128, 2)):
    sd = near_identity_state_dict(width, k, seed=3, kernel_gain=1e-2, feature_gain=1e-1, kernel_to_coords=1.0)
    pack = ops.ParamPack(sd, depth, dev)
    frames = torch.from_numpy(win).to(dev).unsqueeze(1).contiguous()
    x0 = torch.empty((N, width), device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.mdno_node_prologue_fwd(pack.ref, frames.data_ptr(), 1, W, N, aa.to(dev).data_ptr(), 0,
                                          x0.data_ptr(), st.data_ptr(), None))
    h = O._lstm_last_hidden(torch.from_numpy(win), sd)
    x = F.linear(h, sd["lstm_fc.weight"], sd["lstm_fc.bias"])
    ref0 = F.relu(F.linear(torch.cat((F.embedding(aa, sd["emb.weight"]), x), 1), sd["fc1.weight"], sd["fc1.bias"]))
    e = (x0.cpu() - ref0).abs()
    print(f"width {width}: prologue max err {float(e.max()):.3e} (ref max {float(ref0.abs().max()):.3f}), "
          f"coord-channel err {float(e[:, :6].max()):.3e}; lstm_fc out vs last frame: "
          f"{float((x - torch.from_numpy(win[-1])).abs().max()):.3e}")
    s = O.construct_pairdata(win, aa, 8.0)
    g = ops.radius_graph(frames[-1, 0], N, 8.0)
    out, lat = ops.kernelnn_forward(pack, frames, aa, g, edge_pos=frames[-1, 0], return_latent=True)
    ro, rl = O.kernelnn_forward(sd, s["x_position"], aa, s["edge_index"], s["edge_attr"], depth, return_latent=True)
    print(f"   forward: out err {float((out.cpu() - ro).abs().max()):.3e}, latent err "
          f"{float((lat.cpu() - rl).abs().max()):.3e}")
