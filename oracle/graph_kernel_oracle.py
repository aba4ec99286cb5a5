"""CPU restatement of the reference's hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module; the product package (``molecular_dynamics_neural_operator_amd``) never does
and fails loudly when its HIP library is missing.

What it restates (plain ``torch`` CPU fp32 + scipy, no torch_geometric), function by function:

  edge_mlp              graph_kernel.py:217-242 (DenseNet.forward), built at :271
  nnconv_forward        graph_kernel.py:194-209 (NNConv_old.forward/message/update) and the
                        torch_geometric ``MessagePassing.propagate`` it calls (PyG >= 2.0, version
                        unpinned in requirements.txt:1-4): gather ``x_j = x[edge_index[0]]``,
                        scatter over ``edge_index[1]`` with ``dim_size = N``, ``mean`` =
                        sum / count.clamp(min=1)
  kernelnn_forward      graph_kernel.py:277-309 (KernelNN.forward) at B=1
  construct_pairdata    graph_kernel.py:362-393 (window version; the notebook's single-frame
                        version, bba_analysis.ipynb raw lines 302-334, is the W=1 case)
  recursive_propagation graph_kernel.py:396-413
  dataset_sample        dataset.py:180-227 (ContactMapDataset.__getitem__)
  lp_loss_rel           graph_kernel.py:105-119
  train_step            graph_kernel.py:453-463 (forward, LpLoss(size_average=False), backward) — autograd over
                        the functions above, B=1 per sample

Pinning: the reference ships no tests, goldens or KATs for this path.  The
oracle is pinned against outputs of the reference's own code run in the build container by
``oracle/gen_golden.py`` (committed under ``tests/golden/``; checked by
``tests/test_oracle_golden.py``).  At the torch_geometric boundary nothing of the reference pins
results, so ``oracle/_stubs/torch_geometric`` is the stated definition there.

``hoist=False`` keeps the reference's behaviour of re-evaluating the (identical) edge-MLP in
every one of the 2*depth conv applications, so that CPU-baseline timings are honest;
``hoist=True`` evaluates it once (same values bit for bit, since the inputs never change).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


# --------------------------------------------------------------------------- edge-MLP
def edge_mlp(edge_attr: Tensor, sd: StateDict, prefix: str = "conv1.net.") -> Tensor:
    """Linear->ReLU->Linear->ReLU->Linear, no final nonlinearity  [E,6] -> [E, Cin*Cout]."""
    h = F.relu(F.linear(edge_attr, sd[prefix + "layers.0.weight"], sd[prefix + "layers.0.bias"]))
    h = F.relu(F.linear(h, sd[prefix + "layers.2.weight"], sd[prefix + "layers.2.bias"]))
    return F.linear(h, sd[prefix + "layers.4.weight"], sd[prefix + "layers.4.bias"])


# --------------------------------------------------------------------------- conv
def nnconv_apply(x: Tensor, edge_index: Tensor, w_e: Tensor, root: Optional[Tensor],
                 bias: Optional[Tensor], aggr: str = "mean") -> Tensor:
    """One conv application given the already-evaluated edge weights ``w_e [E, Cin*Cout]``."""
    n, cin = x.shape
    cout = w_e.shape[1] // cin
    src, dst = edge_index[0], edge_index[1]
    x_j = x.index_select(0, src)
    msg = torch.matmul(x_j.unsqueeze(1), w_e.view(-1, cin, cout)).squeeze(1)
    out = torch.zeros(n, cout, dtype=x.dtype)
    out.index_add_(0, dst, msg)
    if aggr == "mean":
        cnt = torch.zeros(n, dtype=x.dtype)
        cnt.index_add_(0, dst, torch.ones(dst.shape[0], dtype=x.dtype))
        out = out / cnt.clamp(min=1).unsqueeze(-1)
    elif aggr == "max":      # torch_geometric's scatter max: per channel over a node's messages, 0 for no message
        out = torch.zeros(n, cout, dtype=x.dtype)
        out.index_reduce_(0, dst, msg, "amax", include_self=False)
    elif aggr != "add":
        raise ValueError(f"aggr {aggr!r} not restated")
    if root is not None:
        out = out + torch.mm(x, root)
    if bias is not None:
        out = out + bias
    return out


def nnconv_forward(x: Tensor, edge_index: Tensor, edge_attr: Tensor, sd: StateDict,
                   prefix: str = "", aggr: str = "mean") -> Tensor:
    """``NNConv_old.forward`` with parameters taken from ``sd[prefix + ...]``."""
    if x.dim() == 1:
        x = x.unsqueeze(-1)
    if edge_attr.dim() == 1:
        edge_attr = edge_attr.unsqueeze(-1)
    w_e = edge_mlp(edge_attr, sd, prefix + "net.")
    return nnconv_apply(x, edge_index, w_e, sd.get(prefix + "root"), sd.get(prefix + "bias"), aggr)


# --------------------------------------------------------------------------- model
def _lstm_last_hidden(x_position: Tensor, sd: StateDict) -> Tensor:
    """W sequential single-step LSTM calls with batch = N, state carried (graph_kernel.py:279-284)."""
    w, n, d = x_position.shape
    lstm = torch.nn.LSTM(d, d)
    with torch.no_grad():
        for name in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            getattr(lstm, name).copy_(sd["lstm." + name])
    hidden = (torch.zeros(1, n, d), torch.zeros(1, n, d))
    out = None
    with torch.no_grad():
        for t in range(w):
            out, hidden = lstm(x_position[t].unsqueeze(0), hidden)
    return out.reshape(n, d)


def kernelnn_forward(sd: StateDict, x_position: Tensor, x_aminoacid: Tensor, edge_index: Tensor,
                     edge_attr: Tensor, depth: int, return_latent: bool = False,
                     hoist: bool = False):
    """``KernelNN.forward`` for one sample (B=1).  x_position [W,N,3] -> [N, out_width]."""
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}
    with torch.no_grad():
        x = _lstm_last_hidden(x_position.to(torch.float32), sd)
        x = F.linear(x, sd["lstm_fc.weight"], sd["lstm_fc.bias"])
        emb = F.embedding(x_aminoacid, sd["emb.weight"])
        x = torch.cat((emb, x.reshape(emb.shape[0], -1)), dim=1)
        x = F.relu(F.linear(x, sd["fc1.weight"], sd["fc1.bias"]))
        for conv in ("conv1", "conv2"):
            w_e = edge_mlp(edge_attr, sd, conv + ".net.") if hoist else None
            for _ in range(depth):
                if not hoist:
                    w_e = edge_mlp(edge_attr, sd, conv + ".net.")
                x = F.relu(nnconv_apply(x, edge_index, w_e, sd[conv + ".root"], sd[conv + ".bias"], "mean"))
        latent = x.clone()
        out = F.linear(x, sd["fc2.weight"], sd["fc2.bias"])
    return (out, latent) if return_latent else out


def kernelnn_notebook_forward(sd: StateDict, x_position: Tensor, x_aminoacid: Tensor, edge_index: Tensor,
                              edge_attr: Tensor, depth: int, hoist: bool = True) -> Tensor:
    """Notebook-era model (bba_analysis.ipynb:123-128: emb, fc1, conv1, fc2; window 1).  Its source is
    not in the reference tree; this restates the in-tree forward (graph_kernel.py:292-305) without the
    LSTM front-end and the conv2 block.  x_position [N,3] (or [1,N,3])."""
    with torch.no_grad():
        pos = x_position.reshape(-1, 3).to(torch.float32)
        x = torch.cat((F.embedding(x_aminoacid, sd["emb.weight"]), pos), dim=1)
        x = F.relu(F.linear(x, sd["fc1.weight"], sd["fc1.bias"]))
        w_e = edge_mlp(edge_attr, sd, "conv1.net.") if hoist else None
        for _ in range(depth):
            if not hoist:
                w_e = edge_mlp(edge_attr, sd, "conv1.net.")
            x = F.relu(nnconv_apply(x, edge_index, w_e, sd["conv1.root"], sd["conv1.bias"], "mean"))
        return F.linear(x, sd["fc2.weight"], sd["fc2.bias"])


# --------------------------------------------------------------------------- differentiable forward + train step
def lstm_last_hidden_functional(x_position: Tensor, sd: StateDict) -> Tensor:
    """The same W sequential LSTM steps as `_lstm_last_hidden` (graph_kernel.py:279-284; torch.nn.LSTM cell,
    gate order i, f, g, o; zero initial state; batch = atoms), written out so that autograd sees the
    parameters of `sd` and any dtype works."""
    w, n, d = x_position.shape
    w_ih, w_hh = sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"]
    b = sd["lstm.bias_ih_l0"] + sd["lstm.bias_hh_l0"]
    h = torch.zeros(n, d, dtype=w_ih.dtype)
    c = torch.zeros(n, d, dtype=w_ih.dtype)
    for t in range(w):
        gates = x_position[t].to(w_ih.dtype) @ w_ih.t() + h @ w_hh.t() + b
        i, f, g, o = gates.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
    return h


def kernelnn_forward_autograd(sd: StateDict, x_position: Tensor, x_aminoacid: Tensor, edge_index: Tensor,
                              edge_attr: Tensor, depth: int) -> Tensor:
    """`kernelnn_forward` (graph_kernel.py:277-309, B=1) with autograd left on and in the dtype of `sd`
    (the shared edge-MLP evaluated once: identical values, graph_kernel.py:271-273)."""
    dt = sd["fc1.weight"].dtype
    x = lstm_last_hidden_functional(x_position, sd)
    x = F.linear(x, sd["lstm_fc.weight"], sd["lstm_fc.bias"])
    emb = F.embedding(x_aminoacid, sd["emb.weight"])
    x = F.relu(F.linear(torch.cat((emb, x), dim=1), sd["fc1.weight"], sd["fc1.bias"]))
    w_e = edge_mlp(edge_attr.to(dt), sd, "conv1.net.")
    for conv in ("conv1", "conv2"):
        for _ in range(depth):
            x = F.relu(nnconv_apply(x, edge_index, w_e, sd[conv + ".root"], sd[conv + ".bias"], "mean"))
    return F.linear(x, sd["fc2.weight"], sd["fc2.bias"])


def train_step(sd: StateDict, samples: Sequence[dict], depth: int, dtype=torch.float64):
    """Loss and gradients of one `train` iteration (graph_kernel.py:453-467 with
    `LpLoss(size_average=False)`, :547) up to `l2.backward()`: every sample is an independent B=1 forward
    (the only batch size at which the reference's forward is well defined), the loss is the
    sum over samples of ||out_b - y_b|| / ||y_b||.  conv1.net and conv2.net are ONE module in the reference
    (:271-273): `sd`'s conv2.net.* entries are ignored and both key sets receive the shared gradient.
    Returns (loss, out [B*N, out_width], {name: grad})."""
    shared = {k: v for k, v in sd.items() if not k.startswith("conv2.net.")}
    params = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in shared.items()}
    outs, ys = [], []
    for s in samples:
        outs.append(kernelnn_forward_autograd(params, s["x_position"], s["x_aminoacid"], s["edge_index"],
                                              s["edge_attr"], depth))
        ys.append(s["y"].to(dtype))
    out, y = torch.cat(outs), torch.cat(ys)
    b = len(samples)
    loss = lp_loss_rel(out.view(b, -1), y.view(b, -1), size_average=False)
    names = list(params)
    grads = torch.autograd.grad(loss, [params[n] for n in names])
    g = dict(zip(names, grads))
    for k in list(g):
        if k.startswith("conv1.net."):
            g["conv2.net." + k[len("conv1.net."):]] = g[k]
    return float(loss.detach()), out.detach(), g


# --------------------------------------------------------------------------- graph
def radius_graph_coo(frame: np.ndarray, threshold: float = 8.0) -> np.ndarray:
    """Row-major COO of ``distance_matrix(frame, frame) < threshold`` (f64 distances, strict <,
    self-loops kept) -> int64 [2, E] = [rows; cols]  (graph_kernel.py:363-368)."""
    from scipy.sparse import coo_matrix
    from scipy.spatial import distance_matrix
    cm = (distance_matrix(frame, frame) < threshold).astype("int8")
    sp = coo_matrix(cm)
    return np.array([sp.row, sp.col]).astype(np.int64)


def construct_pairdata(x_position: np.ndarray, x_aminoacid: Tensor, threshold: float = 8.0) -> dict:
    """Graph + edge attributes of the LAST window frame; ``edge_attr[e] = [p_row, p_col]``."""
    x_position = np.asarray(x_position)
    if x_position.ndim == 2:  # notebook-era single-frame call
        x_position = x_position[None]
    last = x_position[-1]
    ei = radius_graph_coo(last, threshold)
    ea = np.concatenate([last[ei[0]], last[ei[1]]], axis=1).reshape(-1, 6)
    return dict(
        x_aminoacid=x_aminoacid,
        x_position=torch.from_numpy(np.ascontiguousarray(x_position)).to(torch.float32),
        edge_attr=torch.from_numpy(ea).to(torch.float32),
        edge_index=torch.from_numpy(ei).to(torch.long),
    )


def recursive_propagation(sd: StateDict, depth: int, start_sample: dict, num_steps: int,
                          threshold: float = 8.0, hoist: bool = False) -> List[dict]:
    """Autoregressive loop: forward -> slide window -> rebuild graph on the new last frame."""
    forecasts = []
    inp = start_sample
    for _ in range(num_steps):
        out = kernelnn_forward(sd, inp["x_position"], inp["x_aminoacid"], inp["edge_index"],
                               inp["edge_attr"], depth, hoist=hoist)
        last_window = inp["x_position"].numpy()[1:, :, :]
        new_x = np.vstack([last_window, out.numpy()[None]])
        inp = construct_pairdata(new_x, inp["x_aminoacid"], threshold)
        forecasts.append(inp)
    return forecasts


# --------------------------------------------------------------------------- dataset layout
def dataset_sample(point_cloud: np.ndarray, contact_map: Sequence[np.ndarray], amino_acids: np.ndarray,
                   idx: int, window_size: int = 1, horizon: int = 1) -> dict:
    """``ContactMapDataset.__getitem__``: positions ``point_cloud [T,3,N]`` (as stored on disk),
    ragged flat COO ``contact_map[t] = [rows..., cols...]``.  Graph and edge_attr come from the
    FIRST frame of the window (dataset.py:189-201), target from ``idx+W+h-1`` (:182, :204)."""
    pos = np.transpose(point_cloud, [0, 2, 1])
    ei = np.asarray(contact_map[idx]).reshape(2, -1)
    ea = np.concatenate([pos[idx][ei[0]], pos[idx][ei[1]]], axis=1).reshape(-1, 6)
    return dict(
        x_aminoacid=torch.from_numpy(np.asarray(amino_acids)).to(torch.long),
        x_position=torch.from_numpy(np.ascontiguousarray(pos[idx:idx + window_size])).to(torch.float32),
        y=torch.from_numpy(np.ascontiguousarray(pos[idx + window_size + horizon - 1])).to(torch.float32),
        edge_attr=torch.from_numpy(ea).to(torch.float32),
        edge_index=torch.from_numpy(ei).to(torch.long),
    )


def dataset_len(num_frames: int, window_size: int, horizon: int) -> int:
    return num_frames - window_size - horizon + 1  # dataset.py:177-178


# --------------------------------------------------------------------------- loss
def lp_loss_rel(x: Tensor, y: Tensor, p: int = 2, size_average: bool = True, reduction: bool = True) -> Tensor:
    b = x.size(0)
    diff = torch.norm(x.reshape(b, -1) - y.reshape(b, -1), p, 1)
    yn = torch.norm(y.reshape(b, -1), p, 1)
    if not reduction:
        return diff / yn
    return torch.mean(diff / yn) if size_average else torch.sum(diff / yn)


def lp_loss_abs(x: Tensor, y: Tensor, d: int = 2, p: int = 2, size_average: bool = True) -> Tensor:
    b = x.size(0)
    h = 1.0 / (x.size(1) - 1.0)
    n = (h ** (d / p)) * torch.norm(x.reshape(b, -1) - y.reshape(b, -1), p, 1)
    return torch.mean(n) if size_average else torch.sum(n)


# --------------------------------------------------------------------------- reference init order
def reference_init_state_dict(width: int, ker_width: int, depth: int, ker_in: int, in_width: int = 1,
                              out_width: int = 1, num_embeddings: int = 20, embedding_dim: int = 4,
                              x_position_dim: int = 3, seed: Optional[int] = None) -> StateDict:
    """Parameters in the order the reference's ``KernelNN.__init__`` draws them from the global
    torch RNG (graph_kernel.py:264-275): LSTM, lstm_fc, emb, fc1, the three kernel Linears, then
    conv1 = {re-init of the kernel via ``reset(net)``, root, bias ~ U(+-1/sqrt(Cin))}, conv2 = the
    same again on the SAME kernel object (so conv1.net == conv2.net == the third draw), fc2."""
    import math
    if seed is not None:
        torch.manual_seed(seed)
    lstm = torch.nn.LSTM(x_position_dim, x_position_dim)
    lstm_fc = torch.nn.Linear(x_position_dim, x_position_dim)
    emb = torch.nn.Embedding(num_embeddings, embedding_dim)
    fc1 = torch.nn.Linear(in_width, width)
    dims = [ker_in, ker_width, ker_width, width * width]
    lins = [torch.nn.Linear(dims[j], dims[j + 1]) for j in range(3)]
    convs = {}
    bound = 1.0 / math.sqrt(width)
    for conv in ("conv1", "conv2"):
        for lin in lins:
            lin.reset_parameters()
        root = torch.empty(width, width).uniform_(-bound, bound)
        bias = torch.empty(width).uniform_(-bound, bound)
        convs[conv] = (root, bias)
    fc2 = torch.nn.Linear(width, out_width)
    sd: StateDict = {}
    for k, v in lstm.state_dict().items():
        sd["lstm." + k] = v
    sd["lstm_fc.weight"], sd["lstm_fc.bias"] = lstm_fc.weight.data, lstm_fc.bias.data
    sd["emb.weight"] = emb.weight.data
    sd["fc1.weight"], sd["fc1.bias"] = fc1.weight.data, fc1.bias.data
    for conv in ("conv1", "conv2"):
        sd[conv + ".root"], sd[conv + ".bias"] = convs[conv]
        for j, lin in zip((0, 2, 4), lins):
            sd[f"{conv}.net.layers.{j}.weight"] = lin.weight.data
            sd[f"{conv}.net.layers.{j}.bias"] = lin.bias.data
    sd["fc2.weight"], sd["fc2.bias"] = fc2.weight.data, fc2.bias.data
    return {k: v.detach().clone() for k, v in sd.items()}
