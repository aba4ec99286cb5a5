"""Drop-in for the hot path of the reference's ``graph_kernel.py`` — same names, constructor and
``forward`` signatures, state_dict keys and RNG-draw order, with the arithmetic done by the HIP
kernels of libmdno.so (include/mdno.h) on an MI355X.  There is no CPU fallback: calling any
``forward`` with CPU tensors, or without the built library, raises.

  LpLoss                 graph_kernel.py:75-122
  NNConv_old             graph_kernel.py:125-214   (+ torch_geometric MessagePassing.propagate)
  DenseNet               graph_kernel.py:217-242
  KernelNN               graph_kernel.py:245-309
  construct_pairdata     graph_kernel.py:362-393   (notebook variant bba_analysis.ipynb:302-334)
  recursive_propagation  graph_kernel.py:396-413
  propogate              bba_analysis.ipynb:336-358

Differences from the reference, all explicit:
  * ``KernelNN.forward`` reads the window length and atom count from ``data.x_position``'s shape
    ([W,N,3]) instead of a module-global ``args`` (graph_kernel.py:279) and never calls ``.cuda()``.
  * B=1 semantics per sample: a list of samples or a collated batch runs as independent
    block-diagonal members of one forward (``validate``, graph_kernel.py:476-493 -> ``training.validate_epoch``).
  * ``KernelNN.forward`` in training mode with autograd enabled runs the differentiable path of
    ``training.py`` (HIP forward + backward of the kernel-integral block, fp32); stand-alone
    ``NNConv_old`` / ``DenseNet`` forwards are inference-only and raise in that situation.
"""
from __future__ import annotations

import math
from collections import defaultdict
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import MdnoError, require_gpu
from .dataset import ContactMapDataset, PairData  # noqa: F401  (re-exported like the reference)

EPS = 1e-15


# --------------------------------------------------------------------------- init helpers
def uniform(size: int, tensor: Optional[torch.Tensor]) -> None:
    """U(-1/sqrt(size), 1/sqrt(size)) in place (torch_geometric.nn.inits.uniform semantics)."""
    if tensor is not None:
        bound = 1.0 / math.sqrt(size)
        tensor.data.uniform_(-bound, bound)


def reset(value) -> None:
    """Recursively call ``reset_parameters`` (torch_geometric.nn.inits.reset semantics)."""
    if hasattr(value, "reset_parameters"):
        value.reset_parameters()
    else:
        for child in value.children() if hasattr(value, "children") else []:
            reset(child)


def _no_training(module: nn.Module) -> None:
    if module.training and torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(
            "a stand-alone NNConv_old / DenseNet forward is inference-only (the differentiable HIP path is "
            "KernelNN.forward in training mode, training.py): call .eval() or run under torch.no_grad()")


# --------------------------------------------------------------------------- loss
class _LpLossRelFn(torch.autograd.Function):
    """relative L2 loss of a batch and its MSE on the device (csrc/loss.hip): three small launches for what costs
    ~15 ATen ones per training batch.  Returns (loss, mse); only `loss` is differentiable."""

    @staticmethod
    def forward(ctx, x, y, size_average):
        from . import ops
        x, y = x.contiguous(), y.contiguous()
        ctx.set_materialize_grads(False)       # (no zeros tensor — a fill launch — for the MSE output's absent gradient)
        res, stats = ops.lploss_rel_fwd(x, y, size_average)
        ctx.save_for_backward(x, y, stats)
        ctx.size_average = bool(size_average)
        loss, mse = res[0], res[1]
        ctx.mark_non_differentiable(mse)
        return loss, mse

    @staticmethod
    def backward(ctx, g, _g_mse):
        from . import ops
        x, y, stats = ctx.saved_tensors
        if g is None:
            return None, None, None
        return ops.lploss_rel_bwd(x, y, stats, g, ctx.size_average), None, None


class LpLoss(object):
    """Relative / absolute Lp loss (graph_kernel.py:75-122).  The form train() uses — `rel` with p = 2 and a reduction,
    on device tensors (:462, :547) — runs in libmdno (`mdno_lploss_rel_fwd/_bwd`, fixed summation orders); every other
    form (`abs`, p != 2, reduction=False, host tensors) is the reference's torch expression."""

    def __init__(self, d=2, p=2, size_average=True, reduction=True):
        assert d > 0 and p > 0
        self.d, self.p, self.reduction, self.size_average = d, p, reduction, size_average

    def _reduce(self, v):
        if not self.reduction:
            return v
        return torch.mean(v) if self.size_average else torch.sum(v)

    def abs(self, x, y):
        n = x.size()[0]
        h = 1.0 / (x.size()[1] - 1.0)
        norms = (h ** (self.d / self.p)) * torch.norm(x.reshape(n, -1) - y.reshape(n, -1), self.p, 1)
        return self._reduce(norms)

    def rel_with_mse(self, x, y):
        """(loss, F.mse_loss(x, y)) from one pass over the batch — what one iteration of train() computes
        (graph_kernel.py:462-465); device tensors, p = 2, with reduction."""
        n = x.size()[0]
        if not (x.is_cuda and self.p == 2 and self.reduction and x.dtype == torch.float32):
            return self.rel(x, y), torch.nn.functional.mse_loss(x.detach().reshape(n, -1), y.reshape(n, -1))
        return _LpLossRelFn.apply(x.reshape(n, -1), y.to(x.device, torch.float32).reshape(n, -1), self.size_average)

    def rel(self, x, y):
        n = x.size()[0]
        if x.is_cuda and self.p == 2 and self.reduction and x.dtype == torch.float32:
            return _LpLossRelFn.apply(x.reshape(n, -1), y.to(x.device, torch.float32).reshape(n, -1), self.size_average)[0]
        diff = torch.norm(x.reshape(n, -1) - y.reshape(n, -1), self.p, 1)
        ynorm = torch.norm(y.reshape(n, -1), self.p, 1)
        return self._reduce(diff / ynorm)

    def __call__(self, x, y):
        return self.rel(x, y)


# --------------------------------------------------------------------------- edge-MLP
class DenseNet(nn.Module):
    """``DenseNet(layers, nonlinearity, out_nonlinearity=None, normalize=False)`` (graph_kernel.py:217-242).  The
    configuration the model uses — three Linear layers with ReLU between them (graph_kernel.py:271) — runs as the
    fused edge-MLP kernels; any other depth, ``normalize=True`` (BatchNorm1d in eval mode, folded into the Linear in
    front of it) and a ReLU ``out_nonlinearity`` run layer by layer on the library's Linear kernel.  Other
    nonlinearities and batch statistics (BatchNorm1d in training mode) are not implemented."""

    def __init__(self, layers, nonlinearity, out_nonlinearity=None, normalize=False):
        super().__init__()
        self.n_layers = len(layers) - 1
        assert self.n_layers >= 1
        self.layers = nn.ModuleList()
        for j in range(self.n_layers):
            self.layers.append(nn.Linear(layers[j], layers[j + 1]))
            if j != self.n_layers - 1:
                if normalize:
                    self.layers.append(nn.BatchNorm1d(layers[j + 1]))
                self.layers.append(nonlinearity())
        if out_nonlinearity is not None:
            self.layers.append(out_nonlinearity())
        self._hip_ok = (self.n_layers == 3 and not normalize and out_nonlinearity is None
                        and nonlinearity is nn.ReLU)
        self._dims = list(layers)

    def hip_weights(self):
        if not self._hip_ok:
            raise NotImplementedError(
                "libmdno implements the edge-MLP as Linear-ReLU-Linear-ReLU-Linear (graph_kernel.py:271); "
                f"got layers={self._dims}")
        l0, l2, l4 = self.layers[0], self.layers[2], self.layers[4]
        return (l0.weight, l0.bias, l2.weight, l2.bias, l4.weight, l4.bias)

    def _layerwise(self, x):
        """Any depth / eval-mode BatchNorm / ReLU output: one Linear kernel per layer (graph_kernel.py:239-242)."""
        mods = list(self.layers)
        i = 0
        while i < len(mods):
            lin = mods[i]
            if not isinstance(lin, nn.Linear):
                raise NotImplementedError(f"DenseNet: unexpected layer {type(lin).__name__} at position {i}")
            w, b = lin.weight, lin.bias
            i += 1
            if i < len(mods) and isinstance(mods[i], nn.BatchNorm1d):
                bn = mods[i]
                if bn.training or not bn.track_running_stats:
                    raise NotImplementedError("DenseNet(normalize=True): batch statistics are not implemented; call .eval()")
                scale = (bn.weight if bn.affine else torch.ones_like(bn.running_var)) / torch.sqrt(bn.running_var + bn.eps)
                shift = (bn.bias if bn.affine else torch.zeros_like(bn.running_mean)) - bn.running_mean * scale
                w, b = w * scale[:, None], (b if b is not None else 0.0) * scale + shift
                i += 1
            relu = False
            if i < len(mods) and not isinstance(mods[i], (nn.Linear, nn.BatchNorm1d)):
                if not isinstance(mods[i], nn.ReLU):
                    raise NotImplementedError(f"DenseNet: nonlinearity {type(mods[i]).__name__} (the HIP path has ReLU)")
                relu = True
                i += 1
            x = ops.linear(x, w, b, relu=relu)
        return x

    def forward(self, x):
        _no_training(self)
        if not self._hip_ok:
            with torch.no_grad():
                return self._layerwise(ops.f32(x))
        w = self.hip_weights()
        x = ops.f32(x)
        E = x.shape[0]
        ne = torch.full((1,), E, dtype=torch.int32, device=x.device)
        g = ops.CSRGraph(None, None, None, ne, max(E, 1), None, None)
        with torch.no_grad():
            return ops.edge_mlp(w, self._dims[0], self._dims[1], self._dims[3], g, edge_attr=x)[:E]


# --------------------------------------------------------------------------- conv
class NNConv_old(nn.Module):
    """Edge-conditioned convolution, ``out_i = aggr_j(x_j . net(e_ji)) + x_i . root + bias``."""

    def __init__(self, in_channels, out_channels, net, aggr="add", root_weight=True, bias=True, **kwargs):
        super().__init__()
        if kwargs.get("flow", "source_to_target") != "source_to_target" or kwargs.get("node_dim", -2) != -2:
            raise NotImplementedError("only flow='source_to_target', node_dim=-2 (the reference's defaults)")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.net = net
        self.aggr = aggr
        if root_weight:
            self.root = nn.Parameter(torch.Tensor(in_channels, out_channels))
        else:
            self.register_parameter("root", None)
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        reset(self.net)
        size = self.in_channels
        uniform(size, self.root)
        uniform(size, self.bias)

    def forward(self, x, edge_index, edge_attr):
        _no_training(self)
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        pseudo = edge_attr.unsqueeze(-1) if edge_attr.dim() == 1 else edge_attr
        if self.aggr not in ("add", "mean", "max"):
            raise NotImplementedError(f"aggr={self.aggr!r}: the HIP path implements 'add', 'mean' and 'max'")
        with torch.no_grad():
            graph = ops.coo_to_csr(edge_index, x.shape[0])
            if getattr(self.net, "_hip_ok", False):
                dims = self.net._dims
                w_e = ops.edge_mlp(self.net.hip_weights(), dims[0], dims[1], dims[3], graph, edge_attr=pseudo)
            else:       # any other edge network: evaluated in COO order, rows then put in the conv's edge order
                w_e = self.net(pseudo).index_select(0, graph.perm[:pseudo.shape[0]].long()).contiguous()
            return ops.nnconv(x, graph, w_e, self.root, self.bias, self.aggr, relu=False)

    def __repr__(self):
        return "{}({}, {})".format(self.__class__.__name__, self.in_channels, self.out_channels)


# --------------------------------------------------------------------------- model
class KernelNN(nn.Module):
    def __init__(self, width: int, ker_width: int, depth: int, ker_in: int, in_width: int = 1,
                 out_width: int = 1, num_embeddings: int = 20, embedding_dim: int = 4,
                 x_position_dim: int = 3) -> None:
        super().__init__()
        self.depth = depth
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        self.x_position_dim = x_position_dim
        # module order == RNG draw order of the reference (graph_kernel.py:264-275)
        self.lstm = nn.LSTM(x_position_dim, x_position_dim)
        self.lstm_fc = nn.Linear(x_position_dim, x_position_dim)
        self.emb = nn.Embedding(num_embeddings, embedding_dim)
        self.fc1 = nn.Linear(in_width, width)
        kernel = DenseNet([ker_in, ker_width, ker_width, width ** 2], nn.ReLU)
        self.conv1 = NNConv_old(width, width, kernel, aggr="mean")
        self.conv2 = NNConv_old(width, width, kernel, aggr="mean")
        self.fc2 = nn.Linear(width, out_width)
        self._pack = None
        self._pack_key = None
        # how the wide edge-MLP GEMMs run (include/mdno.h MDNO_GEMM_*): "split_bf16" = exact 3-way bf16
        # split of the fp32 operands, 6 products, fp32 accumulation (fp32-level error, 2-3x faster than
        # fp32 MFMA); "split_f16" (default) = the same, with the k x k hidden layer of the factored path on
        # two fp16 planes and 3 products (half the matrix work at the same measured error; values outside
        # fp16's range are detected on the device and that chunk is redone by the bf16 kernels);
        # "f32" = fp32-input MFMA, bit-for-bit an fmaf chain
        self.gemm_mode = "split_f16"
        # training (training.py): "fp32" or "bf16" (bf16 storage of h1, h2, W_e, dW_e + single-product bf16
        # GEMMs with fp32 accumulation; fp32 master parameters)
        self.train_precision = "fp32"
        # how conv applications run inside the on-device rollout / position-graph forward
        # (include/mdno.h MDNO_CONV_*): "factored" = the reference's sums reassociated per destination
        # node, W_e never formed (csrc/moment.hip; needs width 64 and ker_width % 128 == 0, otherwise the
        # library itself runs materialized); "materialized" = the reference's W_e formulation; "auto"
        # (default) = factored once the graph is large enough to pay for its fixed cost per application
        # (edge capacity >= 24,576), materialized below.  forward(data) with an explicit
        # edge_index/edge_attr follows the same rule on its counted graph.
        self.conv_mode = "auto"
        # gemm_mode "split_f16" decides on the device, piece by piece, whether a product runs on two fp16 planes or is
        # redone on three bf16 planes (same result, more matrix work).  With `track_fallbacks = True` an inference
        # forward(data) reads the device counters back into `last_fallback_counts` (ops.FALLBACK_KEYS; all zero = the
        # fast path everywhere) at the cost of one stream synchronisation; RolloutEngine.fallback_counts() is the
        # rollout's counterpart.
        self.track_fallbacks = False
        self.last_fallback_counts = None

    def __getstate__(self):
        # the cached ParamPack holds device pointers in a ctypes struct: never copied or pickled
        # (copy.deepcopy(model), torch.save(model)); it is rebuilt on first use
        state = self.__dict__.copy()
        state["_pack"], state["_pack_key"], state["_packs"] = None, None, {}
        return state

    # -- parameter pack (device pointers) cached until a parameter changes
    def param_pack(self, device=None, conv_mode: Optional[str] = None) -> ops.ParamPack:
        device = require_gpu(device)
        conv_mode = conv_mode or self.conv_mode
        params = list(self.parameters())
        # fp32 contiguous parameters on `device` are VIEWED by the pack (device pointers to their storage), so an
        # in-place update (an optimizer step) needs no new pack; anything the pack had to copy is keyed by version
        viewed = all(p.dtype == torch.float32 and p.is_contiguous() and p.device.type == device.type and
                     (device.index is None or p.device.index == device.index) for p in params)
        key = (str(device), self.gemm_mode, conv_mode) + tuple(
            (p.data_ptr(), 0 if viewed else p._version) for p in params)
        # an UNTIED conv2.net: the pack decides by VALUE whether the two edge-MLPs are one (and then evaluates one);
        # that decision must not outlive an in-place change of either, so their versions stay in the key
        conv2 = getattr(self, "conv2", None)
        if conv2 is not None and conv2.net is not self.conv1.net:
            key += tuple(p._version for net in (self.conv1.net, conv2.net) for p in net.parameters())
        # one pack per formulation: a model that serves both (the rollout engine's factored plan and explicit-edge
        # forwards that resolve to materialized, say) does not rebuild either on every switch
        packs = self.__dict__.setdefault("_packs", {})
        hit = packs.get(conv_mode)
        if hit is None or hit[0] != key:
            hit = packs[conv_mode] = (key, ops.ParamPack(self.state_dict(), self.depth, device, self.gemm_mode, conv_mode))
        self._pack, self._pack_key = hit[1], key
        return hit[1]

    def _conv_mode_for_edges(self, device, members: int, n_atoms: int, n_edges: int) -> str:
        """Formulation for a forward on an explicit edge list.  "auto": the library's rule on the counted graph
        (mdno_conv_mode_for_graph: factored for dense graphs — mean degree >= 40 and >= 16,384 edges per member —,
        materialized for protein-like chains).  The factored form (csrc/moment.hip) takes any edge list."""
        if self.conv_mode == "materialized":
            return "materialized"
        from . import _lib
        lib = _lib.load()
        if self.conv_mode == "factored":      # asked for: taken wherever the model's dimensions allow it
            pack = self.param_pack(device, conv_mode="factored")      # (the pack the forward then runs with)
            return "factored" if int(lib.mdno_resolve_conv_mode(pack.ref, int(members), 1 << 40)) == _lib.CONV_MODES["factored"] \
                else "materialized"
        # "auto": the rule reads the GEMM mode and the dimensions only, not the pack's conv_mode — whichever pack is
        # cached serves (asking for a "factored" pack here evicted the materialized one on every forward)
        cached = next((v[1] for v in self.__dict__.get("_packs", {}).values() if v[1].gemm_mode == self.gemm_mode), None)
        pack = cached if cached is not None else self.param_pack(device, conv_mode="materialized")
        mode = int(lib.mdno_conv_mode_for_graph(pack.ref, int(members), int(n_atoms), int(n_edges)))
        return {v: k for k, v in _lib.CONV_MODES.items()}[mode]

    def forward(self, data, return_latent: bool = False, single_example: bool = False, _status=None):
        """``data``: one ``PairData`` sample, a list of samples (what the reference's DataListLoader yields and
        ``validate`` / ``train`` pass to ``model(batch)``, graph_kernel.py:454, :485) or an already collated batch
        (``training.collate`` / ``DeviceTrajectory.batch``: time-major x_position [W,B*N,3], ``num_graphs`` = B).
        Training mode with autograd on: the differentiable path.  Otherwise inference under no_grad: the B samples
        run as B block-diagonal members of ONE forward (B=1 semantics per sample, nothing kept for a backward);
        every sample's rows are bit-identical to ``model(sample)`` on it alone."""
        if self.training and torch.is_grad_enabled():
            # differentiable path (training.py): HIP forward + backward of the kernel-integral block
            from .training import train_forward
            if return_latent:
                raise NotImplementedError("return_latent is an inference-time option")
            return train_forward(self, data)
        batched = not isinstance(data, PairData)
        if batched:
            from .training import collate
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise MdnoError("KernelNN.forward needs the model on the GPU (model.to('cuda')); no CPU fallback")
            if len(data) == 0:
                raise MdnoError("KernelNN.forward: empty batch")
            data = collate([s if s.x_position.is_cuda else PairData(**{k: getattr(s, k) for k in PairData._FIELDS}).to(dev)
                            for s in data])
        x_position = data.x_position
        if x_position.dim() == 2:  # notebook-era single-frame sample [N,3]
            x_position = x_position.unsqueeze(0)
        if not x_position.is_cuda:
            raise MdnoError("KernelNN.forward needs the sample on the GPU (data.to('cuda')); no CPU fallback")
        n_rows = data.x_aminoacid.shape[0]
        B = int(getattr(data, "num_graphs", 1))
        if x_position.shape[1] != n_rows or B < 1 or n_rows % B:
            raise MdnoError(
                f"x_position {tuple(x_position.shape)} vs {n_rows} nodes in {B} graph(s): a batch is a list of PairData "
                "or a collated one (training.collate / DeviceTrajectory.batch: x_position [W,B*N,3], num_graphs = B)")
        W = x_position.shape[0]
        with torch.no_grad():
            pack = self.param_pack(x_position.device, conv_mode=self._conv_mode_for_edges(
                x_position.device, B, n_rows // B, int(data.edge_index.shape[1])))
            graph = ops.coo_to_csr(data.edge_index.to(x_position.device), n_rows, validate=_status is None, status=_status)
            counts = {} if getattr(self, "track_fallbacks", False) else None
            out, latent = ops.kernelnn_forward(pack, ops.f32(x_position).reshape(W, B, n_rows // B, 3),
                                               data.x_aminoacid.to(x_position.device), graph,
                                               edge_attr=data.edge_attr.to(x_position.device), return_latent=return_latent,
                                               check_status=_status is None, fallback_counts=counts)
            if counts is not None:
                self.last_fallback_counts = counts
        return [out, latent] if return_latent else out


class KernelNNNotebook(KernelNN):
    """The model the notebook was run with (bba_analysis.ipynb:45-47, 61-70, 123-128): window 1, no
    LSTM, ONE conv block applied `depth` times, `kernel_width` 512 —
    ``Embedding(20,4), Linear(7,64), NNConv_old(64,64), Linear(64,3)``.  Its source is not in the
    reference tree (the in-tree file is a later commit); the forward is inferred from
    the repr, the checkpoint keys and the in-tree forward with the LSTM and conv2 removed:
    ``x = relu(fc1(cat(emb(aa), pos)))``; ``depth`` x ``relu(conv1)``; ``fc2``.  Parity for this variant
    is therefore against this repo's oracle only."""

    def __init__(self, width: int, ker_width: int, depth: int, ker_in: int, in_width: int = 1,
                 out_width: int = 1, num_embeddings: int = 20, embedding_dim: int = 4) -> None:
        nn.Module.__init__(self)
        self.depth = depth
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        self.x_position_dim = 3
        self.emb = nn.Embedding(num_embeddings, embedding_dim)
        self.fc1 = nn.Linear(in_width, width)
        kernel = DenseNet([ker_in, ker_width, ker_width, width ** 2], nn.ReLU)
        self.conv1 = NNConv_old(width, width, kernel, aggr="mean")
        self.fc2 = nn.Linear(width, out_width)
        self._pack = None
        self._pack_key = None
        self.gemm_mode = "split_f16"
        self.conv_mode = "auto"         # as KernelNN.conv_mode


# --------------------------------------------------------------------------- graph construction
def construct_pairdata(x_position, x_aminoacid, threshold: float = 8.0) -> PairData:
    """Radius graph + edge attributes of the LAST frame of ``x_position`` ([W,N,3]; a single frame
    [N,3] is accepted as in the notebook).  Returns tensors on the GPU."""
    dev = require_gpu()
    xp = torch.as_tensor(np.asarray(x_position) if not torch.is_tensor(x_position) else x_position)
    xp = xp.to(device=dev, dtype=torch.float32)
    single = xp.dim() == 2
    last = (xp if single else xp[-1]).contiguous()
    n = last.shape[0]
    g = ops.radius_graph(last, n, threshold)
    edge_index = g.to_edge_index()
    edge_attr = torch.cat([last[edge_index[0]], last[edge_index[1]]], dim=1)
    if torch.is_tensor(x_aminoacid):
        x_aminoacid = x_aminoacid.to(dev)
    return PairData(x_aminoacid=x_aminoacid, x_position=xp, edge_attr=edge_attr, edge_index=edge_index)


# --------------------------------------------------------------------------- rollout
def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def recursive_propagation(model, dataset, device, num_steps: int, starting_points: list,
                          threshold: float = 8.0) -> List[PairData]:
    """Autoregressive rollout from each starting sample; returns the per-step ``PairData`` (on the
    CPU, like the reference).  The loop itself runs on the device (rollout.RolloutEngine)."""
    from .rollout import RolloutEngine
    net = _unwrap(model)
    net.eval()
    forecasts: List[PairData] = []
    for start in starting_points:
        sample = dataset[start]
        win = sample.x_position if sample.x_position.dim() == 3 else sample.x_position.unsqueeze(0)
        W, N, _ = win.shape
        eng = RolloutEngine(net, members=1, n_atoms=N, window=W, threshold=threshold, max_steps=num_steps,
                            device=device)
        eng.reset(win.unsqueeze(1), sample.x_aminoacid)
        if num_steps > 0:
            # iteration 0 runs on the sample's own graph (first window frame), later ones on the
            # graph rebuilt from the newest frame — exactly the reference's call sites
            eng.first_step_from_sample(sample.edge_index, sample.edge_attr)
            eng.step(num_steps - 1)
        eng.synchronize()
        traj = eng.frames()                                                      # [steps,1,N,3]
        frames = torch.cat([win.to(traj.device), traj[:, 0]], dim=0)             # [W+steps,N,3]
        forecasts.extend(_pairdata_of_windows(frames, W, num_steps, sample.x_aminoacid, threshold))
    return forecasts


def _pairdata_of_windows(frames: torch.Tensor, W: int, num_steps: int, x_aminoacid, threshold: float) -> List[PairData]:
    """``construct_pairdata(frames[i+1 : i+1+W]).to("cpu")`` for i = 0 .. num_steps-1 (what the reference returns per
    step, graph_kernel.py:406-412), with the graphs of many steps built in ONE launch — every newest frame is a
    "member" of one radius-graph call — and one device-to-host copy per tensor: at N = 28 a step of the rollout takes
    0.1 ms, a graph + four copies per step took longer than that."""
    N = frames.shape[1]
    aa = x_aminoacid.cpu() if torch.is_tensor(x_aminoacid) else x_aminoacid
    frames_cpu = frames.cpu()
    out: List[PairData] = []
    chunk = max(1, min(num_steps, (64 << 20) // max(N * N, 1)))      # <= 64M edge slots per call
    for s0 in range(0, num_steps, chunk):
        s1 = min(num_steps, s0 + chunk)
        last = frames[s0 + W:s1 + W].contiguous()                    # newest frame of windows s0 .. s1-1: [S,N,3]
        S = s1 - s0
        g = ops.radius_graph(last.reshape(S * N, 3), N, threshold)
        e = g.edge_count()
        dst, src = g.dst[:e].long(), g.src[:e].long()
        flat = last.reshape(S * N, 3)
        edge_attr = torch.cat([flat[dst], flat[src]], dim=1).cpu()   # [pos[row], pos[col]] as construct_pairdata
        offs = g.row_ptr[::N].cpu().tolist()                          # first edge of every member (and the total)
        dst, src = dst.cpu(), src.cpu()
        for m in range(S):
            a, b = offs[m], offs[m + 1]
            ei = torch.stack([dst[a:b] - m * N, src[a:b] - m * N])
            i = s0 + m
            out.append(PairData(x_aminoacid=aa, x_position=frames_cpu[i + 1:i + 1 + W].clone(),
                                edge_attr=edge_attr[a:b].clone(), edge_index=ei))
    return out


def propogate(model, dataset, device, num_steps: int, threshold: float = 8.0):
    """Notebook rollout: starts at ``dataset[0]`` and also records the per-step MSE against
    ``dataset[i+1].x_position`` (bba_analysis.ipynb:351)."""
    forecasts = recursive_propagation(model, dataset, device, num_steps, [0], threshold)
    metrics = defaultdict(list)
    for i, f in enumerate(forecasts):
        truth = dataset[i + 1].x_position
        truth = (truth if truth.dim() == 2 else truth[-1]).cpu().numpy()
        metrics["mse"].append(float(((f.x_position[-1].numpy() - truth) ** 2).mean()))
    return forecasts, dict(metrics)
