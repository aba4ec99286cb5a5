"""Training path (BASELINE.json configs[3]): the whole differentiable forward and
backward of `KernelNN` runs in libmdno's HIP kernels behind three `torch.autograd.Function`s — the node
prologue (LSTM(3,3) over the window, lstm_fc, Embedding, fc1, ReLU: graph_kernel.py:279-298), the
kernel-integral block (shared edge-MLP + 2*depth conv applications, :299-302) and fc2 (:305).  PyTorch
holds the parameters, chains the three functions and runs the optimizer; it computes nothing.

Replaces what autograd + torch_geometric do in `train()` (graph_kernel.py:445-474).  Members of a
batch are independent B=1 problems (block-diagonal graph); the reference's batched mode threads one
LSTM state through the batch axis and is not reproduced.

Two precisions (`model.train_precision`):
  "fp32" (default)  fp32 storage everywhere; the wide GEMMs per `model.gemm_mode` (bf16-split planes at
                    fp32-level accuracy, or fp32 MFMA); gradients match an fp64 replica to ~1e-6.
  "bf16"            BASELINE.json configs[3] ("bf16"): the block's large tensors (h1, h2, W_e, dW_e) are
                    stored in bf16 and every GEMM is one bf16 MFMA product with fp32 accumulation
                    (csrc/train_bf16.hip); parameters stay fp32 masters, reductions stay fp32.  Half the
                    memory of the two E x 16 KiB tensors, gradients match the fp64 replica to ~1e-2.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from ._lib import MdnoError
from .dataset import PairData


class KernelIntegralBlock(torch.autograd.Function):
    """x_L = conv2^depth(conv1^depth(x_0)) with W_e = net(edge_attr) shared by every application
    (graph_kernel.py:271-273, :299-302), materialised formulation.  gemm_mode as for inference:
    "split_bf16" runs the four wide C = A.W^T products (two forward, two backward) on the bf16 matrix
    pipe with the exact 3-way split, "f32" on the fp32 MFMA; the weight-gradient products A^T.B are
    fp32 either way."""

    @staticmethod
    def forward(ctx, x0, edge_attr, graph, depth, gemm_mode, w0, b0, w1, b1, w2, b2, root1, bias1, root2, bias2):
        """graph.x_stack (optional): f32 [2*depth+1, R, 64] whose layer 0 IS x0 — train_forward lets the node prologue
        write there, so the block copies nothing in.  Returns layer 2*depth of the stack (a view: the stack is kept
        for the backward anyway)."""
        R = x0.shape[0]
        X = getattr(graph, "x_stack", None)
        E = graph.edge_count()
        ea = ops.permute_rows(edge_attr, graph.perm, E) if graph.perm is not None else ops.f32(edge_attr).contiguous()
        L = 2 * depth
        if X is None or X.data_ptr() != x0.data_ptr() or tuple(X.shape) != (L + 1, R, 64):
            X = torch.empty((L + 1, R, 64), dtype=torch.float32, device=x0.device)
            X[0].copy_(x0)
        ctx.bf16 = gemm_mode == "bf16"
        if ctx.bf16:
            return KernelIntegralBlock._forward_bf16(ctx, X, ea, graph, depth, w0, b0, w1, b1, w2, b2, root1, bias1,
                                                     root2, bias2)
        h1 = ops.linear(ea, w0, b0, relu=True)
        h2 = ops.linear(h1, w1, b1, relu=True, gemm_mode=gemm_mode)
        w_e = ops.linear(h2, w2, b2, relu=False, gemm_mode=gemm_mode)
        ops.nnconv_chain_fwd(X, graph, w_e, root1, bias1, root2, bias2, depth)        # the 2*depth applications, one call
        ctx.graph, ctx.depth, ctx.gemm_mode = graph, depth, gemm_mode
        ctx.save_for_backward(ea, h1, h2, w_e, X, w0, w1, w2, root1, root2)
        return X[L]

    @staticmethod
    def _forward_bf16(ctx, X, ea, graph, depth, w0, b0, w1, b1, w2, b2, root1, bias1, root2, bias2):
        h1 = ops.linear_smallk_bf16(ea, w0, b0, relu=True)               # K = 6: fp32 fmaf chains, stored bf16
        h2 = ops.linear_bf16(h1, w1, b1, relu=True, out_bf16=True)
        w_e = ops.linear_bf16(h2, w2, b2, relu=False, out_bf16=True)     # [E, 4096] bf16: 8 KiB per edge
        L = 2 * depth
        ops.nnconv_chain_fwd(X, graph, w_e, root1, bias1, root2, bias2, depth)
        ctx.graph, ctx.depth, ctx.gemm_mode = graph, depth, "bf16"
        ctx.save_for_backward(ea, h1, h2, w_e, X, w0, w1, w2, root1, root2)
        return X[L]

    @staticmethod
    def _backward_bf16(ctx, g_out):
        ea, h1, h2, w_e, X, w0, w1, w2, root1, root2 = ctx.saved_tensors
        graph, depth = ctx.graph, ctx.depth
        L, R = 2 * depth, X.shape[1]
        by_src = getattr(graph, "by_src", None) or ops.source_sorted(graph, R)
        inv = ops.inv_degree(graph, "mean")
        GZ, GS, g = ops.nnconv_chain_bwd(g_out, X, inv, by_src, w_e, root1, root2, depth)
        d_root1, d_bias1, d_root2, d_bias2 = ops.nnconv_bwd_root_pair(X[0:L], GZ[0:L])       # both convs, one launch
        d_we, d_b2 = ops.nnconv_bwd_we_bf16(X[0:L], GS, graph, with_colsum=True)      # bf16 [E, 4096] and its column sums, one pass
        del GZ, GS
        d_w2 = ops.gemm_atb_bf16(d_we, h2)
        gz2 = ops.linear_bf16_relu_bwd(d_we, ops.transpose(w2), h2)        # bf16((h2 > 0) * (dW_e . W2))
        del d_we
        d_b1 = ops.colsum_bf16(gz2)
        d_w1 = ops.gemm_atb_bf16(gz2, h1)
        gz1 = ops.linear_bf16_relu_bwd(gz2, ops.transpose(w1), h1)
        # first layer: bias and weight gradient in ONE pass over gz1 (d_w0 = gz1^T . ea with the fp32 attributes)
        d_b0, d_w0 = ops.colsum_atb_bf16(gz1, ea)
        return (g, None, None, None, None, d_w0, d_b0, d_w1, d_b1, d_w2, d_b2, d_root1, d_bias1, d_root2, d_bias2)

    @staticmethod
    def backward(ctx, g_out):
        if ctx.bf16:
            return KernelIntegralBlock._backward_bf16(ctx, g_out)
        ea, h1, h2, w_e, X, w0, w1, w2, root1, root2 = ctx.saved_tensors
        graph, depth, gemm_mode = ctx.graph, ctx.depth, ctx.gemm_mode
        L, R = 2 * depth, X.shape[1]
        by_src = getattr(graph, "by_src", None) or ops.source_sorted(graph, R)
        inv = ops.inv_degree(graph, "mean")
        GZ, GS, g = ops.nnconv_chain_bwd(g_out, X, inv, by_src, w_e, root1, root2, depth)
        d_root1, d_bias1, d_root2, d_bias2 = ops.nnconv_bwd_root_pair(X[0:L], GZ[0:L])       # both convs, one launch
        d_we, d_b2 = ops.nnconv_bwd_we(X[0:L], GS, graph, with_colsum=True)        # and its column sums, one pass
        del GZ, GS
        # edge-MLP backward
        d_w2 = ops.gemm_atb(d_we, h2, gemm_mode=gemm_mode)
        gz2 = ops.relu_bwd(ops.linear(d_we, ops.transpose(w2), None, gemm_mode=gemm_mode), h2)
        del d_we
        d_b1 = ops.colsum(gz2)
        d_w1 = ops.gemm_atb(gz2, h1, gemm_mode=gemm_mode)
        gz1 = ops.relu_bwd(ops.linear(gz2, ops.transpose(w1), None, gemm_mode=gemm_mode), h1)
        d_b0 = ops.colsum(gz1)
        d_w0 = ops.gemm_atb(gz1, ea)
        return (g, None, None, None, None, d_w0, d_b0, d_w1, d_b1, d_w2, d_b2, d_root1, d_bias1, d_root2, d_bias2)


class NodePrologue(torch.autograd.Function):
    """x0 = relu(fc1([emb(aa), lstm_fc(LSTM over the window)]))  (graph_kernel.py:279-298), B=1 semantics per
    sample.  Parameters arrive as tensors (autograd tracks them) and as the model's ParamPack (device pointers)."""

    @staticmethod
    def forward(ctx, pack, frames, aa, *params):
        x0 = ops.node_prologue(pack, frames, aa, status=pack.train_status, out=getattr(pack, "prologue_out", None))
        ctx.pack, ctx.names = pack, pack.prologue_names
        ctx.save_for_backward(frames, aa, x0)
        return x0

    @staticmethod
    def backward(ctx, g0):
        frames, aa, x0 = ctx.saved_tensors
        grads = ops.node_prologue_bwd(ctx.pack, frames, aa, x0, g0.contiguous())
        return (None, None, None) + tuple(grads[n] for n in ctx.names)


class FcOut(torch.autograd.Function):
    """fc2 (graph_kernel.py:305)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return ops.fc_out(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx, d_w, d_b = ops.fc_out_bwd(x, w, g.contiguous())
        return dx, d_w, d_b


_PROLOGUE_KEYS = ("lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "lstm_fc.weight",
                  "lstm_fc.bias", "emb.weight", "fc1.weight", "fc1.bias")


def collate(samples: Sequence[PairData]) -> PairData:
    """Block-diagonal batch; x_position stacked time-major [W, B*N, 3]."""
    if isinstance(samples, PairData):
        return samples
    b = PairData.collate(samples)
    W = samples[0].x_position.shape[0]
    b.x_position = torch.cat([s.x_position for s in samples], dim=1) if samples[0].x_position.dim() == 3 else \
        torch.cat([s.x_position.unsqueeze(0) for s in samples], dim=1)
    assert b.x_position.shape[0] == W or samples[0].x_position.dim() == 2
    b.num_graphs = len(samples)
    return b


class DeviceTrajectory:
    """A `ContactMapDataset` resident in HBM, handing out training batches built ON the device
    (include/mdno.h mdno_collate_samples).  Stands in for the reference's `DataListLoader` + torch_geometric
    collation (graph_kernel.py:513-519; `model(batch)` at :454 collates per step on the host): per batch the
    host only fills a 3*B+1 word table from the dataset's offsets — no per-sample Python, no per-sample H2D
    copies, no device->host read.  `batch(indices)` returns a collated `PairData` (time-major x_position
    [W,B*N,3], y, edge_index, edge_attr, x_aminoacid tiled) whose samples are `dataset[i]` for i in indices,
    in that order — the same tensors `collate([dataset[i] ...])` builds on the host (tested bitwise)."""

    def __init__(self, dataset, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise MdnoError("DeviceTrajectory needs a GPU device (no CPU fallback)")
        self.W, self.horizon = int(dataset.window_size), int(dataset.horizon)
        self.length = len(dataset)
        pos = np.ascontiguousarray(dataset.edge_attrs, dtype=np.float32)                  # [T,N,3]
        self.N = int(pos.shape[1])
        cms = [np.asarray(c).reshape(2, -1) for c in dataset.edge_indices]
        counts = np.array([c.shape[1] for c in cms], dtype=np.int64)
        self.counts = counts
        self.offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.pos = torch.from_numpy(pos).to(self.device)
        self.rows = torch.from_numpy(np.concatenate([c[0] for c in cms]).astype(np.int32)).to(self.device)
        self.cols = torch.from_numpy(np.concatenate([c[1] for c in cms]).astype(np.int32)).to(self.device)
        self.x_aminoacid = dataset.x_aminoacid.to(self.device)
        self._aa_tiled = {}
        # the per-batch index table travels through a ring of PINNED host buffers: the upload is asynchronous, and a
        # slot is only rewritten once the copy that read it has passed (its event) — a pageable source would have to
        # be staged synchronously by the runtime to be safe
        self._meta_ring, self._meta_slot = [], 0

    _META_SLOTS = 4

    def _meta_buffer(self, words: int):
        if len(self._meta_ring) < self._META_SLOTS:
            self._meta_ring.append([torch.empty(max(words, 1024), dtype=torch.int64).pin_memory(), None])
            slot = self._meta_ring[-1]
        else:
            slot = self._meta_ring[self._meta_slot]
            self._meta_slot = (self._meta_slot + 1) % self._META_SLOTS
            if slot[1] is not None:
                slot[1].synchronize()
            if slot[0].numel() < words:
                slot[0] = torch.empty(words, dtype=torch.int64).pin_memory()
        return slot

    def __len__(self) -> int:
        return self.length

    def batch(self, indices) -> PairData:
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        if idx.size == 0 or idx.min() < 0 or idx.max() >= self.length:
            raise IndexError(f"sample indices must lie in [0, {self.length})")
        B = int(idx.size)
        cnt = self.counts[idx]
        slot = self._meta_buffer(3 * B + 1)
        meta = slot[0][:3 * B + 1].numpy()
        meta[:B] = idx
        meta[B:2 * B] = self.offsets[idx]
        meta[2 * B] = 0
        np.cumsum(cnt, out=meta[2 * B + 1:])
        E = int(meta[3 * B])
        meta_d = slot[0][:3 * B + 1].to(self.device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(self.device))
        xp, y, ei, ea = ops.collate_samples(self.pos, self.rows, self.cols, meta_d, B, self.N, self.W, self.horizon, E,
                                            int(cnt.max()))
        if B not in self._aa_tiled:
            self._aa_tiled[B] = self.x_aminoacid.repeat(B)
        out = PairData(x_aminoacid=self._aa_tiled[B], x_position=xp, y=y, edge_attr=ea, edge_index=ei)
        out.num_graphs = B
        return out


def train_forward(model, data) -> torch.Tensor:
    """Differentiable forward of `KernelNN` for one sample, a list of samples, or an already collated batch
    (`collate`, `DeviceTrajectory.batch`) -> [B*N, out].  Nothing here waits for the device: index errors
    (amino-acid id or node id out of range — IndexError in the reference) are left in `model`'s training
    status word and raised by `check_train_status(model)`, which `train_epoch` calls once per epoch."""
    batch = collate(data) if not isinstance(data, PairData) else data
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise MdnoError("training needs the model on the GPU (model.to('cuda')); no CPU fallback")
    xp = batch.x_position.to(dev, torch.float32)
    if xp.dim() == 2:
        xp = xp.unsqueeze(0)
    W, R, _ = xp.shape
    aa = batch.x_aminoacid.to(dev)
    if getattr(model, "conv2", None) is not None and model.conv1.net is not model.conv2.net:
        raise NotImplementedError("training assumes the reference's single shared edge-MLP (graph_kernel.py:271-273)")
    # per-atom prologue (graph_kernel.py:279-298 with B=1 semantics per sample): HIP forward + backward.
    # The ParamPack holds device pointers to the parameters' CURRENT storage (fp32 contiguous parameters
    # are viewed, not copied), the tensors themselves are passed so that autograd routes their gradients.
    sd = dict(model.named_parameters())
    names = tuple(k for k in _PROLOGUE_KEYS if k in sd)
    pack = model.param_pack(dev, conv_mode="materialized")
    pack.prologue_names = names
    status = getattr(model, "_train_status", None)
    if status is None or status.device != dev:
        status = model._train_status = torch.zeros(1, dtype=torch.int32, device=dev)
    pack.train_status = status
    conv2 = getattr(model, "conv2", None)
    depth = model.depth if conv2 is not None else model.depth // 2
    # the feature stack of the kernel-integral block [2*depth+1, R, 64]: the prologue writes layer 0 in place and the
    # block returns its last layer as a view — no copy on either side
    X = torch.empty((2 * depth + 1, R, model.fc1.out_features), dtype=torch.float32, device=dev) \
        if model.fc1.out_features == 64 else None
    pack.prologue_out = X[0] if X is not None else None
    x0 = NodePrologue.apply(pack, xp.unsqueeze(1).contiguous(), aa, *[sd[k] for k in names])
    pack.prologue_out = None
    ei = batch.edge_index.to(dev)
    graph = ops.coo_to_csr(ei, R, validate=False, status=status)
    # the same edges grouped by source, for the input-gradient kernel: built now, next to the forward's sort
    # (ids already validated by it), so that the backward starts with everything in place
    graph.by_src = ops.source_sorted(graph, R, status=status) if torch.is_grad_enabled() else None
    graph.x_stack = X
    net = model.conv1.net
    w0, b0, w1, b1, w2, b2 = net.hip_weights()
    if conv2 is None and model.depth % 2:
        raise NotImplementedError("notebook-era variant: training needs an even depth")
    c2 = conv2 if conv2 is not None else model.conv1
    precision = getattr(model, "train_precision", "fp32")
    if precision not in ("fp32", "bf16"):
        raise MdnoError(f"train_precision={precision!r} (fp32, bf16)")
    if precision == "bf16" and (model.fc1.out_features != 64 or w1.shape[0] % 128 or w1.shape[1] % 32):
        raise NotImplementedError("bf16 training needs width 64 and ker_width a multiple of 128")
    x = KernelIntegralBlock.apply(x0, batch.edge_attr.to(dev), graph, depth,
                                  "bf16" if precision == "bf16" else getattr(model, "gemm_mode", "f32"),
                                  w0, b0, w1, b1, w2, b2,
                                  model.conv1.root, model.conv1.bias, c2.root, c2.bias)
    return FcOut.apply(x, model.fc2.weight, model.fc2.bias)


def check_train_status(model) -> None:
    """Read (one device->host word) and clear the status the training forwards since the last call left
    behind; raises what the reference's nn.Embedding / index_select would have raised in that forward."""
    from ._lib import raise_on_status
    st = getattr(model, "_train_status", None)
    if st is None:
        return
    word = int(st.item())
    if word:                     # (cleared only when something was raised: no fill launch per epoch otherwise)
        st.zero_()
    raise_on_status(word, "training forward")


class Adam(torch.optim.Optimizer):
    """`torch.optim.Adam(params, lr, betas, eps, weight_decay)` — the optimiser of the reference's main()
    (graph_kernel.py:541-543) — with its step as ONE libmdno launch over all parameter tensors (`mdno_adam_step`: the same
    arithmetic in the same order; torch's fused form walks tensor lists through multi_tensor_apply).  A
    `torch.optim.Optimizer`: lr schedulers (StepLR, :544-546) drive `param_groups[i]["lr"]`, and `state_dict()` has
    torch.optim.Adam's layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so a checkpoint written with either
    (graph_kernel.py:633-639) loads into the other.  fp32 parameters on the GPU; no amsgrad / maximize."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("Adam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_step = {}      # parameters of one group normally share their step count: one launch per distinct count
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32)                     # (host tensor, as torch.optim.Adam keeps it)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(int(st["step"]), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            b1, b2 = group["betas"]
            for t, items in by_step.items():
                ops.adam_step([i[0] for i in items], [i[1] for i in items], [i[2] for i in items], [i[3] for i in items],
                              group["lr"], b1, b2, group["eps"], group["weight_decay"], t)
        return loss


def train_epoch(model, batches, optimizer, loss_fn, batch_size: Optional[int] = None):
    """One pass over `batches` — an iterable of lists of PairData (as the reference's DataListLoader yields)
    or of collated batches (`DeviceTrajectory.batch`): returns (avg relative-L2 loss, avg MSE) like train()
    (graph_kernel.py:445-474).  The per-batch losses stay on the device until the pass is over (the
    reference's `l2.item()` per batch would stall the GPU once per step); they are then added on the host in
    double precision in batch order, which is what `avg_loss += l2.item()` does."""
    model.train()
    losses, mses = [], []
    one = None
    for batch in batches:
        if isinstance(batch, PairData):
            B = batch_size or getattr(batch, "num_graphs", 1)
        else:
            B = len(batch)
        optimizer.zero_grad()
        out = train_forward(model, batch)
        y = torch.cat([s.y for s in batch]).to(out.device) if not isinstance(batch, PairData) else batch.y.to(out.device)
        if hasattr(loss_fn, "rel_with_mse"):      # LpLoss: loss and the logged MSE from one pass (csrc/loss.hip)
            l2, mse = loss_fn.rel_with_mse(out.view(B, -1), y.view(B, -1))
        else:
            l2, mse = loss_fn(out.view(B, -1), y.view(B, -1)), F.mse_loss(out.detach(), y)
        if one is None or one.device != l2.device:
            one = torch.ones((), dtype=l2.dtype, device=l2.device)      # d loss / d loss, made once (not a fill per batch)
        l2.backward(one)
        optimizer.step()
        losses.append(l2.detach())
        mses.append(mse)
    check_train_status(model)
    n = len(losses)
    if n == 0:
        return 0.0, 0.0
    # ONE device->host copy once the pass is over (the values stacked on the device), added on the host in double
    # precision in batch order
    vals = torch.stack([v.reshape(()) for v in losses] + [v.detach().reshape(()) for v in mses]).double().cpu().tolist()
    tot, tot_mse = sum(vals[:n]), sum(vals[n:])
    return tot / n, tot_mse / n


def validate_epoch(model, batches, loss_fn, batch_size: Optional[int] = None):
    """One validation pass — `validate()` of the reference (graph_kernel.py:476-493): `model.eval()`, no autograd,
    `out = model(batch)` on every batch (lists of PairData or collated batches), returns (avg relative-L2 loss,
    avg MSE).  The forward is the inference path (B block-diagonal members of one `mdno_kernelnn_fwd`; nothing is
    kept for a backward, so a pass needs less memory than a training step); losses stay on the device until the
    pass is over and index errors are raised once, like `train_epoch`."""
    was_training = model.training
    model.eval()
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise MdnoError("validation needs the model on the GPU (model.to('cuda')); no CPU fallback")
    status = getattr(model, "_train_status", None)
    if status is None or status.device != dev:
        status = model._train_status = torch.zeros(1, dtype=torch.int32, device=dev)
    losses, mses = [], []
    try:
        with torch.no_grad():
            for batch in batches:
                if isinstance(batch, PairData):
                    B = batch_size or getattr(batch, "num_graphs", 1)
                    y = batch.y.to(dev)
                else:
                    B = len(batch)
                    y = torch.cat([s.y for s in batch]).to(dev)
                out = model(batch, _status=status)
                if hasattr(loss_fn, "rel_with_mse"):
                    l2, mse = loss_fn.rel_with_mse(out.view(B, -1), y.view(B, -1))
                else:
                    l2, mse = loss_fn(out.view(B, -1), y.view(B, -1)), F.mse_loss(out, y)
                losses.append(l2)
                mses.append(mse)
        check_train_status(model)
    finally:
        model.train(was_training)
    n = len(losses)
    if n == 0:
        return 0.0, 0.0
    vals = torch.stack([v.detach().reshape(()) for v in losses] + [v.detach().reshape(()) for v in mses]).double().cpu().tolist()
    tot, tot_mse = sum(vals[:n]), sum(vals[n:])      # (one copy; batch-order sums in double, as train_epoch)
    return tot / n, tot_mse / n
