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
        R = x0.shape[0]
        E = graph.edge_count()
        ea = ops.f32(edge_attr)[graph.perm[:E].long()] if graph.perm is not None else ops.f32(edge_attr)
        ea = ea.contiguous()
        ctx.bf16 = gemm_mode == "bf16"
        if ctx.bf16:
            return KernelIntegralBlock._forward_bf16(ctx, x0, ea, graph, depth, w0, b0, w1, b1, w2, b2, root1, bias1,
                                                     root2, bias2)
        h1 = ops.linear(ea, w0, b0, relu=True)
        h2 = ops.linear(h1, w1, b1, relu=True, gemm_mode=gemm_mode)
        w_e = ops.linear(h2, w2, b2, relu=False, gemm_mode=gemm_mode)
        L = 2 * depth
        X = torch.empty((L + 1, R, 64), dtype=torch.float32, device=x0.device)
        X[0].copy_(x0)
        for a in range(1, L + 1):
            root, bias = (root1, bias1) if a <= depth else (root2, bias2)
            X[a].copy_(ops.nnconv(X[a - 1], graph, w_e, root, bias, "mean", relu=True))
        ctx.graph, ctx.depth, ctx.gemm_mode = graph, depth, gemm_mode
        ctx.save_for_backward(ea, h1, h2, w_e, X, w0, w1, w2, root1, root2)
        return X[L].clone()

    @staticmethod
    def _forward_bf16(ctx, x0, ea, graph, depth, w0, b0, w1, b1, w2, b2, root1, bias1, root2, bias2):
        R = x0.shape[0]
        h1 = ops.linear_smallk_bf16(ea, w0, b0, relu=True)               # K = 6: fp32 fmaf chains, stored bf16
        h2 = ops.linear_bf16(h1, w1, b1, relu=True, out_bf16=True)
        w_e = ops.linear_bf16(h2, w2, b2, relu=False, out_bf16=True)     # [E, 4096] bf16: 8 KiB per edge
        L = 2 * depth
        X = torch.empty((L + 1, R, 64), dtype=torch.float32, device=x0.device)
        X[0].copy_(x0)
        for a in range(1, L + 1):
            root, bias = (root1, bias1) if a <= depth else (root2, bias2)
            X[a].copy_(ops.nnconv_bf16w(X[a - 1], graph, w_e, root, bias, "mean", relu=True))
        ctx.graph, ctx.depth, ctx.gemm_mode = graph, depth, "bf16"
        ctx.save_for_backward(ea, h1, h2, w_e, X, w0, w1, w2, root1, root2)
        return X[L].clone()

    @staticmethod
    def _backward_bf16(ctx, g_out):
        ea, h1, h2, w_e, X, w0, w1, w2, root1, root2 = ctx.saved_tensors
        graph, depth = ctx.graph, ctx.depth
        L, R = 2 * depth, X.shape[1]
        by_src = ops.source_sorted(graph, R)
        inv = ops.inv_degree(graph, "mean")
        GZ = torch.empty((L, R, 64), dtype=torch.float32, device=X.device)
        GS = torch.empty((L, R, 64), dtype=torch.float32, device=X.device)
        g = ops.f32(g_out)
        for a in range(L, 0, -1):
            ops.relu_bwd(g, X[a], None, out=GZ[a - 1])
            ops.relu_bwd(g, X[a], inv, out=GS[a - 1])
            g = ops.nnconv_bwd_x_bf16w(GZ[a - 1], GS[a - 1], by_src, w_e, root1 if a <= depth else root2)
        d_root1, d_bias1 = ops.nnconv_bwd_root(X[0:depth].reshape(-1, 64), GZ[0:depth].reshape(-1, 64))
        d_root2, d_bias2 = ops.nnconv_bwd_root(X[depth:L].reshape(-1, 64), GZ[depth:L].reshape(-1, 64))
        d_we = ops.nnconv_bwd_we_bf16(X[0:L], GS, graph)                 # bf16 [E, 4096]
        del GZ, GS
        d_b2 = ops.colsum_bf16(d_we)
        d_w2 = ops.gemm_atb_bf16(d_we, h2)
        gz2 = ops.relu_bwd_bf16(ops.linear_bf16(d_we, ops.transpose(w2), None, out_bf16=False), h2, out_bf16=True)
        del d_we
        d_b1 = ops.colsum_bf16(gz2)
        d_w1 = ops.gemm_atb_bf16(gz2, h1)
        gz1 = ops.relu_bwd_bf16(ops.linear_bf16(gz2, ops.transpose(w1), None, out_bf16=False), h1, out_bf16=True)
        d_b0 = ops.colsum_bf16(gz1)
        # d_w0 = gz1^T . ea with ea [E, 6]: the six attribute columns ride in a zero-padded 128-column bf16
        # operand of the same A^T.B kernel (the generic fp32 kernel took as long as the big products)
        ea_pad = torch.zeros((ea.shape[0], 128), dtype=torch.float32, device=ea.device)
        ea_pad[:, :ea.shape[1]].copy_(ea)
        d_w0 = ops.gemm_atb_bf16(gz1, ops.cast_bf16(ea_pad))[:, :ea.shape[1]].contiguous()
        return (g, None, None, None, None, d_w0, d_b0, d_w1, d_b1, d_w2, d_b2, d_root1, d_bias1, d_root2, d_bias2)

    @staticmethod
    def backward(ctx, g_out):
        if ctx.bf16:
            return KernelIntegralBlock._backward_bf16(ctx, g_out)
        ea, h1, h2, w_e, X, w0, w1, w2, root1, root2 = ctx.saved_tensors
        graph, depth, gemm_mode = ctx.graph, ctx.depth, ctx.gemm_mode
        L, R = 2 * depth, X.shape[1]
        by_src = ops.source_sorted(graph, R)
        inv = ops.inv_degree(graph, "mean")
        GZ = torch.empty((L, R, 64), dtype=torch.float32, device=X.device)
        GS = torch.empty((L, R, 64), dtype=torch.float32, device=X.device)
        g = ops.f32(g_out)
        for a in range(L, 0, -1):
            ops.relu_bwd(g, X[a], None, out=GZ[a - 1])
            ops.relu_bwd(g, X[a], inv, out=GS[a - 1])
            g = ops.nnconv_bwd_x(GZ[a - 1], GS[a - 1], by_src, w_e, root1 if a <= depth else root2)
        d_root1, d_bias1 = ops.nnconv_bwd_root(X[0:depth].reshape(-1, 64), GZ[0:depth].reshape(-1, 64))
        d_root2, d_bias2 = ops.nnconv_bwd_root(X[depth:L].reshape(-1, 64), GZ[depth:L].reshape(-1, 64))
        d_we = ops.nnconv_bwd_we(X[0:L], GS, graph)
        del GZ, GS
        # edge-MLP backward
        d_b2 = ops.colsum(d_we)
        d_w2 = ops.gemm_atb(d_we, h2)
        gz2 = ops.relu_bwd(ops.linear(d_we, ops.transpose(w2), None, gemm_mode=gemm_mode), h2)
        del d_we
        d_b1 = ops.colsum(gz2)
        d_w1 = ops.gemm_atb(gz2, h1)
        gz1 = ops.relu_bwd(ops.linear(gz2, ops.transpose(w1), None, gemm_mode=gemm_mode), h1)
        d_b0 = ops.colsum(gz1)
        d_w0 = ops.gemm_atb(gz1, ea)
        return (g, None, None, None, None, d_w0, d_b0, d_w1, d_b1, d_w2, d_b2, d_root1, d_bias1, d_root2, d_bias2)


class NodePrologue(torch.autograd.Function):
    """x0 = relu(fc1([emb(aa), lstm_fc(LSTM over the window)]))  (graph_kernel.py:279-298), B=1 semantics per
    sample.  Parameters arrive as tensors (autograd tracks them) and as the model's ParamPack (device pointers)."""

    @staticmethod
    def forward(ctx, pack, frames, aa, *params):
        x0 = ops.node_prologue(pack, frames, aa)
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
    return b


def train_forward(model, data) -> torch.Tensor:
    """Differentiable forward of `KernelNN` for one sample or a list/batch of samples -> [B*N, out]."""
    batch = collate(data) if not isinstance(data, PairData) else data
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise MdnoError("training needs the model on the GPU (model.to('cuda')); no CPU fallback")
    xp = batch.x_position.to(dev, torch.float32)
    if xp.dim() == 2:
        xp = xp.unsqueeze(0)
    W, R, _ = xp.shape
    aa = batch.x_aminoacid.to(dev)
    if model.conv1.net is not model.conv2.net:
        raise NotImplementedError("training assumes the reference's single shared edge-MLP (graph_kernel.py:271-273)")
    # per-atom prologue (graph_kernel.py:279-298 with B=1 semantics per sample): HIP forward + backward.
    # The ParamPack holds device pointers to the parameters' CURRENT storage (fp32 contiguous parameters
    # are viewed, not copied), the tensors themselves are passed so that autograd routes their gradients.
    sd = dict(model.named_parameters())
    names = tuple(k for k in _PROLOGUE_KEYS if k in sd)
    pack = model.param_pack(dev, conv_mode="materialized")
    pack.prologue_names = names
    x0 = NodePrologue.apply(pack, xp.unsqueeze(1).contiguous(), aa, *[sd[k] for k in names])
    graph = ops.coo_to_csr(batch.edge_index.to(dev), R)
    net = model.conv1.net
    w0, b0, w1, b1, w2, b2 = net.hip_weights()
    conv2 = getattr(model, "conv2", None)
    depth = model.depth if conv2 is not None else model.depth // 2
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


def train_epoch(model, batches, optimizer, loss_fn, batch_size: Optional[int] = None):
    """One pass over `batches` (an iterable of lists of PairData, as the reference's DataListLoader
    yields): returns (avg relative-L2 loss, avg MSE) like train() (graph_kernel.py:445-474)."""
    model.train()
    tot, tot_mse, n = 0.0, 0.0, 0
    for batch in batches:
        B = len(batch) if not isinstance(batch, PairData) else (batch_size or 1)
        optimizer.zero_grad()
        out = train_forward(model, batch)
        y = torch.cat([s.y for s in batch]).to(out.device) if not isinstance(batch, PairData) else batch.y.to(out.device)
        l2 = loss_fn(out.view(B, -1), y.view(B, -1))
        l2.backward()
        optimizer.step()
        tot += float(l2.item())
        tot_mse += float(F.mse_loss(out.detach(), y).item())
        n += 1
    return tot / max(n, 1), tot_mse / max(n, 1)
