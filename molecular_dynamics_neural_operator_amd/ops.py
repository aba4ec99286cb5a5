"""Tensor-level wrappers over the C ABI (include/mdno.h).  PyTorch is used for device memory and
streams only; every computation below is a libmdno HIP kernel.  All tensors must be CUDA(HIP)
tensors — there is no CPU path."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import AGGR, KernelNNParams, MdnoError, check, f32, ptr, raise_on_status, stream_ptr


@dataclass
class CSRGraph:
    """Destination-sorted CSR (see mdno.h): row r lists the sources of its in-edges."""
    row_ptr: torch.Tensor            # i32 [R+1]
    src: torch.Tensor                # i32 [cap]
    dst: torch.Tensor                # i32 [cap]
    num_edges: torch.Tensor          # i32 [1] (device)
    edge_cap: int
    perm: Optional[torch.Tensor] = None   # i32 [E]: CSR position p holds input edge perm[p]
    status: Optional[torch.Tensor] = None
    n_edges: Optional[int] = None         # the edge count when the host knows it (COO input): no device read

    def edge_count(self) -> int:
        return self.n_edges if self.n_edges is not None else int(self.num_edges.item())

    def to_edge_index(self) -> torch.Tensor:
        """Reference-order COO `[rows; cols]` (graph_kernel.py:368) — valid for radius graphs, whose
        contact map is symmetric, so (dst, src) read in CSR order IS the row-major COO."""
        e = self.edge_count()
        return torch.stack([self.dst[:e], self.src[:e]]).to(torch.long)


def radius_graph(pos: torch.Tensor, n_atoms: int, cutoff: float = 8.0, edge_cap: Optional[int] = None,
                 cell_list: bool = True) -> CSRGraph:
    """pos f32 [M*N,3] (or [M,N,3]) -> CSRGraph.  Replaces graph_kernel.py:363-368.  Members of >= 8,192 atoms go
    through a cell list (`cell_list=False`: the N^2 pair tests; the same graph, bit for bit)."""
    lib = _lib.load()
    pos = f32(pos).reshape(-1, 3)
    R = pos.shape[0]
    if R % n_atoms:
        raise MdnoError(f"{R} rows is not a multiple of n_atoms={n_atoms}")
    M = R // n_atoms
    cap = int(edge_cap) if edge_cap is not None else M * n_atoms * n_atoms
    cap = max(cap, R)
    dev = pos.device
    row_ptr = torch.empty(R + 1, dtype=torch.int32, device=dev)
    src = torch.empty(cap, dtype=torch.int32, device=dev)
    dst = torch.empty(cap, dtype=torch.int32, device=dev)
    ne = torch.zeros(1, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    nbytes = lib.mdno_radius_graph_workspace_bytes(M, n_atoms) if cell_list else 0      # > 0: large members, cell list
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    check(lib.mdno_radius_graph_csr_ws(ptr(pos), M, n_atoms, float(cutoff), ptr(row_ptr), ptr(src), ptr(dst), cap,
                                       ptr(ne), ptr(status), ptr(ws), nbytes, stream_ptr(dev)), "mdno_radius_graph_csr_ws")
    return CSRGraph(row_ptr, src, dst, ne, cap, None, status)


def coo_to_csr(edge_index: torch.Tensor, num_nodes: int, validate: bool = True,
               status: Optional[torch.Tensor] = None) -> CSRGraph:
    """edge_index i64 [2,E] (row 0 = source, row 1 = target) -> CSRGraph with `perm`.  A node id
    outside [0, num_nodes) raises (as the reference's gather / scatter do); `validate=False` defers
    that check: the bit stays in `graph.status` (`status` if given: a device word the kernels OR into)
    for the caller to read after its own work — no host synchronisation here."""
    lib = _lib.load()
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise MdnoError(f"edge_index must be [2,E], got {tuple(edge_index.shape)}")
    ei = edge_index.to(torch.long).contiguous()
    E = ei.shape[1]
    dev = ei.device
    row_ptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    cap = max(E, 1)
    src = torch.empty(cap, dtype=torch.int32, device=dev)
    dst = torch.empty(cap, dtype=torch.int32, device=dev)
    perm = torch.empty(cap, dtype=torch.int32, device=dev)
    nbytes = lib.mdno_coo_to_csr_workspace_bytes(E, num_nodes)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    ne = torch.empty(1, dtype=torch.int32, device=dev)       # written by the sort itself (no fill launch)
    check(lib.mdno_coo_to_csr(ptr(ei), E, num_nodes, ptr(row_ptr), ptr(src), ptr(dst), ptr(perm), ptr(ne), ptr(status),
                              ptr(ws), nbytes, stream_ptr(dev)), "mdno_coo_to_csr")
    if validate:
        raise_on_status(status.item(), "coo_to_csr")
    return CSRGraph(row_ptr, src, dst, ne, cap, perm, status, n_edges=E)


def edge_mlp(weights, ker_in: int, ker_width: int, out_dim: int, graph: CSRGraph,
             edge_pos: Optional[torch.Tensor] = None, edge_attr: Optional[torch.Tensor] = None,
             gemm_mode: str = "split_bf16") -> torch.Tensor:
    """W_e f32 [cap, out_dim] in CSR edge order.  `weights` = (w0,b0,w1,b1,w2,b2) torch Linear
    layout.  Attributes from `edge_pos` [R,3] + CSR, or `edge_attr` [E,ker_in] (+ graph.perm)."""
    lib = _lib.load()
    w = [f32(t) for t in weights]
    dev = w[0].device
    cap = graph.edge_cap
    w_e = torch.empty((cap, out_dim), dtype=torch.float32, device=dev)
    mode = _lib.GEMM_MODES[gemm_mode]
    nbytes = lib.mdno_edge_mlp_workspace_bytes(ker_width, out_dim, cap, mode)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ea = f32(edge_attr) if edge_attr is not None else None
    ep = f32(edge_pos).reshape(-1, 3) if edge_pos is not None else None
    check(lib.mdno_edge_mlp_fwd(ptr(ep), ptr(graph.src), ptr(graph.dst), ptr(ea),
                                ptr(graph.perm) if ea is not None else None, ptr(graph.num_edges), cap,
                                ker_in, ker_width, out_dim, mode, *[ptr(t) for t in w], ptr(w_e), ptr(ws), nbytes,
                                stream_ptr(dev)), "mdno_edge_mlp_fwd")
    return w_e


def nnconv(x: torch.Tensor, graph: CSRGraph, w_e: torch.Tensor, root: Optional[torch.Tensor],
           bias: Optional[torch.Tensor], aggr: str = "mean", relu: bool = False,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    if aggr not in AGGR:
        raise MdnoError(f"aggr={aggr!r} is not implemented by the HIP path (add, mean)")
    x = f32(x)
    R, cin = x.shape
    cout = w_e.shape[1] // cin
    y = out if out is not None else torch.empty((R, cout), dtype=torch.float32, device=x.device)
    root_c = f32(root) if root is not None else None
    bias_c = f32(bias) if bias is not None else None
    check(lib.mdno_nnconv_fwd(ptr(x), ptr(graph.row_ptr), ptr(graph.src), R, ptr(w_e), ptr(root_c), ptr(bias_c),
                              cin, cout, AGGR[aggr], int(relu), ptr(y), stream_ptr(x.device)), "mdno_nnconv_fwd")
    return y


class ParamPack:
    """Owns contiguous fp32 device copies (or views) of a KernelNN state_dict and the C struct
    pointing at them.  Keep it alive while kernels that use it are in flight."""

    KEYS = {
        "lstm_w_ih": "lstm.weight_ih_l0", "lstm_w_hh": "lstm.weight_hh_l0",
        "lstm_b_ih": "lstm.bias_ih_l0", "lstm_b_hh": "lstm.bias_hh_l0",
        "lstm_fc_w": "lstm_fc.weight", "lstm_fc_b": "lstm_fc.bias", "emb_w": "emb.weight",
        "fc1_w": "fc1.weight", "fc1_b": "fc1.bias",
        "k_w0": "conv1.net.layers.0.weight", "k_b0": "conv1.net.layers.0.bias",
        "k_w1": "conv1.net.layers.2.weight", "k_b1": "conv1.net.layers.2.bias",
        "k_w2": "conv1.net.layers.4.weight", "k_b2": "conv1.net.layers.4.bias",
        "k2_w0": "conv2.net.layers.0.weight", "k2_b0": "conv2.net.layers.0.bias",
        "k2_w1": "conv2.net.layers.2.weight", "k2_b1": "conv2.net.layers.2.bias",
        "k2_w2": "conv2.net.layers.4.weight", "k2_b2": "conv2.net.layers.4.bias",
        "conv1_root": "conv1.root", "conv1_bias": "conv1.bias",
        "conv2_root": "conv2.root", "conv2_bias": "conv2.bias",
        "fc2_w": "fc2.weight", "fc2_b": "fc2.bias",
    }

    def __init__(self, state_dict, depth: int, device, gemm_mode: str = "split_bf16",
                 conv_mode: str = "materialized"):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        self.tensors = {}
        p = KernelNNParams()
        for field, key in self.KEYS.items():
            if key not in sd:
                # optional groups: conv2's own kernel (shared otherwise), and — notebook-era model
                # (bba_analysis.ipynb:123-128) — the LSTM front-end and the whole conv2 block
                if field.startswith(("k2_", "lstm_", "conv2_")):
                    setattr(p, field, None)
                    continue
                raise MdnoError(f"state_dict lacks {key!r}")
            t = sd[key].detach().to(device=device, dtype=torch.float32).contiguous()
            self.tensors[field] = t
            setattr(p, field, t.data_ptr())
        # the reference shares ONE edge-MLP between conv1 and conv2 (graph_kernel.py:271-273):
        # tied storage or equal values -> evaluate once
        shared = all(
            ("k2_" + n) not in self.tensors
            or self.tensors["k2_" + n].data_ptr() == self.tensors["k_" + n].data_ptr()
            or torch.equal(self.tensors["k2_" + n], self.tensors["k_" + n])
            for n in ("w0", "b0", "w1", "b1", "w2", "b2"))
        if shared:
            for n in ("w0", "b0", "w1", "b1", "w2", "b2"):
                setattr(p, "k2_" + n, None)
        self.shared_kernel = shared
        width = self.tensors["fc1_w"].shape[0]
        p.width = width
        p.ker_width = self.tensors["k_w0"].shape[0]
        p.depth = int(depth)
        p.ker_in = self.tensors["k_w0"].shape[1]
        p.in_width = self.tensors["fc1_w"].shape[1]
        p.out_width = self.tensors["fc2_w"].shape[0]
        p.num_embeddings, p.embedding_dim = self.tensors["emb_w"].shape
        p.x_position_dim = self.tensors["lstm_w_ih"].shape[1] if "lstm_w_ih" in self.tensors else 3
        if ("lstm_w_ih" in self.tensors) != ("lstm_fc_w" in self.tensors) or \
                ("conv2_root" in self.tensors) != ("conv2_bias" in self.tensors):
            raise MdnoError("state_dict has a partial lstm/conv2 parameter group")
        p.gemm_mode = _lib.GEMM_MODES[gemm_mode]
        self.gemm_mode = gemm_mode
        p.conv_mode = _lib.CONV_MODES[conv_mode]
        self.conv_mode = conv_mode
        if self.tensors["k_w2"].shape[0] != width * width:
            raise MdnoError("edge-MLP output size != width**2")
        self.struct = p
        self.device = torch.device(device)

    @property
    def ref(self):
        return C.byref(self.struct)


def node_prologue(pack: ParamPack, frames: torch.Tensor, x_aminoacid: torch.Tensor,
                  status: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """frames f32 [W,M,N,3] -> x0 f32 [M*N,width]: LSTM over the window, lstm_fc, Embedding, concat, fc1,
    ReLU (graph_kernel.py:279-298).  Raises on an amino-acid id outside [0, num_embeddings) — unless the
    caller passes its own `status` word (int32 [1] on the device): the bit is then left there for it to
    read later and this call does not synchronise."""
    lib = _lib.load()
    frames = f32(frames)
    if frames.dim() == 3:
        frames = frames.unsqueeze(1)
    W, M, N, _ = frames.shape
    dev = frames.device
    aa = x_aminoacid.to(device=dev, dtype=torch.long).contiguous()
    if aa.numel() not in (N, M * N):
        raise MdnoError(f"x_aminoacid has {aa.numel()} entries, expected {N} or {M * N}")
    x0 = out if out is not None else torch.empty((M * N, pack.struct.width), dtype=torch.float32, device=dev)
    if tuple(x0.shape) != (M * N, pack.struct.width) or x0.dtype != torch.float32:
        raise MdnoError(f"node_prologue: out must be f32 {(M * N, pack.struct.width)}")
    deferred = status is not None
    if not deferred:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib.mdno_node_prologue_fwd(pack.ref, ptr(frames), M, W, N, ptr(aa), int(aa.numel() == M * N and M > 1),
                                     ptr(x0), ptr(status), stream_ptr(dev)), "mdno_node_prologue_fwd")
    if not deferred:
        raise_on_status(status.item(), "node_prologue")
    return x0


FALLBACK_KEYS = ("conv_k1_workgroups_rerun_bf16", "conv_destinations_unscaled", "edge_mlp_products_bf16")


def kernelnn_forward(pack: ParamPack, frames: torch.Tensor, x_aminoacid: torch.Tensor, graph: CSRGraph,
                     edge_pos: Optional[torch.Tensor] = None, edge_attr: Optional[torch.Tensor] = None,
                     return_latent: bool = False, workspace: Optional[torch.Tensor] = None,
                     check_status: bool = True, fallback_counts: Optional[dict] = None
                     ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """frames f32 [W,M,N,3] (time-major) -> out [M*N,out_width] (+ latent [M*N,width]).
    `fallback_counts`: a dict that receives which path gemm_mode "split_f16" took in this forward (FALLBACK_KEYS:
    all zero = every product on two fp16 planes; include/mdno.h mdno_kernelnn_fallback_counts) — reading it
    synchronises the stream.
    The device status word (bad amino-acid id, edge overflow, bad edge index) is read back and
    raised after the call — the reference's nn.Embedding raises IndexError at that point; pass
    `check_status=False` to keep the call asynchronous and read `graph.status` yourself."""
    lib = _lib.load()
    frames = f32(frames)
    if frames.dim() == 3:
        frames = frames.unsqueeze(1)
    W, M, N, _ = frames.shape
    dev = frames.device
    aa = x_aminoacid.to(device=dev, dtype=torch.long).contiguous()
    if aa.numel() not in (N, M * N):
        raise MdnoError(f"x_aminoacid has {aa.numel()} entries, expected {N} or {M * N}")
    aa_pm = int(aa.numel() == M * N and M > 1)
    p = pack.struct
    out = torch.empty((M * N, p.out_width), dtype=torch.float32, device=dev)
    latent = torch.empty((M * N, p.width), dtype=torch.float32, device=dev) if return_latent else None
    nbytes = lib.mdno_kernelnn_workspace_bytes(pack.ref, M, N, graph.edge_cap)
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if graph.status is None:
        graph.status = torch.zeros(1, dtype=torch.int32, device=dev)
    status = graph.status
    ea = f32(edge_attr) if edge_attr is not None else None
    ep = f32(edge_pos).reshape(-1, 3) if edge_pos is not None else None
    if ea is None and ep is None:
        raise MdnoError("kernelnn_forward needs edge_pos (a library-built radius graph) or edge_attr")
    check(lib.mdno_kernelnn_fwd(pack.ref, ptr(frames), M, W, N, ptr(aa), aa_pm, ptr(graph.row_ptr), ptr(graph.src),
                                ptr(graph.dst), ptr(graph.num_edges), graph.edge_cap, ptr(ep),
                                ptr(ea),
                                ptr(graph.perm) if ea is not None else None, ptr(out), ptr(latent), ptr(workspace),
                                workspace.numel(), ptr(status), stream_ptr(dev)), "mdno_kernelnn_fwd")
    if fallback_counts is not None:
        cnt = (C.c_int64 * 4)()
        check(lib.mdno_kernelnn_fallback_counts(pack.ref, M, N, graph.edge_cap, int(ep is not None and ea is None and graph.dst is not None),
                                                ptr(workspace), cnt, stream_ptr(dev)), "mdno_kernelnn_fallback_counts")
        fallback_counts.update({k: int(cnt[i]) for i, k in enumerate(FALLBACK_KEYS)})
    if check_status:
        raise_on_status(status.item(), "kernelnn_forward")
    return out, latent


class AdamTensor(C.Structure):
    """include/mdno.h mdno_adam_tensor"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_int64)]


def adam_step(params, grads, exp_avgs, exp_avg_sqs, lr: float, beta1: float, beta2: float, eps: float,
              weight_decay: float, step: int) -> None:
    """torch.optim.Adam's update (L2 weight decay, no amsgrad) of all the given fp32 tensors in ONE launch
    (include/mdno.h mdno_adam_step), in place, on the current stream."""
    n = len(params)
    if n == 0:
        return
    arr = (AdamTensor * n)()
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs)):
        if not (p.dtype == g.dtype == m.dtype == v.dtype == torch.float32):
            raise MdnoError("adam_step: fp32 tensors only")
        if not (p.numel() == g.numel() == m.numel() == v.numel()):
            raise MdnoError("adam_step: parameter, gradient and moments must have the same number of elements")
        arr[i] = AdamTensor(ptr(p), ptr(g), ptr(m), ptr(v), p.numel())
    check(_lib.load().mdno_adam_step(n, arr, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                     stream_ptr(params[0].device)), "mdno_adam_step")


# ------------------------------------------------------------------------------------------------
# Training ops (include/mdno.h "Training ops"): thin wrappers, torch only allocates the outputs.
def _ws(nbytes: int, dev) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def linear(a: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], relu: bool = False,
           gemm_mode: str = "f32") -> torch.Tensor:
    """act(a . w^T + b): a [rows,k], w [n,k] (torch Linear layout).  gemm_mode "split_bf16" runs the
    product on the bf16 matrix pipe with the exact 3-way split (fp32-level error), "split_f16" on two fp16
    planes per operand with every row scaled by its own power of two (3 products instead of 6, fp32-level
    error), where the shape tiles (k % 32 == 0, n % 128 == 0); other shapes and "f32" use the fp32 kernels."""
    lib = _lib.load()
    a, w = f32(a), f32(w)
    rows, k = a.shape
    n = w.shape[0]
    c = torch.empty((rows, n), dtype=torch.float32, device=a.device)
    bb = f32(b) if b is not None else None
    if gemm_mode == "split_f16" and k % 32 == 0 and n % 128 == 0 and rows > 0:
        ws = _ws(lib.mdno_linear_split_f16_workspace_bytes(rows, n, k), a.device)
        check(lib.mdno_linear_split_f16_fwd(ptr(a), ptr(w), ptr(bb), rows, n, k, int(relu), ptr(c), ptr(ws), ws.numel(),
                                            stream_ptr(a.device)), "mdno_linear_split_f16_fwd")
        return c
    if gemm_mode != "f32" and k % 32 == 0 and n % 128 == 0 and rows > 0:
        ws = _ws(lib.mdno_linear_split_workspace_bytes(rows, n, k), a.device)
        check(lib.mdno_linear_split_fwd(ptr(a), ptr(w), ptr(bb), rows, n, k, int(relu), ptr(c), ptr(ws), ws.numel(),
                                        stream_ptr(a.device)), "mdno_linear_split_fwd")
        return c
    check(lib.mdno_linear_fwd(ptr(a), ptr(w), ptr(bb), rows, n, k, int(relu), ptr(c), stream_ptr(a.device)),
          "mdno_linear_fwd")
    return c


def gemm_atb(a: torch.Tensor, b: torch.Tensor, gemm_mode: str = "f32") -> torch.Tensor:
    """a^T . b over rows: a [rows,n1], b [rows,n2] -> [n1,n2] (fixed-order partial sums).  gemm_mode "split_f16":
    on two fp16 planes per operand, columns scaled by powers of two (fp32-level error), where the shape tiles
    (n1, n2 multiples of 256); otherwise the exact fp32 MFMA."""
    lib = _lib.load()
    a, b = f32(a), f32(b)
    rows, n1 = a.shape
    n2 = b.shape[1]
    c = torch.empty((n1, n2), dtype=torch.float32, device=a.device)
    if gemm_mode == "split_f16" and lib.mdno_gemm_atb_split_f16_supported(rows, n1, n2):
        ws = _ws(lib.mdno_gemm_atb_split_f16_workspace_bytes(rows, n1, n2), a.device)
        check(lib.mdno_gemm_atb_split_f16(ptr(a), ptr(b), rows, n1, n2, ptr(c), 0, ptr(ws), ws.numel(), stream_ptr(a.device)),
              "mdno_gemm_atb_split_f16")
        return c
    nb = lib.mdno_reduce_workspace_bytes(n1, n2)
    ws = _ws(nb, a.device)
    check(lib.mdno_gemm_atb(ptr(a), ptr(b), rows, n1, n2, ptr(c), 0, ptr(ws), ws.numel(), stream_ptr(a.device)),
          "mdno_gemm_atb")
    return c


def colsum(a: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    a = f32(a)
    rows, n = a.shape
    out = torch.empty(n, dtype=torch.float32, device=a.device)
    ws = _ws(lib.mdno_reduce_workspace_bytes(n, 1), a.device)
    check(lib.mdno_colsum(ptr(a), rows, n, ptr(out), 0, ptr(ws), ws.numel(), stream_ptr(a.device)), "mdno_colsum")
    return out


def relu_bwd(g: torch.Tensor, y: torch.Tensor, row_scale: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    g, y = f32(g), f32(y)
    rows, n = g.shape
    if out is None:
        out = torch.empty_like(g)
    check(lib.mdno_relu_bwd(ptr(g), ptr(y), ptr(row_scale), rows, n, ptr(out), stream_ptr(g.device)), "mdno_relu_bwd")
    return out


def relu_bwd2(g: torch.Tensor, y: torch.Tensor, row_scale: torch.Tensor, gz: torch.Tensor, gs: torch.Tensor) -> None:
    """gz = g * (y > 0), gs = gz * row_scale[row] (both written in place)."""
    lib = _lib.load()
    g, y = f32(g), f32(y)
    rows, n = g.shape
    check(lib.mdno_relu_bwd2(ptr(g), ptr(y), ptr(row_scale), rows, n, ptr(gz), ptr(gs), stream_ptr(g.device)),
          "mdno_relu_bwd2")


def transpose(a: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    a = f32(a)
    r, c = a.shape
    at = torch.empty((c, r), dtype=torch.float32, device=a.device)
    check(lib.mdno_transpose(ptr(a), r, c, ptr(at), stream_ptr(a.device)), "mdno_transpose")
    return at


def inv_degree(graph: CSRGraph, aggr: str = "mean") -> torch.Tensor:
    lib = _lib.load()
    rows = graph.row_ptr.numel() - 1
    inv = torch.empty(rows, dtype=torch.float32, device=graph.row_ptr.device)
    check(lib.mdno_inv_degree(ptr(graph.row_ptr), rows, AGGR[aggr], ptr(inv), stream_ptr(inv.device)), "mdno_inv_degree")
    return inv


def source_sorted(graph: CSRGraph, num_nodes: int, status: Optional[torch.Tensor] = None) -> CSRGraph:
    """The same edges grouped by SOURCE (include/mdno.h mdno_csr_by_source): row_ptr over sources, `src` field =
    destination of each out-edge, `perm` = the edge's position in the destination-sorted arrays (and in W_e)."""
    lib = _lib.load()
    E = graph.edge_count()
    dev = graph.row_ptr.device
    cap = max(E, 1)
    row_ptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    nbr = torch.empty(cap, dtype=torch.int32, device=dev)
    rowid = torch.empty(cap, dtype=torch.int32, device=dev)
    perm = torch.empty(cap, dtype=torch.int32, device=dev)
    nbytes = lib.mdno_coo_to_csr_workspace_bytes(E, num_nodes)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = status if status is not None else graph.status        # (ids come from a CSR that was validated when built)
    check(lib.mdno_csr_by_source(ptr(graph.src), ptr(graph.dst), E, num_nodes, ptr(row_ptr), ptr(nbr), ptr(rowid), ptr(perm),
                                 ptr(st), ptr(ws), nbytes, stream_ptr(dev)), "mdno_csr_by_source")
    return CSRGraph(row_ptr, nbr, rowid, graph.num_edges, cap, perm, st, n_edges=E)


def permute_rows(x: torch.Tensor, perm: torch.Tensor, rows: int) -> torch.Tensor:
    """out[p] = x[perm[p]] for p < rows (include/mdno.h mdno_permute_rows): per-edge rows into a graph's CSR order."""
    lib = _lib.load()
    x = f32(x)
    out = torch.empty((rows,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    width = int(x[0].numel()) if x.shape[0] else 1
    check(lib.mdno_permute_rows(ptr(x), ptr(perm), rows, width, ptr(out), stream_ptr(x.device)), "mdno_permute_rows")
    return out


def nnconv_bwd_x(gz: torch.Tensor, gs: torch.Tensor, by_src: CSRGraph, w_e: torch.Tensor,
                 root: Optional[torch.Tensor]) -> torch.Tensor:
    lib = _lib.load()
    rows = gz.shape[0]
    g_prev = torch.empty_like(gz)
    check(lib.mdno_nnconv_bwd_x(ptr(gz), ptr(gs), ptr(by_src.row_ptr), ptr(by_src.perm), ptr(by_src.src), rows,
                                ptr(w_e), ptr(f32(root)) if root is not None else None, 64, 64, ptr(g_prev),
                                stream_ptr(gz.device)), "mdno_nnconv_bwd_x")
    return g_prev


def nnconv_bwd_root(x: torch.Tensor, gz: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x, gz [rows,64] (layers stacked along rows) -> (d_root [64,64], d_bias [64])."""
    lib = _lib.load()
    x, gz = f32(x), f32(gz)
    rows = x.shape[0]
    d_root = torch.empty((64, 64), dtype=torch.float32, device=x.device)
    d_bias = torch.empty(64, dtype=torch.float32, device=x.device)
    ws = _ws(lib.mdno_nnconv_bwd_root_workspace_bytes(rows), x.device)
    check(lib.mdno_nnconv_bwd_root(ptr(x), ptr(gz), rows, 64, 64, ptr(d_root), ptr(d_bias), 0, ptr(ws), ws.numel(),
                                   stream_ptr(x.device)), "mdno_nnconv_bwd_root")
    return d_root, d_bias


def nnconv_bwd_root_pair(x_layers: torch.Tensor, gz_layers: torch.Tensor):
    """x_layers, gz_layers [2*depth, R, 64] (conv1's applications first, then conv2's) ->
    (d_root1, d_bias1, d_root2, d_bias2): both convs' root / bias gradients from one launch, bitwise what two
    `nnconv_bwd_root` calls on the halves return."""
    lib = _lib.load()
    x, gz = f32(x_layers), f32(gz_layers)
    L, R, _ = x.shape
    assert L % 2 == 0 and gz.shape == x.shape
    rows_each = (L // 2) * R
    outs = [torch.empty(sh, dtype=torch.float32, device=x.device) for sh in ((64, 64), (64,), (64, 64), (64,))]
    ws = _ws(lib.mdno_nnconv_bwd_root_pair_workspace_bytes(rows_each), x.device)
    check(lib.mdno_nnconv_bwd_root_pair(ptr(x), ptr(gz), rows_each, ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]),
                                        ptr(ws), ws.numel(), stream_ptr(x.device)), "mdno_nnconv_bwd_root_pair")
    return tuple(outs)


def nnconv_bwd_we(x_layers: torch.Tensor, gs_layers: torch.Tensor, graph: CSRGraph, with_colsum: bool = False):
    """x_layers, gs_layers [L,R,64] -> d_we [E,4096]; `with_colsum`: also its column sums from the same pass (up to 16
    conv applications: the matrix-pipe kernel; beyond, the FMA kernel and `ops.colsum`)."""
    lib = _lib.load()
    L, R, _ = x_layers.shape
    e = graph.edge_count()
    d_we = torch.empty((e, 4096), dtype=torch.float32, device=x_layers.device)
    if with_colsum and L <= 16 and e > 0:
        cs = torch.empty(4096, dtype=torch.float32, device=x_layers.device)
        nb = lib.mdno_nnconv_bwd_we_colsum_workspace_bytes()
        ws = _ws(nb, x_layers.device)
        check(lib.mdno_nnconv_bwd_we_colsum(ptr(x_layers), ptr(gs_layers), ptr(graph.src), ptr(graph.dst), e, L, R * 64,
                                            ptr(d_we), ptr(cs), ptr(ws), nb, stream_ptr(d_we.device)), "mdno_nnconv_bwd_we_colsum")
        return d_we, cs
    check(lib.mdno_nnconv_bwd_we(ptr(x_layers), ptr(gs_layers), ptr(graph.src), ptr(graph.dst), e, L, R * 64, 64, 64,
                                 ptr(d_we), 0, stream_ptr(d_we.device)), "mdno_nnconv_bwd_we")
    return (d_we, colsum(d_we)) if with_colsum else d_we


# ------------------------------------------------------------------------------------------------
# Training ops, bf16 (include/mdno.h "Training ops, bf16"): torch only allocates; dtype torch.bfloat16
# tensors are the bf16 buffers of the C ABI.
def _bf16(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.bfloat16:
        raise MdnoError(f"expected a bfloat16 tensor, got {t.dtype}")
    return t.contiguous()


def cast_bf16(a: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    a = f32(a)
    if a.numel() % 4:
        raise MdnoError("cast_bf16: element count must be a multiple of 4")
    out = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device)
    check(lib.mdno_cast_bf16(ptr(a), a.numel(), ptr(out), stream_ptr(a.device)), "mdno_cast_bf16")
    return out


def linear_smallk_bf16(a: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], relu: bool = False) -> torch.Tensor:
    """bf16(act(a . w^T + b)) for the first edge-MLP layer: a fp32 [rows,k<=8] (edge attributes), w fp32 [n,k]."""
    lib = _lib.load()
    a, w = f32(a), f32(w)
    rows, k = a.shape
    n = w.shape[0]
    if not (1 <= k <= 8 and n % 8 == 0):        # (other shapes: the fp32 kernel, then the cast)
        return cast_bf16(linear(a, w, b, relu=relu))
    c = torch.empty((rows, n), dtype=torch.bfloat16, device=a.device)
    check(lib.mdno_linear_smallk_bf16_fwd(ptr(a), ptr(w), ptr(f32(b)) if b is not None else None, rows, n, k, int(relu),
                                          ptr(c), stream_ptr(a.device)), "mdno_linear_smallk_bf16_fwd")
    return c


def linear_bf16(a: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], relu: bool = False,
                out_bf16: bool = True) -> torch.Tensor:
    """act(a . w^T + b): a bf16 [rows,k], w fp32 master [n,k] (cast per call), fp32 accumulation."""
    lib = _lib.load()
    a, w = _bf16(a), f32(w)
    rows, k = a.shape
    n = w.shape[0]
    c = torch.empty((rows, n), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=a.device)
    ws = _ws(lib.mdno_linear_bf16_workspace_bytes(n, k), a.device)
    check(lib.mdno_linear_bf16_fwd(ptr(a), ptr(w), ptr(f32(b)) if b is not None else None, rows, n, k, int(relu),
                                   int(out_bf16), ptr(c), ptr(ws), ws.numel(), stream_ptr(a.device)), "mdno_linear_bf16_fwd")
    return c


def linear_bf16_relu_bwd(g: torch.Tensor, w_t: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """bf16((y > 0) * (g . w_t^T)): the input gradient of a Linear + ReLU layer whose stored (bf16) output is y —
    g bf16 [rows,k], w_t fp32 [n,k] (the layer's weight, transposed), y bf16 [rows,n].  One GEMM with the mask in its
    epilogue where the shape tiles (n % 256 == 0, k % 32 == 0); otherwise the GEMM, then mdno_relu_bwd_bf16 —
    the same fp32 accumulation, mask and single rounding either way."""
    lib = _lib.load()
    g, w_t, y = _bf16(g), f32(w_t), _bf16(y)
    rows, k = g.shape
    n = w_t.shape[0]
    if not lib.mdno_linear_bf16_masked_supported(rows, n, k):
        return relu_bwd_bf16(linear_bf16(g, w_t, None, out_bf16=False), y, out_bf16=True)
    c = torch.empty((rows, n), dtype=torch.bfloat16, device=g.device)
    ws = _ws(lib.mdno_linear_bf16_workspace_bytes(n, k), g.device)
    check(lib.mdno_linear_bf16_masked(ptr(g), ptr(w_t), ptr(y), rows, n, k, ptr(c), ptr(ws), ws.numel(),
                                      stream_ptr(g.device)), "mdno_linear_bf16_masked")
    return c


def gemm_atb_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    a, b = _bf16(a), _bf16(b)
    rows, n1 = a.shape
    n2 = b.shape[1]
    c = torch.empty((n1, n2), dtype=torch.float32, device=a.device)
    ws = _ws(lib.mdno_gemm_atb_bf16_workspace_bytes(n1, n2), a.device)
    check(lib.mdno_gemm_atb_bf16(ptr(a), ptr(b), rows, n1, n2, ptr(c), ptr(ws), ws.numel(), stream_ptr(a.device)),
          "mdno_gemm_atb_bf16")
    return c


def nnconv_bf16w(x: torch.Tensor, graph: CSRGraph, w_e: torch.Tensor, root, bias, aggr: str = "mean",
                 relu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    x, w_e = f32(x), _bf16(w_e)
    if x.shape[1] != 64 or w_e.shape[1] != 4096:
        raise MdnoError("nnconv_bf16w: width 64 only")
    y = out if out is not None else torch.empty_like(x)
    check(lib.mdno_nnconv_bf16w_fwd(ptr(x), ptr(graph.row_ptr), ptr(graph.src), x.shape[0], ptr(w_e),
                                    ptr(f32(root)) if root is not None else None,
                                    ptr(f32(bias)) if bias is not None else None, AGGR[aggr], int(relu), ptr(y),
                                    stream_ptr(x.device)), "mdno_nnconv_bf16w_fwd")
    return y


def nnconv_bwd_x_bf16w(gz: torch.Tensor, gs: torch.Tensor, by_src: CSRGraph, w_e: torch.Tensor, root) -> torch.Tensor:
    lib = _lib.load()
    g_prev = torch.empty_like(gz)
    check(lib.mdno_nnconv_bwd_x_bf16w(ptr(gz), ptr(gs), ptr(by_src.row_ptr), ptr(by_src.perm), ptr(by_src.src),
                                      gz.shape[0], ptr(_bf16(w_e)), ptr(f32(root)) if root is not None else None,
                                      ptr(g_prev), stream_ptr(gz.device)), "mdno_nnconv_bwd_x_bf16w")
    return g_prev


def nnconv_bwd_we_bf16(x_layers: torch.Tensor, gs_layers: torch.Tensor, graph: CSRGraph, with_colsum: bool = False):
    """d_we bf16 [E,4096]; `with_colsum`: also its column sums (fp32 [4096], the sums of the rounded values) from the
    same pass — up to 16 conv applications (depth <= 8); beyond that the plain kernel and `colsum_bf16`."""
    lib = _lib.load()
    L, R, _ = x_layers.shape
    e = graph.edge_count()
    d_we = torch.empty((e, 4096), dtype=torch.bfloat16, device=x_layers.device)
    if with_colsum and L <= 16 and e > 0:
        cs = torch.empty(4096, dtype=torch.float32, device=x_layers.device)
        nb = lib.mdno_nnconv_bwd_we_bf16_colsum_workspace_bytes()
        ws = _ws(nb, x_layers.device)
        check(lib.mdno_nnconv_bwd_we_bf16_colsum(ptr(x_layers), ptr(gs_layers), ptr(graph.src), ptr(graph.dst), e, L, R * 64,
                                                 ptr(d_we), ptr(cs), ptr(ws), nb, stream_ptr(d_we.device)),
              "mdno_nnconv_bwd_we_bf16_colsum")
        return d_we, cs
    check(lib.mdno_nnconv_bwd_we_bf16(ptr(x_layers), ptr(gs_layers), ptr(graph.src), ptr(graph.dst), e, L, R * 64,
                                      ptr(d_we), stream_ptr(d_we.device)), "mdno_nnconv_bwd_we_bf16")
    return (d_we, colsum_bf16(d_we)) if with_colsum else d_we


def relu_bwd_bf16(g: torch.Tensor, y: torch.Tensor, out_bf16: bool = True) -> torch.Tensor:
    lib = _lib.load()
    g, y = f32(g), _bf16(y)
    rows, n = g.shape
    out = torch.empty((rows, n), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=g.device)
    check(lib.mdno_relu_bwd_bf16(ptr(g), ptr(y), rows, n, int(out_bf16), ptr(out), stream_ptr(g.device)),
          "mdno_relu_bwd_bf16")
    return out


def colsum_bf16(a: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    a = _bf16(a)
    rows, n = a.shape
    out = torch.empty(n, dtype=torch.float32, device=a.device)
    ws = _ws(lib.mdno_colsum_bf16_workspace_bytes(n), a.device)
    check(lib.mdno_colsum_bf16(ptr(a), rows, n, ptr(out), ptr(ws), ws.numel(), stream_ptr(a.device)), "mdno_colsum_bf16")
    return out


def nnconv_chain_fwd(x_layers: torch.Tensor, graph: CSRGraph, w_e: torch.Tensor, root1, bias1, root2, bias2,
                     depth: int) -> None:
    """x_layers f32 [2*depth+1, R, 64]: [0] given, [a] = relu(conv(x[a-1])) written in place — the 2*depth conv
    applications of the block in one call (w_e fp32 or bf16 [E,4096])."""
    lib = _lib.load()
    R = x_layers.shape[1]
    args = (ptr(x_layers), ptr(graph.row_ptr), ptr(graph.src), R, ptr(w_e), ptr(f32(root1)), ptr(f32(bias1)),
            ptr(f32(root2)), ptr(f32(bias2)), int(depth), stream_ptr(x_layers.device))
    if w_e.dtype == torch.bfloat16:
        check(lib.mdno_nnconv_chain_bf16w_fwd(*args), "mdno_nnconv_chain_bf16w_fwd")
    else:
        check(lib.mdno_nnconv_chain_fwd(*args), "mdno_nnconv_chain_fwd")


def nnconv_chain_bwd(g_out: torch.Tensor, x_layers: torch.Tensor, inv_deg: torch.Tensor, by_src: CSRGraph,
                     w_e: torch.Tensor, root1, root2, depth: int):
    """Backward through the 2*depth applications -> (gz [L,R,64], gs [L,R,64], g_in [R,64])."""
    lib = _lib.load()
    L, R = 2 * depth, x_layers.shape[1]
    dev = x_layers.device
    gz = torch.empty((L, R, 64), dtype=torch.float32, device=dev)
    gs = torch.empty((L, R, 64), dtype=torch.float32, device=dev)
    g_in = torch.empty((R, 64), dtype=torch.float32, device=dev)
    args = (ptr(f32(g_out)), ptr(x_layers), ptr(inv_deg), ptr(by_src.row_ptr), ptr(by_src.perm), ptr(by_src.src), R,
            ptr(w_e), ptr(f32(root1)), ptr(f32(root2)), int(depth), ptr(gz), ptr(gs), ptr(g_in), stream_ptr(dev))
    if w_e.dtype == torch.bfloat16:
        check(lib.mdno_nnconv_chain_bf16w_bwd(*args), "mdno_nnconv_chain_bf16w_bwd")
    else:
        check(lib.mdno_nnconv_chain_bwd(*args), "mdno_nnconv_chain_bwd")
    return gz, gs, g_in


def colsum_atb_bf16(a: torch.Tensor, b: torch.Tensor):
    """(column sums of a bf16 [rows,n], a^T . b for b fp32 [rows,6 or 8]) in one pass over a -> ([n], [n,kb]);
    other shapes: the two separate ops."""
    lib = _lib.load()
    a, b = _bf16(a), f32(b)
    rows, n = a.shape
    kb = b.shape[1]
    if kb not in (6, 8) or n % 8:
        pad = torch.zeros((rows, 128), dtype=torch.float32, device=a.device)
        pad[:, :kb].copy_(b)
        return colsum_bf16(a), gemm_atb_bf16(a, cast_bf16(pad))[:, :kb].contiguous()
    colsum = torch.empty(n, dtype=torch.float32, device=a.device)
    atb = torch.empty((n, kb), dtype=torch.float32, device=a.device)
    ws = _ws(lib.mdno_colsum_atb_bf16_workspace_bytes(n, kb), a.device)
    check(lib.mdno_colsum_atb_bf16(ptr(a), ptr(b), rows, n, kb, ptr(colsum), ptr(atb), ptr(ws), ws.numel(),
                                   stream_ptr(a.device)), "mdno_colsum_atb_bf16")
    return colsum, atb


# ------------------------------------------------------------------------------------------------
# Per-atom ends (node prologue, fc2) forward + backward for training (include/mdno.h, csrc/train_nodes.hip)
def fc_out(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """x . w^T + b  (fc2, graph_kernel.py:305): x [rows,width], w [out_width,width]."""
    lib = _lib.load()
    x, w = f32(x), f32(w)
    rows, width = x.shape
    ow = w.shape[0]
    out = torch.empty((rows, ow), dtype=torch.float32, device=x.device)
    check(lib.mdno_fc_out_fwd(ptr(x), ptr(w), ptr(f32(b)) if b is not None else None, rows, width, ow, ptr(out),
                              stream_ptr(x.device)), "mdno_fc_out_fwd")
    return out


def fc_out_bwd(x: torch.Tensor, w: torch.Tensor, g: torch.Tensor):
    """-> (dx [rows,width], d_w [out_width,width], d_b [out_width])"""
    lib = _lib.load()
    x, w, g = f32(x), f32(w), f32(g)
    rows, width = x.shape
    ow = w.shape[0]
    dx = torch.empty_like(x)
    d_w = torch.empty_like(w)
    d_b = torch.empty(ow, dtype=torch.float32, device=x.device)
    ws = _ws(lib.mdno_fc_out_bwd_workspace_bytes(rows, width, ow), x.device)
    check(lib.mdno_fc_out_bwd(ptr(x), ptr(w), ptr(g), rows, width, ow, ptr(dx), ptr(d_w), ptr(d_b), ptr(ws), ws.numel(),
                              stream_ptr(x.device)), "mdno_fc_out_bwd")
    return dx, d_w, d_b


def node_prologue_bwd(pack: ParamPack, frames: torch.Tensor, x_aminoacid: torch.Tensor, x0: torch.Tensor,
                      g0: torch.Tensor):
    """Backward of `node_prologue`: -> dict of gradients under the state_dict key names."""
    lib = _lib.load()
    frames = f32(frames)
    if frames.dim() == 3:
        frames = frames.unsqueeze(1)
    W, M, N, _ = frames.shape
    dev = frames.device
    aa = x_aminoacid.to(device=dev, dtype=torch.long).contiguous()
    p = pack.struct
    has_lstm = "lstm_w_ih" in pack.tensors
    d_lstm = torch.empty(108, dtype=torch.float32, device=dev) if has_lstm else None
    d_emb = torch.empty((p.num_embeddings, p.embedding_dim), dtype=torch.float32, device=dev)
    d_w = torch.empty((p.width, p.in_width), dtype=torch.float32, device=dev)
    d_b = torch.empty(p.width, dtype=torch.float32, device=dev)
    ws = _ws(lib.mdno_node_prologue_bwd_workspace_bytes(pack.ref, M * N), dev)
    check(lib.mdno_node_prologue_bwd(pack.ref, ptr(frames), M, W, N, ptr(aa), int(aa.numel() == M * N and M > 1),
                                     ptr(f32(x0)), ptr(f32(g0)), ptr(d_lstm), ptr(d_emb), ptr(d_w), ptr(d_b), ptr(ws),
                                     ws.numel(), stream_ptr(dev)), "mdno_node_prologue_bwd")
    out = {"emb.weight": d_emb, "fc1.weight": d_w, "fc1.bias": d_b}
    if has_lstm:
        # six DISJOINT slices of one buffer (the kernel writes b_hh's gradient — the same values as b_ih's — a second
        # time at [96:108]): autograd's AccumulateGrad keeps the tensor it is handed, so no two .grad tensors may
        # share memory (clip_grad_norm_, accumulation without zero_grad), and none of them needs a copy
        out.update({"lstm.weight_ih_l0": d_lstm[0:36].view(12, 3), "lstm.weight_hh_l0": d_lstm[36:72].view(12, 3),
                    "lstm.bias_ih_l0": d_lstm[72:84], "lstm.bias_hh_l0": d_lstm[96:108],
                    "lstm_fc.weight": d_lstm[84:93].view(3, 3), "lstm_fc.bias": d_lstm[93:96]})
    return out


# ------------------------------------------------------------------------------------------------
def collate_samples(pos: torch.Tensor, rows: torch.Tensor, cols: torch.Tensor, meta: torch.Tensor, B: int, N: int,
                    W: int, horizon: int, n_edges: int, max_edges_per_sample: int):
    """Block-diagonal training batch built on the device (include/mdno.h mdno_collate_samples).
    -> (x_position [W,B*N,3], y [B*N,3], edge_index i64 [2,E], edge_attr [E,6])"""
    lib = _lib.load()
    dev = pos.device
    x_position = torch.empty((W, B * N, 3), dtype=torch.float32, device=dev)
    y = torch.empty((B * N, 3), dtype=torch.float32, device=dev)
    edge_index = torch.empty((2, n_edges), dtype=torch.long, device=dev)
    edge_attr = torch.empty((n_edges, 6), dtype=torch.float32, device=dev)
    check(lib.mdno_collate_samples(ptr(pos), ptr(rows), ptr(cols), ptr(meta), B, N, W, horizon, max_edges_per_sample,
                                   ptr(x_position), ptr(y), ptr(edge_index), ptr(edge_attr), stream_ptr(dev)),
          "mdno_collate_samples")
    return x_position, y, edge_index, edge_attr


# ------------------------------------------------------------------------------------------------
def lploss_rel_fwd(out: torch.Tensor, y: torch.Tensor, size_average: bool):
    """LpLoss.rel, p = 2, and the batch MSE in one pass (include/mdno.h mdno_lploss_rel_fwd): out, y f32 [B, D] ->
    (loss_mse f32 [2] = [loss, mse], stats f32 [B, 4] for the backward)."""
    lib = _lib.load()
    out, y = f32(out), f32(y)
    if out.dim() != 2 or out.shape != y.shape:
        raise MdnoError(f"lploss_rel: out {tuple(out.shape)} and y {tuple(y.shape)} must be the same [B, D]")
    B, D = out.shape
    stats = torch.empty((B, 4), dtype=torch.float32, device=out.device)
    res = torch.empty(2, dtype=torch.float32, device=out.device)
    check(lib.mdno_lploss_rel_fwd(ptr(out), ptr(y), B, D, int(bool(size_average)), ptr(stats), ptr(res),
                                  stream_ptr(out.device)), "mdno_lploss_rel_fwd")
    return res, stats


def lploss_rel_bwd(out: torch.Tensor, y: torch.Tensor, stats: torch.Tensor, grad_loss, size_average: bool) -> torch.Tensor:
    """d loss / d out (include/mdno.h mdno_lploss_rel_bwd); grad_loss: the f32 scalar tensor autograd hands down, or None."""
    lib = _lib.load()
    B, D = out.shape
    g = torch.empty_like(out)
    gl = None if grad_loss is None else f32(grad_loss).reshape(1)
    check(lib.mdno_lploss_rel_bwd(ptr(out), ptr(y), ptr(stats), ptr(gl), B, D, int(bool(size_average)), ptr(g),
                                  stream_ptr(out.device)), "mdno_lploss_rel_bwd")
    return g
