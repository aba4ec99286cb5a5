"""ctypes binding of libmdno.so (include/mdno.h).  No CPU fallback exists: if the library is
missing, or a tensor is not on the GPU, the call raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MDNO_LIB", _HERE / "libmdno.so"))

OK, EINVAL, ELAUNCH, EWORKSPACE, EUNSUPPORTED = 0, -1, -2, -3, -4
AGGR = {"add": 0, "mean": 1, "max": 2}      # "max": mdno_nnconv_fwd only (inference)
STATUS_EDGE_OVERFLOW, STATUS_BAD_AMINOACID = 1, 2
ABI_VERSION = 15
GEMM_MODES = {"split_bf16": 0, "f32": 1, "split_f16": 2}
CONV_MODES = {"materialized": 0, "factored": 1, "auto": 2}
STATUS_BAD_EDGE_INDEX = 16


class MdnoError(RuntimeError):
    pass


class MdnoIndexError(MdnoError, IndexError):
    """An index the reference's nn.Embedding / index_select / scatter would reject with IndexError."""


class KernelNNParams(C.Structure):
    """struct mdno_kernelnn_params"""
    _INTS = ["width", "ker_width", "depth", "ker_in", "in_width", "out_width",
             "num_embeddings", "embedding_dim", "x_position_dim", "gemm_mode", "conv_mode", "reserved0"]
    _PTRS = ["lstm_w_ih", "lstm_w_hh", "lstm_b_ih", "lstm_b_hh", "lstm_fc_w", "lstm_fc_b", "emb_w",
             "fc1_w", "fc1_b", "k_w0", "k_b0", "k_w1", "k_b1", "k_w2", "k_b2",
             "k2_w0", "k2_b0", "k2_w1", "k2_b1", "k2_w2", "k2_b2",
             "conv1_root", "conv1_bias", "conv2_root", "conv2_bias", "fc2_w", "fc2_b"]
    _fields_ = [(n, C.c_int32) for n in _INTS] + [(n, C.c_void_p) for n in _PTRS]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_SZ = C.c_size_t
_D = C.c_double

# name -> (restype, argtypes); kept in step with include/mdno.h (tests/test_cabi.py checks both ways)
SIGNATURES = {
    "mdno_abi_version": (_I, []),
    "mdno_last_error": (C.c_char_p, []),
    "mdno_build_id": (C.c_char_p, []),
    "mdno_radius_graph_csr": (_I, [_P, _I, _I, _D, _P, _P, _P, _L, _P, _P, _P]),
    "mdno_radius_graph_workspace_bytes": (_SZ, [_I, _I]),
    "mdno_radius_graph_csr_ws": (_I, [_P, _I, _I, _D, _P, _P, _P, _L, _P, _P, _P, _SZ, _P]),
    "mdno_coo_to_csr_workspace_bytes": (_SZ, [_L, _I]),
    "mdno_coo_to_csr": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "mdno_csr_by_source": (_I, [_P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "mdno_permute_rows": (_I, [_P, _P, _L, _I, _P, _P]),
    "mdno_edge_mlp_workspace_bytes": (_SZ, [_I, _I, _L, _I]),
    "mdno_edge_mlp_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _SZ,
                               _P]),
    "mdno_nnconv_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "mdno_node_prologue_fwd": (_I, [C.POINTER(KernelNNParams), _P, _I, _I, _I, _P, _I, _P, _P, _P]),
    "mdno_fc_out_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "mdno_kernelnn_workspace_bytes": (_SZ, [C.POINTER(KernelNNParams), _I, _I, _L]),
    "mdno_resolve_conv_mode": (_I, [C.POINTER(KernelNNParams), _I, _L]),
    "mdno_conv_mode_for_graph": (_I, [C.POINTER(KernelNNParams), _I, _I, _L]),
    "mdno_kernelnn_fwd": (_I, [C.POINTER(KernelNNParams), _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _L,
                               _P, _P, _P, _P, _P, _P, _SZ, _P, _P]),
    "mdno_rollout_workspace_bytes": (_SZ, [C.POINTER(KernelNNParams), _I, _I, _L]),
    "mdno_rollout": (_I, [C.POINTER(KernelNNParams), _P, _I, _I, _I, _I, _P, _I, _D, _L, _P, _SZ, _P, _P, _I, _P]),
    "mdno_rollout_plan_create": (_I, [C.POINTER(_P), C.POINTER(KernelNNParams), _P, _I, _I, _I, _I, _P, _I, _D, _L,
                                      _P, _SZ, _P, _P, _I, _P]),
    "mdno_rollout_plan_run": (_I, [_P, _I, _I, _P]),
    "mdno_rollout_plan_steps_per_launch": (_I, [_P]),
    "mdno_rollout_plan_destroy": (_I, [_P]),
    "mdno_rollout_plan_timer_attach": (_I, [_P, _I]),
    "mdno_rollout_plan_timer_read": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mdno_rollout_plan_timer_detach": (_I, [_P]),
    "mdno_rollout_plan_fallback_counts": (_I, [_P, _P, _P]),
    "mdno_kernelnn_fallback_counts": (_I, [C.POINTER(KernelNNParams), _I, _I, _L, _I, _P, _P, _P]),
    "mdno_adam_step": (_I, [_I, _P, _D, _D, _D, _D, _D, _L, _P]),
    "mdno_linear_fwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "mdno_linear_split_workspace_bytes": (_SZ, [_L, _I, _I]),
    "mdno_linear_split_fwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _P, _P, _SZ, _P]),
    "mdno_gemm_atb_split_f16_supported": (_I, [_L, _I, _I]),
    "mdno_gemm_atb_split_f16_workspace_bytes": (_SZ, [_L, _I, _I]),
    "mdno_gemm_atb_split_f16": (_I, [_P, _P, _L, _I, _I, _P, _I, _P, _SZ, _P]),
    "mdno_linear_split_f16_workspace_bytes": (_SZ, [_L, _I, _I]),
    "mdno_linear_split_f16_fwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _P, _P, _SZ, _P]),
    "mdno_reduce_workspace_bytes": (_SZ, [_I, _I]),
    "mdno_gemm_atb": (_I, [_P, _P, _L, _I, _I, _P, _I, _P, _SZ, _P]),
    "mdno_colsum": (_I, [_P, _L, _I, _P, _I, _P, _SZ, _P]),
    "mdno_relu_bwd": (_I, [_P, _P, _P, _L, _I, _P, _P]),
    "mdno_relu_bwd2": (_I, [_P, _P, _P, _L, _I, _P, _P, _P]),
    "mdno_transpose": (_I, [_P, _I, _I, _P, _P]),
    "mdno_inv_degree": (_I, [_P, _I, _I, _P, _P]),
    "mdno_nnconv_bwd_x": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P, _P]),
    "mdno_nnconv_bwd_root_workspace_bytes": (_SZ, [_L]),
    "mdno_nnconv_bwd_root": (_I, [_P, _P, _L, _I, _I, _P, _P, _I, _P, _SZ, _P]),
    "mdno_nnconv_bwd_root_pair_workspace_bytes": (_SZ, [_L]),
    "mdno_nnconv_bwd_root_pair": (_I, [_P, _P, _L, _P, _P, _P, _P, _P, _SZ, _P]),
    "mdno_nnconv_bwd_we": (_I, [_P, _P, _P, _P, _L, _I, _L, _I, _I, _P, _I, _P]),
    "mdno_cast_bf16": (_I, [_P, _L, _P, _P]),
    "mdno_linear_smallk_bf16_fwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "mdno_linear_bf16_workspace_bytes": (_SZ, [_I, _I]),
    "mdno_linear_bf16_fwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    "mdno_linear_bf16_masked_supported": (_I, [_L, _I, _I]),
    "mdno_linear_bf16_masked": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _SZ, _P]),
    "mdno_gemm_atb_bf16_workspace_bytes": (_SZ, [_I, _I]),
    "mdno_gemm_atb_bf16": (_I, [_P, _P, _L, _I, _I, _P, _P, _SZ, _P]),
    "mdno_nnconv_bf16w_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P]),
    "mdno_nnconv_bwd_x_bf16w": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mdno_nnconv_bwd_we_bf16": (_I, [_P, _P, _P, _P, _L, _I, _L, _P, _P]),
    "mdno_nnconv_bwd_we_bf16_colsum_workspace_bytes": (_SZ, []),
    "mdno_nnconv_bwd_we_colsum_workspace_bytes": (_SZ, []),
    "mdno_nnconv_bwd_we_colsum": (_I, [_P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _SZ, _P]),
    "mdno_nnconv_bwd_we_bf16_colsum": (_I, [_P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _SZ, _P]),
    "mdno_relu_bwd_bf16": (_I, [_P, _P, _L, _I, _I, _P, _P]),
    "mdno_colsum_bf16_workspace_bytes": (_SZ, [_I]),
    "mdno_colsum_bf16": (_I, [_P, _L, _I, _P, _P, _SZ, _P]),
    "mdno_nnconv_chain_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P]),
    "mdno_nnconv_chain_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mdno_nnconv_chain_bf16w_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P]),
    "mdno_nnconv_chain_bf16w_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mdno_colsum_atb_bf16_workspace_bytes": (_SZ, [_I, _I]),
    "mdno_colsum_atb_bf16": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _SZ, _P]),
    "mdno_node_prologue_bwd_workspace_bytes": (_SZ, [C.POINTER(KernelNNParams), _I]),
    "mdno_node_prologue_bwd": (_I, [C.POINTER(KernelNNParams), _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _SZ,
                                    _P]),
    "mdno_fc_out_bwd_workspace_bytes": (_SZ, [_I, _I, _I]),
    "mdno_fc_out_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "mdno_collate_samples": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "mdno_lploss_rel_fwd": (_I, [_P, _P, _L, _I, _I, _P, _P, _P]),
    "mdno_lploss_rel_bwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P]),
}

_lib = None


def source_build_id() -> str | None:
    """The id csrc/build.sh compiles into the library, recomputed from the tree: sha256 over csrc/*.{hip,h,sh} and
    include/mdno.h (bytes, C-locale name order), first 16 hex digits.  None where the sources are not beside the
    package (a binary-only install)."""
    import hashlib
    csrc = _HERE / "csrc"
    header = _HERE.parent / "include" / "mdno.h"
    if not csrc.is_dir() or not header.exists():
        return None
    files = sorted([f for f in csrc.iterdir() if f.suffix in (".hip", ".h", ".sh")], key=lambda f: str(f).encode())
    h = hashlib.sha256()
    for f in files + [header]:
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def load() -> C.CDLL:
    """Load libmdno.so once; raise (never fall back) if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise MdnoError(
            f"{LIB_PATH} not found: the HIP library is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or molecular_dynamics_neural_operator_amd/csrc/build.sh). There is no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    ver = lib.mdno_abi_version()
    if ver != ABI_VERSION:
        raise MdnoError(f"libmdno ABI {ver} != binding {ABI_VERSION}; rebuild the library")
    want, have = source_build_id(), lib.mdno_build_id().decode()
    # (an experimental build — scripts/micro/build_exp.sh, loaded through MDNO_LIB — says so in its build id; MDNO_LIB
    # pointing at any other library, the in-tree one included, does not switch the check off)
    if want is not None and have != want and have != "experimental":
        raise MdnoError(f"{LIB_PATH} was built from other sources (build id {have}, the tree is {want}): run "
                        f"molecular_dynamics_neural_operator_amd/csrc/build.sh — a stale library is never used")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != OK:
        msg = load().mdno_last_error().decode(errors="replace")
        raise MdnoError(f"{what or 'libmdno'} failed (code {rc}): {msg}")


def ptr(t) -> int | None:
    """Device pointer of a CUDA(HIP) tensor, None for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MdnoError("libmdno kernels run on the GPU only: got a CPU tensor (no CPU fallback exists)")
    if not t.is_contiguous():
        raise MdnoError("libmdno needs contiguous tensors")
    return t.data_ptr()


def f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise MdnoError("no HIP device visible: this package only runs on an MI355X-class GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def raise_on_status(status_word: int, what: str = "") -> None:
    """Turn the device status bits (include/mdno.h MDNO_STATUS_*) into the exception the reference
    would have raised at that point (IndexError from nn.Embedding / index_select) or an MdnoError."""
    st = int(status_word)
    if not st:
        return
    pre = (what + ": ") if what else ""
    if st & STATUS_BAD_AMINOACID:
        raise MdnoIndexError(pre + "x_aminoacid outside [0, num_embeddings) (index out of range in self)")
    if st & STATUS_BAD_EDGE_INDEX:
        raise MdnoIndexError(pre + "edge_index holds a node id outside [0, num_nodes)")
    if st & STATUS_EDGE_OVERFLOW:
        raise MdnoError(pre + "radius graph exceeded edge_cap; use a larger capacity")
    raise MdnoError(pre + f"device status {st:#x}")
