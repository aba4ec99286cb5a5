"""MI355X-native graph-kernel neural-operator rollout engine.

Drop-in for the hot path of ramanathanlab/molecular_dynamics_neural_operator
(`graph_kernel.py` kernel-integral message passing + autoregressive rollout): import
`graph_kernel` / `dataset` from this package instead of the reference's modules.
The arithmetic lives in libmdno.so (hand-written HIP for gfx950, C ABI in include/mdno.h).
"""
from . import dataset, graph_kernel, synthetic, weights  # noqa: F401
from ._lib import MdnoError, MdnoIndexError  # noqa: F401
from .checkpoint import load_reference_checkpoint  # noqa: F401
from .dataset import ContactMapDataset, PairData  # noqa: F401
from .graph_kernel import (DenseNet, KernelNN, KernelNNNotebook, LpLoss, NNConv_old, construct_pairdata,  # noqa: F401
                           propogate, recursive_propagation)

__all__ = ["dataset", "graph_kernel", "synthetic", "weights", "MdnoError", "MdnoIndexError", "load_reference_checkpoint", "ContactMapDataset", "PairData",
           "DenseNet", "KernelNN", "KernelNNNotebook", "LpLoss", "NNConv_old", "construct_pairdata", "propogate",
           "recursive_propagation"]
