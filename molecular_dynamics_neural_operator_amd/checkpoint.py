"""Loading checkpoints written by the reference.

The reference saves ``{"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict"}``
to ``run_path/"best.pt"`` on the best validation loss (graph_kernel.py:630-639).  The model is
wrapped in ``torch_geometric.nn.DataParallel`` before ``state_dict()`` is taken (graph_kernel.py:528,
:635), so every key carries a ``module.`` prefix.  The notebook loads such a file with
``torch.load`` + ``load_state_dict`` (bba_analysis.ipynb raw lines 80-111); its checkpoint predates
the LSTM / conv2 block and holds only ``emb, fc1, conv1.*, fc2`` (raw lines 123-128).

Nothing here touches the GPU: the tensors land in the ``nn.Module`` and reach the device with
``model.to("cuda")`` like any other state_dict.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Any, Dict, Mapping, Optional, Tuple, Union

import torch

from ._lib import MdnoError

_NOTEBOOK_GROUPS = ("emb.", "fc1.", "conv1.", "fc2.")
_INTREE_ONLY = ("lstm.", "lstm_fc.", "conv2.")


def _strip_module(sd: Mapping[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    out = OrderedDict()
    for k, v in sd.items():
        while k.startswith("module."):       # DataParallel (graph_kernel.py:528); tolerate double wrapping
            k = k[len("module."):]
        out[k] = v
    return out


def read_checkpoint(source: Union[str, os.PathLike, Mapping[str, Any]]) -> Tuple["OrderedDict[str, torch.Tensor]", Dict[str, Any]]:
    """-> (state_dict without ``module.`` prefixes, the other entries of the checkpoint).

    `source`: a path to a ``torch.save``d file, the ``best.pt`` dict itself, or a bare state_dict."""
    if isinstance(source, (str, os.PathLike)):
        obj = torch.load(os.fspath(source), map_location="cpu", weights_only=True)
    else:
        obj = source
    if not isinstance(obj, Mapping):
        raise MdnoError(f"checkpoint is a {type(obj).__name__}, expected a dict")
    if "model_state_dict" in obj:
        sd = obj["model_state_dict"]
        meta = {k: v for k, v in obj.items() if k != "model_state_dict"}
    else:
        sd, meta = obj, {}
    if not isinstance(sd, Mapping) or not sd or not all(torch.is_tensor(v) for v in sd.values()):
        raise MdnoError("checkpoint holds no tensor state_dict (expected 'model_state_dict' or a bare state_dict)")
    return _strip_module(sd), meta


def checkpoint_variant(sd: Mapping[str, torch.Tensor]) -> str:
    """"intree" (LSTM + conv1 + conv2, graph_kernel.py:264-275) or "notebook" (emb, fc1, conv1, fc2)."""
    has_intree = any(k.startswith(_INTREE_ONLY) for k in sd)
    return "intree" if has_intree else "notebook"


def infer_constructor_args(sd: Mapping[str, torch.Tensor]) -> Dict[str, int]:
    """Constructor arguments the shapes pin down (everything but `depth`, which no tensor records)."""
    try:
        width, in_width = sd["fc1.weight"].shape
        ker_width, ker_in = sd["conv1.net.layers.0.weight"].shape
        out_width = sd["fc2.weight"].shape[0]
        num_embeddings, embedding_dim = sd["emb.weight"].shape
    except KeyError as e:
        raise MdnoError(f"checkpoint lacks {e.args[0]!r}: not a KernelNN state_dict") from e
    return dict(width=int(width), ker_width=int(ker_width), ker_in=int(ker_in), in_width=int(in_width),
                out_width=int(out_width), num_embeddings=int(num_embeddings), embedding_dim=int(embedding_dim))


def load_reference_checkpoint(source, model: Optional[torch.nn.Module] = None, depth: Optional[int] = None,
                              strict: bool = True):
    """Load a reference checkpoint into `model` (a ``KernelNN`` / ``KernelNNNotebook``), or build the
    matching model when `model` is None (`depth` is then required: graph_kernel.py:327 default 6).

    Accepts ``best.pt`` dicts and bare state_dicts, with or without the ``module.`` prefix, for the
    in-tree key set and the notebook-era one.  Returns ``(model, meta)`` where `meta` holds the
    checkpoint's other entries (``epoch``, optimizer / scheduler state) plus ``variant``."""
    from .graph_kernel import KernelNN, KernelNNNotebook
    sd, meta = read_checkpoint(source)
    variant = checkpoint_variant(sd)
    meta["variant"] = variant
    if model is None:
        if depth is None:
            raise MdnoError("depth is not recorded in a checkpoint: pass depth= (reference default 6)")
        a = infer_constructor_args(sd)
        cls = KernelNN if variant == "intree" else KernelNNNotebook
        model = cls(a["width"], a["ker_width"], depth, a["ker_in"], a["in_width"], a["out_width"],
                    a["num_embeddings"], a["embedding_dim"])
    target = model.module if hasattr(model, "module") else model
    is_notebook_model = not hasattr(target, "lstm")
    if variant == "notebook" and not is_notebook_model:
        raise MdnoError("notebook-era checkpoint (emb, fc1, conv1, fc2 only; bba_analysis.ipynb:123-128) "
                        "does not fit the in-tree KernelNN: load it into KernelNNNotebook")
    if variant == "intree" and is_notebook_model:
        raise MdnoError("in-tree checkpoint (lstm, conv1, conv2) does not fit KernelNNNotebook: load it into KernelNN")
    result = target.load_state_dict(sd, strict=strict)
    meta["load_result"] = result
    return model, meta
