"""Frame / graph-sample layout of the reference's dataset.py, without torch_geometric or h5py.

`PairData`            mirrors dataset.py:21-53  (field names, num_nodes, __inc__ batching rule)
`ContactMapDataset`   mirrors dataset.py:56-227 (constructor signature, __len__, __getitem__ layout,
                      ValueError when window+horizon exceed the data)

On-disk layout (dataset.py:112-127, 159): datasets `contact_map` (per-frame ragged flat COO
`[rows..., cols...]`), `point_cloud` `[T,3,N]`, `rmsd` `[T]`, `amino_acids` `[N]`.  HDF5 files are
read through `h5py` when it is importable and otherwise through the HDF5 C library itself (`hdf5_io.py`, ctypes;
tests/test_host_logic.py reads a file written by real h5py that way); an `.npz` with the same dataset names is read
everywhere.  In the `.npz` twin a ragged dataset is stored without pickling, as the concatenation of
its rows under its own name plus `<name>_offsets` (int64 [T+1]); object arrays written by older
versions are only read with `allow_pickle=True`.  Directory mode concatenates the sorted files
(dataset.py:134-141).
"""
from __future__ import annotations

import glob
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch


class PairData:
    """Graph sample container.  Fields (dataset.py:22-35):
    x_aminoacid i64 [N] · x_position f32 [W,N,3] · y f32 [N,3] · edge_attr f32 [E,6] · edge_index i64 [2,E]
    """

    _FIELDS = ("x_aminoacid", "x_position", "y", "edge_attr", "edge_index")

    def __init__(self, x_aminoacid=None, x_position=None, y=None, edge_attr=None, edge_index=None) -> None:
        self.x_aminoacid = x_aminoacid
        self.x_position = x_position
        self.y = y
        self.edge_attr = edge_attr
        self.edge_index = edge_index

    @property
    def num_nodes(self) -> int:
        return self.x_aminoacid.size(0)

    def __inc__(self, key, value=None, *args, **kwargs):
        """Offset added to `key` when samples are concatenated into a batch (dataset.py:41-45)."""
        return self.num_nodes if key == "edge_index" else 0

    def to(self, device, *args, **kwargs) -> "PairData":
        for k in self._FIELDS:
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, v.to(device, *args, **kwargs))
        return self

    def cpu(self) -> "PairData":
        return self.to("cpu")

    def pin_memory(self) -> "PairData":
        for k in self._FIELDS:
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, v.pin_memory())
        return self

    def __repr__(self) -> str:
        body = ", ".join(f"{k}={list(getattr(self, k).shape)}" for k in self._FIELDS
                         if torch.is_tensor(getattr(self, k)))
        return f"PairData({body})"

    @staticmethod
    def collate(samples: Sequence["PairData"]) -> "PairData":
        """Block-diagonal batch as torch_geometric's Batch.from_data_list would build it:
        node-level fields concatenated on dim 0, edge_index offset by the running node count
        (`__inc__`), x_position stacked window-major per sample ([B*W, N, 3])."""
        out = PairData()
        off = 0
        eis = []
        for s in samples:
            eis.append(s.edge_index + off)
            off += s.__inc__("edge_index")
        out.edge_index = torch.cat(eis, dim=1)
        out.edge_attr = torch.cat([s.edge_attr for s in samples], dim=0)
        out.x_aminoacid = torch.cat([s.x_aminoacid for s in samples], dim=0)
        out.x_position = torch.cat([s.x_position for s in samples], dim=0)
        if all(s.y is not None for s in samples):
            out.y = torch.cat([s.y for s in samples], dim=0)
        return out


def _read_container(path: str, names: Sequence[str], allow_pickle: bool = False):
    """Return {name: array} for the datasets of `names` present in an .h5 (needs h5py) or .npz file."""
    p = str(path)
    out = {}
    if p.endswith(".npz"):
        with np.load(p, allow_pickle=allow_pickle) as z:
            for n in names:
                if n not in z.files:
                    continue
                try:
                    arr = z[n]
                except ValueError as e:     # a pickled object array and allow_pickle is off
                    raise ValueError(
                        f"{p}: dataset {n!r} is a pickled object array; rewrite the file with "
                        "write_trajectory_npz (flat rows + offsets) or pass allow_pickle=True for a file "
                        "you trust") from e
                if n + "_offsets" in z.files:       # ragged rows stored flat
                    off = np.asarray(z[n + "_offsets"], dtype=np.int64)
                    rows = np.empty(len(off) - 1, dtype=object)
                    for t in range(len(off) - 1):
                        rows[t] = arr[off[t]:off[t + 1]]
                    arr = rows
                out[n] = arr
        return out
    try:
        import h5py  # noqa: used when present
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(p, "r", libver="latest", swmr=False) as f:
            for n in names:
                if n in f:
                    out[n] = np.array(f[n][...])
        return out
    # no h5py: the HDF5 C library itself through ctypes (hdf5_io.py) — the same arrays
    from . import hdf5_io
    if not hdf5_io.available():
        raise ImportError(f"reading {p} needs h5py or an HDF5 C library (libhdf5.so; MDNO_HDF5_LIB names one); "
                          "or convert the file to .npz with the same dataset names (hdf5_io.h5_to_npz where there is one)")
    return hdf5_io.read_datasets(p, names)


class ContactMapDataset(torch.utils.data.Dataset):
    """Windowed graph samples over a trajectory; whole file(s) loaded into RAM like the reference."""

    def __init__(
        self,
        path: str,
        edge_index_dset_name: str = "contact_map",
        edge_attr_dset_name: str = "point_cloud",
        node_feature_dset_name: Optional[str] = "amino_acids",
        node_feature: str = "amino_acid_onehot",
        constant_num_node_features: int = 20,
        window_size: int = 1,
        horizon: int = 1,
        node_feature_dset_path: Optional[str] = None,
        allow_pickle: bool = False,
    ):
        self._constant_num_node_features = constant_num_node_features
        self.window_size = window_size
        self.horizon = horizon
        ntrain = 100000000  # dataset.py:108
        names = (edge_index_dset_name, edge_attr_dset_name, "rmsd", node_feature_dset_name or "amino_acids")

        p = str(path)
        if p.endswith(".h5") or p.endswith(".npz"):
            files = [p]
        else:
            files = sorted(glob.glob(p + "/*.h5") + glob.glob(p + "/*.npz"))
            if not files:
                raise ValueError(f"no .h5/.npz trajectory files under {p}")
        edge_indices: List[np.ndarray] = []
        edge_attrs: List[np.ndarray] = []
        rmsd: List[np.ndarray] = []
        node_features = None
        for fpath in files:
            d = _read_container(fpath, names, allow_pickle)
            edge_indices.extend(list(d[edge_index_dset_name][:ntrain]))
            edge_attrs.append(np.asarray(d[edge_attr_dset_name][:ntrain]))
            if "rmsd" in d:
                rmsd.append(np.asarray(d["rmsd"][:ntrain]))
            if node_feature_dset_name is not None and names[-1] in d and node_features is None:
                node_features = np.asarray(d[names[-1]])
        if node_feature_dset_name is not None and node_feature_dset_path is not None:
            node_features = np.asarray(_read_container(node_feature_dset_path, (names[-1],), allow_pickle)[names[-1]])
        if node_feature_dset_name is not None and node_features is None:
            raise ValueError(f"dataset {names[-1]!r} not found (pass node_feature_dset_path)")

        self.edge_indices = edge_indices
        self.rmsd_values = np.concatenate(rmsd) if rmsd else []
        pos = np.concatenate(edge_attrs, axis=0)
        if len(self.edge_indices) - self.window_size - self.horizon + 1 < 0:
            raise ValueError("The sum of window_size and horizon is longer than the input data")
        # positions in order (T, num_nodes, 3)   (dataset.py:159)
        self.edge_attrs = np.ascontiguousarray(np.transpose(pos, [0, 2, 1]))
        self._node_features_dset = node_features
        self.x_aminoacid = torch.from_numpy(np.asarray(node_features)).to(torch.long)

    def __len__(self) -> int:
        return len(self.edge_indices) - self.window_size - self.horizon + 1

    def __getitem__(self, idx) -> PairData:
        pred_idx = idx + self.window_size + self.horizon - 1
        x_position = self.edge_attrs[idx:idx + self.window_size]
        # graph and edge attributes of the FIRST window frame (dataset.py:189-201)
        edge_index = np.asarray(self.edge_indices[idx]).reshape(2, -1)
        frame = self.edge_attrs[idx]
        edge_attr = np.concatenate([frame[edge_index[0]], frame[edge_index[1]]], axis=1).reshape(-1, 6)
        y = self.edge_attrs[pred_idx]
        return PairData(
            x_aminoacid=self.x_aminoacid,
            x_position=torch.from_numpy(np.ascontiguousarray(x_position)).to(torch.float32),
            y=torch.from_numpy(np.ascontiguousarray(y)).to(torch.float32),
            edge_attr=torch.from_numpy(edge_attr).to(torch.float32),
            edge_index=torch.from_numpy(np.ascontiguousarray(edge_index)).to(torch.long),
        )


def write_trajectory_npz(path, frames: np.ndarray, contact_maps: Sequence[np.ndarray], amino_acids: np.ndarray,
                         rmsd: Optional[np.ndarray] = None) -> None:
    """Write frames `[T,N,3]` + per-frame flat COO in the on-disk layout above (npz container; the
    ragged contact maps as one flat vector + offsets, so the file loads without pickle)."""
    frames = np.asarray(frames, dtype=np.float32)
    rows = [np.asarray(c, dtype=np.int64).reshape(-1) for c in contact_maps]
    offsets = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum([r.size for r in rows], out=offsets[1:])
    flat = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    np.savez(path, contact_map=flat, contact_map_offsets=offsets, point_cloud=np.ascontiguousarray(np.transpose(frames, (0, 2, 1))),
             rmsd=(np.zeros(len(frames), np.float32) if rmsd is None else np.asarray(rmsd, np.float32)),
             amino_acids=np.asarray(amino_acids, dtype=np.int64))
