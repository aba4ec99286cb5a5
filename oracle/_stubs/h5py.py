"""`h5py` stand-in backed by `.npz` files: `File(path)` exposes `f[name][...]` / `f[name][:n]`.

Lets the reference's `ContactMapDataset` load the npz twin of an HDF5 trajectory so that the
restated loader can be checked against it (ragged `contact_map` rows are stored as an object array).
"""
import numpy as np


class File:
    def __init__(self, path, mode="r", **kwargs):
        assert mode == "r"
        self._z = np.load(str(path), allow_pickle=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._z.close()
        return False

    def __getitem__(self, name):
        if name not in self._z.files:
            raise ValueError(name)
        return self._z[name]
