"""Writes tests/golden/traj_h5py.h5 — a small trajectory in the layout of the reference's data files (dataset.py:112-127,
:159) — with REAL h5py, and its twin tests/golden/traj_h5py_twin.npz holding the same arrays, so that the build's
h5py-free reader (molecular_dynamics_neural_operator_amd/hdf5_io.py, a ctypes binding of libhdf5) is checked against a
file it did not write.  Test infrastructure; run once, by hand, with an interpreter that has h5py — in the build image

    /opt/conda/bin/python3.9 oracle/gen_h5_fixture.py

(the image's main interpreter has no h5py; this one has h5py 3.3.0 / HDF5 1.10.6 and no torch).  The datasets use what
such files use in the wild: a variable-length int16 `contact_map` (one flat [rows..., cols...] vector per frame), a
chunked, gzip-compressed, shuffled float32 `point_cloud` [T,3,N], a float64 `rmsd`, an int32 `amino_acids`, plus an
empty contact map (a frame without contacts) and a dataset the reader is not asked for.
"""
from pathlib import Path

import h5py
import numpy as np

out = Path(__file__).resolve().parents[1] / "tests" / "golden"
T, N = 14, 11
rng = np.random.default_rng(5)
step = rng.normal(size=(N, 3))
step /= np.linalg.norm(step, axis=1, keepdims=True)
base = np.cumsum(step * 3.8, axis=0)
frames = (base[None] + rng.normal(scale=0.4, size=(T, N, 3))).astype(np.float32)
cms = []
for t in range(T):
    d = np.sqrt(((frames[t].astype(np.float64)[:, None] - frames[t].astype(np.float64)[None]) ** 2).sum(-1))
    r, c = np.nonzero(d < 8.0)
    cms.append(np.concatenate([r, c]).astype(np.int16))
cms[6] = np.zeros(0, np.int16)                      # a frame with no contact at all
aa = rng.integers(0, 20, size=N).astype(np.int32)
rmsd = rng.random(T)
pc = np.ascontiguousarray(np.transpose(frames, (0, 2, 1)))

with h5py.File(out / "traj_h5py.h5", "w", libver="latest") as f:
    ds = f.create_dataset("contact_map", (T,), dtype=h5py.vlen_dtype(np.dtype("int16")), chunks=(4,))
    for t in range(T):
        ds[t] = cms[t]
    f.create_dataset("point_cloud", data=pc, chunks=(5, 3, N), compression="gzip", compression_opts=6, shuffle=True)
    f.create_dataset("rmsd", data=rmsd)
    f.create_dataset("amino_acids", data=aa)
    f.create_dataset("fnc", data=rng.random(T).astype(np.float32))

off = np.zeros(T + 1, np.int64)
np.cumsum([c.size for c in cms], out=off[1:])
np.savez(out / "traj_h5py_twin.npz", contact_map=np.concatenate(cms), contact_map_offsets=off, point_cloud=pc, rmsd=rmsd,
         amino_acids=aa)
print("wrote", out / "traj_h5py.h5", (out / "traj_h5py.h5").stat().st_size, "bytes; h5py", h5py.__version__, "hdf5",
      h5py.version.hdf5_version)
