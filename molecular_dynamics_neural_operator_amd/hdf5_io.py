"""HDF5 containers of the reference's data files (dataset.py:112-141) without h5py: a ctypes binding of the HDF5 C
library's own API — the library h5py itself wraps — for the handful of calls reading (and writing) whole datasets
takes.  `read_datasets` returns what `np.array(f[name][...])` returns under h5py: an ndarray in the file's own
dtype for a numeric dataset of any rank, layout and filter pipeline the library was built with (contiguous, chunked,
gzip ...), and a 1-D object array of 1-D arrays for a variable-length dataset (`contact_map`: one flat
`[rows..., cols...]` vector per frame).  `write_trajectory_h5` writes a trajectory in the reference's layout.

The shared library is looked for in `$MDNO_HDF5_LIB`, then by `ctypes.util.find_library("hdf5")`, then in the usual
distribution directories (this project's image ships one under /opt/conda/lib).  `ContactMapDataset` uses h5py when it
is importable and this module otherwise; with neither, reading an `.h5` raises and names the `.npz` twin.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import Dict, Optional, Sequence

import numpy as np

hid_t = C.c_int64          # HDF5 >= 1.10
hsize_t = C.c_ulonglong
_H5T_INTEGER, _H5T_FLOAT, _H5T_VLEN = 0, 1, 9
_H5T_DIR_ASCEND = 1
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5P_DEFAULT = _H5S_ALL = 0


class _hvl_t(C.Structure):
    _fields_ = [("len", C.c_size_t), ("p", C.c_void_p)]


class Hdf5Error(RuntimeError):
    pass


_lib = None


def _candidates():
    env = os.environ.get("MDNO_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for pat in ("/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/lib/x86_64-linux-gnu/libhdf5.so*", "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*",
                "/opt/conda/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def library_path() -> Optional[str]:
    """Path (or soname) of an HDF5 C library that loads, None when there is none."""
    for cand in _candidates():
        try:
            C.CDLL(cand)
            return cand
        except OSError:
            continue
    return None


def available() -> bool:
    """An HDF5 C library is present AND binds (every symbol this module uses; version >= 1.10)."""
    try:
        _load()
        return True
    except Hdf5Error:
        return False


def _load():
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if path is None:
        raise Hdf5Error("no HDF5 C library found (set MDNO_HDF5_LIB to libhdf5.so, or install h5py)")
    lib = C.CDLL(path)

    def sig(name, res, *args):
        try:
            f = getattr(lib, name)
        except AttributeError:
            raise Hdf5Error(f"{path}: no symbol {name} (an HDF5 build this binding does not cover)") from None
        f.restype, f.argtypes = res, list(args)
        return f

    sig("H5open", C.c_int)
    sig("H5get_libversion", C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint))
    sig("H5Eset_auto2", C.c_int, hid_t, C.c_void_p, C.c_void_p)
    sig("H5Fopen", hid_t, C.c_char_p, C.c_uint, hid_t)
    sig("H5Fcreate", hid_t, C.c_char_p, C.c_uint, hid_t, hid_t)
    sig("H5Fclose", C.c_int, hid_t)
    sig("H5Lexists", C.c_int, hid_t, C.c_char_p, hid_t)
    sig("H5Dopen2", hid_t, hid_t, C.c_char_p, hid_t)
    sig("H5Dcreate2", hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t)
    sig("H5Dclose", C.c_int, hid_t)
    sig("H5Dget_space", hid_t, hid_t)
    sig("H5Dget_type", hid_t, hid_t)
    sig("H5Dread", C.c_int, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    sig("H5Dwrite", C.c_int, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    # variable-length rows are freed by H5Treclaim (1.12+); H5Dvlen_reclaim is its deprecated predecessor, absent
    # from libraries built without deprecated symbols and from HDF5 2.x — same signature either way
    try:
        lib._mdno_reclaim = sig("H5Treclaim", C.c_int, hid_t, hid_t, hid_t, C.c_void_p)
    except Hdf5Error:
        lib._mdno_reclaim = sig("H5Dvlen_reclaim", C.c_int, hid_t, hid_t, hid_t, C.c_void_p)
    sig("H5Screate_simple", hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t))
    sig("H5Sget_simple_extent_ndims", C.c_int, hid_t)
    sig("H5Sget_simple_extent_dims", C.c_int, hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t))
    sig("H5Sclose", C.c_int, hid_t)
    sig("H5Tget_class", C.c_int, hid_t)
    sig("H5Tget_size", C.c_size_t, hid_t)
    sig("H5Tget_sign", C.c_int, hid_t)
    sig("H5Tget_super", hid_t, hid_t)
    sig("H5Tget_native_type", hid_t, hid_t, C.c_int)
    sig("H5Tvlen_create", hid_t, hid_t)
    sig("H5Tclose", C.c_int, hid_t)
    sig("H5Pcreate", hid_t, hid_t)
    sig("H5Pset_chunk", C.c_int, hid_t, C.c_int, C.POINTER(hsize_t))
    sig("H5Pset_deflate", C.c_int, hid_t, C.c_uint)
    sig("H5Pclose", C.c_int, hid_t)
    if lib.H5open() < 0:
        raise Hdf5Error(f"{path}: H5open failed")
    maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
    lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        raise Hdf5Error(f"{path}: HDF5 {maj.value}.{mnr.value}.{rel.value} (hid_t is 64-bit from 1.10 on; older libraries are not bound)")
    lib.H5Eset_auto2(0, None, None)        # errors are reported through return codes -> exceptions, not on stderr
    _lib = lib
    return lib


def _native(name: str) -> int:
    """Value of one of the library's global type / property-class ids (valid after H5open)."""
    return hid_t.in_dll(_load(), name).value


def _np_dtype(lib, t) -> np.dtype:
    cls, size = lib.H5Tget_class(t), lib.H5Tget_size(t)
    if cls == _H5T_INTEGER:
        return np.dtype(("i" if lib.H5Tget_sign(t) == 1 else "u") + str(size))
    if cls == _H5T_FLOAT and size in (2, 4, 8):
        return np.dtype("f" + str(size))
    raise Hdf5Error(f"unsupported HDF5 datatype (class {cls}, {size} bytes)")


class _Closer:
    """ids opened inside one call, closed in reverse order on the way out"""

    def __init__(self, lib):
        self.lib, self.ids = lib, []

    def add(self, i, close, what):
        if i < 0:
            raise Hdf5Error(what)
        self.ids.append((i, close))
        return i

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for i, close in reversed(self.ids):
            close(i)
        return False


def read_datasets(path: str, names: Sequence[str]) -> Dict[str, np.ndarray]:
    """{name: array} for those of `names` that exist at the root of the file (whole datasets, like `f[name][...]`)."""
    lib = _load()
    out = {}
    with _Closer(lib) as cl:
        f = cl.add(lib.H5Fopen(os.fsencode(str(path)), _H5F_ACC_RDONLY, _H5P_DEFAULT), lib.H5Fclose,
                   f"cannot open {path} as an HDF5 file")
        for n in names:
            if lib.H5Lexists(f, n.encode(), _H5P_DEFAULT) <= 0:
                continue
            out[n] = _read_one(lib, f, n, str(path))
    return out


def _read_one(lib, f, name, path):
    with _Closer(lib) as cl:
        d = cl.add(lib.H5Dopen2(f, name.encode(), _H5P_DEFAULT), lib.H5Dclose, f"{path}: cannot open dataset {name!r}")
        sp = cl.add(lib.H5Dget_space(d), lib.H5Sclose, f"{path}:{name}: dataspace")
        nd = lib.H5Sget_simple_extent_ndims(sp)
        if nd < 0:
            raise Hdf5Error(f"{path}:{name}: not a simple dataspace")
        dims = (hsize_t * max(nd, 1))()
        if nd:
            lib.H5Sget_simple_extent_dims(sp, dims, None)
        shape = tuple(int(dims[i]) for i in range(nd))
        ft = cl.add(lib.H5Dget_type(d), lib.H5Tclose, f"{path}:{name}: datatype")
        mt = cl.add(lib.H5Tget_native_type(ft, _H5T_DIR_ASCEND), lib.H5Tclose, f"{path}:{name}: no native equivalent of the datatype")
        if lib.H5Tget_class(mt) == _H5T_VLEN:
            base = cl.add(lib.H5Tget_super(mt), lib.H5Tclose, f"{path}:{name}: vlen base type")
            dt = _np_dtype(lib, base)
            count = int(np.prod(shape)) if shape else 1
            buf = (_hvl_t * max(count, 1))()
            if count and lib.H5Dread(d, mt, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, buf) < 0:
                raise Hdf5Error(f"{path}:{name}: H5Dread failed")
            rows = np.empty(count, dtype=object)
            for i in range(count):
                ln = int(buf[i].len)
                rows[i] = (np.frombuffer((C.c_char * (ln * dt.itemsize)).from_address(buf[i].p), dtype=dt).copy()
                           if ln else np.zeros(0, dt))
            if count:
                lib._mdno_reclaim(mt, sp, _H5P_DEFAULT, buf)      # the library allocated the rows
            return rows.reshape(shape) if shape else rows[0]
        dt = _np_dtype(lib, mt)
        arr = np.empty(shape, dtype=dt)
        if arr.size and lib.H5Dread(d, mt, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)) < 0:
            raise Hdf5Error(f"{path}:{name}: H5Dread failed")
        return arr


_NATIVE_OF = {"i1": "H5T_NATIVE_INT8_g", "i2": "H5T_NATIVE_INT16_g", "i4": "H5T_NATIVE_INT32_g", "i8": "H5T_NATIVE_INT64_g",
              "u1": "H5T_NATIVE_UINT8_g", "u2": "H5T_NATIVE_UINT16_g", "u4": "H5T_NATIVE_UINT32_g", "u8": "H5T_NATIVE_UINT64_g",
              "f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g"}


def _type_of(dt: np.dtype) -> int:
    key = dt.kind + str(dt.itemsize)
    if key not in _NATIVE_OF or not dt.isnative:
        raise Hdf5Error(f"dtype {dt} is not written")
    return _native(_NATIVE_OF[key])


def write_datasets(path: str, arrays: Dict[str, np.ndarray], gzip: Optional[int] = None) -> None:
    """Create `path` with one root dataset per entry: a numeric ndarray as it is (chunked by its first axis and
    deflated at level `gzip` when given), a 1-D object array / list of 1-D integer arrays as a variable-length
    dataset of its rows' dtype."""
    lib = _load()
    with _Closer(lib) as cl:
        f = cl.add(lib.H5Fcreate(os.fsencode(str(path)), _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT), lib.H5Fclose,
                   f"cannot create {path}")
        for name, a in arrays.items():
            ragged = isinstance(a, (list, tuple)) or (isinstance(a, np.ndarray) and a.dtype == object)
            with _Closer(lib) as c2:
                if ragged:
                    rows = [np.ascontiguousarray(r) for r in a]
                    dts = {r.dtype for r in rows} or {np.dtype("int64")}
                    if len(dts) != 1:
                        raise Hdf5Error(f"{name}: rows of different dtypes {sorted(map(str, dts))}")
                    vt = c2.add(lib.H5Tvlen_create(_type_of(dts.pop())), lib.H5Tclose, f"{name}: vlen type")
                    dims = (hsize_t * 1)(len(rows))
                    sp = c2.add(lib.H5Screate_simple(1, dims, None), lib.H5Sclose, f"{name}: dataspace")
                    d = c2.add(lib.H5Dcreate2(f, name.encode(), vt, sp, _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT), lib.H5Dclose,
                               f"{path}: cannot create dataset {name!r}")
                    buf = (_hvl_t * max(len(rows), 1))()
                    for i, r in enumerate(rows):
                        buf[i].len, buf[i].p = r.size, r.ctypes.data
                    if rows and lib.H5Dwrite(d, vt, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, buf) < 0:
                        raise Hdf5Error(f"{path}:{name}: H5Dwrite failed")
                    continue
                a = np.ascontiguousarray(a)
                t = _type_of(a.dtype)
                dims = (hsize_t * max(a.ndim, 1))(*a.shape)
                sp = c2.add(lib.H5Screate_simple(a.ndim, dims, None), lib.H5Sclose, f"{name}: dataspace")
                dcpl = _H5P_DEFAULT
                if gzip is not None and a.ndim >= 1 and a.size:
                    dcpl = c2.add(lib.H5Pcreate(_native("H5P_CLS_DATASET_CREATE_ID_g")), lib.H5Pclose, "dataset creation properties")
                    chunk = (hsize_t * a.ndim)(*((min(a.shape[0], 64),) + a.shape[1:]))
                    if lib.H5Pset_chunk(dcpl, a.ndim, chunk) < 0 or lib.H5Pset_deflate(dcpl, int(gzip)) < 0:
                        raise Hdf5Error(f"{name}: chunk / deflate (is the library built with zlib?)")
                d = c2.add(lib.H5Dcreate2(f, name.encode(), t, sp, _H5P_DEFAULT, dcpl, _H5P_DEFAULT), lib.H5Dclose,
                           f"{path}: cannot create dataset {name!r}")
                if a.size and lib.H5Dwrite(d, t, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) < 0:
                    raise Hdf5Error(f"{path}:{name}: H5Dwrite failed")


def write_trajectory_h5(path, frames: np.ndarray, contact_maps: Sequence[np.ndarray], amino_acids: np.ndarray,
                        rmsd: Optional[np.ndarray] = None, gzip: Optional[int] = None) -> None:
    """Frames `[T,N,3]` + per-frame flat COO as the reference's files hold them (dataset.py:112-127, :159):
    `contact_map` variable-length int64 `[T]`, `point_cloud` f32 `[T,3,N]`, `rmsd` f32 `[T]`, `amino_acids` int64 `[N]`."""
    frames = np.asarray(frames, dtype=np.float32)
    write_datasets(path, {
        "contact_map": [np.asarray(c, dtype=np.int64).reshape(-1) for c in contact_maps],
        "point_cloud": np.ascontiguousarray(np.transpose(frames, (0, 2, 1))),
        "rmsd": np.zeros(len(frames), np.float32) if rmsd is None else np.asarray(rmsd, np.float32),
        "amino_acids": np.asarray(amino_acids, dtype=np.int64),
    }, gzip=gzip)


def h5_to_npz(path_in, path_out, names: Sequence[str] = ("contact_map", "point_cloud", "rmsd", "amino_acids")) -> None:
    """The `.npz` twin of an HDF5 trajectory file (same dataset names; ragged datasets flat + `<name>_offsets`, no pickle):
    what `ContactMapDataset` reads on a machine with neither h5py nor libhdf5."""
    d = read_datasets(path_in, names)
    out = {}
    for n, a in d.items():
        if a.dtype == object:
            rows = [np.asarray(r).reshape(-1) for r in a]
            off = np.zeros(len(rows) + 1, np.int64)
            np.cumsum([r.size for r in rows], out=off[1:])
            out[n] = np.concatenate(rows) if rows else np.zeros(0, np.int64)
            out[n + "_offsets"] = off
        else:
            out[n] = a
    np.savez(path_out, **out)
