"""Read whole datasets out of an HDF5 file without h5py.

The reference's data files are HDF5 (`{dataset}.h5` with `feat_ids` / `labels`, written by
data_preprocess/proc_avazu.py:284-288 and read with h5py at code/dataset.py:27-29).  h5py is not
part of the ROCm image this package targets, but the HDF5 C library is; this module binds the
handful of libhdf5 calls a whole-dataset read needs through ctypes (so every layout and filter
the library supports — contiguous, chunked, gzip — reads correctly).

    read_datasets(path, ["feat_ids", "labels"]) -> {"feat_ids": ndarray, "labels": ndarray}

Order of preference in mapx.dataset: `<name>.npz`, h5py if importable, then this module.
"""
import ctypes
import ctypes.util
import glob
import os

import numpy as np

_SEARCH = ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*",
           "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*", "/usr/lib64/libhdf5.so*",
           "/usr/local/lib/libhdf5.so*")
_lib = None

H5F_ACC_RDONLY, H5P_DEFAULT, H5S_ALL = 0, 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_ORDER_LE, H5T_ORDER_BE = 0, 1
H5T_SGN_NONE = 0


class H5Error(OSError):
    pass


def _candidates():
    env = os.environ.get("MAPX_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in _SEARCH:
        for p in sorted(glob.glob(pat)):
            yield p


def library():
    """The loaded libhdf5 with argument/return types set; raises ImportError if none is found."""
    global _lib
    if _lib is not None:
        return _lib
    lib, tried = None, []
    for cand in _candidates():
        try:
            lib = ctypes.CDLL(cand)
            break
        except OSError as e:
            tried.append(f"{cand}: {e}")
    if lib is None:
        raise ImportError("no libhdf5 found (set MAPX_HDF5_LIB, install h5py, or convert the file once with "
                          "np.savez(<name>.npz, feat_ids=..., labels=...)); tried: " + "; ".join(tried or ["-"]))
    if lib.H5open() < 0:
        raise ImportError("H5open() failed")
    ver = (ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint())
    lib.H5get_libversion(*(ctypes.byref(v) for v in ver))
    lib.version = tuple(v.value for v in ver)
    hid = ctypes.c_int64 if lib.version >= (1, 10, 0) else ctypes.c_int       # hid_t grew in 1.10
    lib.hid_t = hid
    hsize_p = ctypes.POINTER(ctypes.c_uint64)
    sigs = {
        "H5Fopen": (hid, [ctypes.c_char_p, ctypes.c_uint, hid]),
        "H5Fclose": (ctypes.c_int, [hid]),
        "H5Dopen2": (hid, [hid, ctypes.c_char_p, hid]),
        "H5Dclose": (ctypes.c_int, [hid]),
        "H5Dget_space": (hid, [hid]),
        "H5Dget_type": (hid, [hid]),
        "H5Dread": (ctypes.c_int, [hid, hid, hid, hid, hid, ctypes.c_void_p]),
        "H5Sget_simple_extent_ndims": (ctypes.c_int, [hid]),
        "H5Sget_simple_extent_dims": (ctypes.c_int, [hid, hsize_p, hsize_p]),
        "H5Sclose": (ctypes.c_int, [hid]),
        "H5Tget_class": (ctypes.c_int, [hid]),
        "H5Tget_size": (ctypes.c_size_t, [hid]),
        "H5Tget_order": (ctypes.c_int, [hid]),
        "H5Tget_sign": (ctypes.c_int, [hid]),
        "H5Tclose": (ctypes.c_int, [hid]),
        "H5Eset_auto2": (ctypes.c_int, [hid, ctypes.c_void_p, ctypes.c_void_p]),
        "H5Lexists": (ctypes.c_int, [hid, ctypes.c_char_p, hid]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.H5Eset_auto2(0, None, None)            # errors come back as return codes, not stderr dumps
    _lib = lib
    return lib


def _numpy_dtype(lib, t):
    cls, size, order = lib.H5Tget_class(t), lib.H5Tget_size(t), lib.H5Tget_order(t)
    if cls == H5T_INTEGER:
        kind = "u" if lib.H5Tget_sign(t) == H5T_SGN_NONE else "i"
    elif cls == H5T_FLOAT:
        kind = "f"
    else:
        raise H5Error(f"unsupported HDF5 datatype class {cls} (integers and floats only)")
    if size not in (1, 2, 4, 8):
        raise H5Error(f"unsupported element size {size}")
    return np.dtype(("<" if order == H5T_ORDER_LE else ">") + kind + str(size))


def read_datasets(path, names):
    """Read the named datasets of `path` whole, in their stored type (native byte order)."""
    lib = library()
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    f = lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise H5Error(f"{path}: not an HDF5 file, or unreadable")
    out = {}
    try:
        for name in names:
            if lib.H5Lexists(f, name.encode(), H5P_DEFAULT) <= 0:
                raise KeyError(f"{path}: no dataset {name!r}")
            d = lib.H5Dopen2(f, name.encode(), H5P_DEFAULT)
            if d < 0:
                raise H5Error(f"{path}: cannot open dataset {name!r}")
            s = t = -1
            try:
                s, t = lib.H5Dget_space(d), lib.H5Dget_type(d)
                nd = lib.H5Sget_simple_extent_ndims(s)
                if nd < 0:
                    raise H5Error(f"{path}:{name}: no simple dataspace")
                dims = (ctypes.c_uint64 * max(nd, 1))()
                lib.H5Sget_simple_extent_dims(s, dims, None)
                arr = np.empty(tuple(int(dims[i]) for i in range(nd)), dtype=_numpy_dtype(lib, t))
                if arr.size and lib.H5Dread(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data) < 0:
                    raise H5Error(f"{path}:{name}: H5Dread failed (missing filter plugin?)")
                out[name] = arr if arr.dtype.isnative else arr.astype(arr.dtype.newbyteorder("="))
            finally:
                if t >= 0:
                    lib.H5Tclose(t)
                if s >= 0:
                    lib.H5Sclose(s)
                lib.H5Dclose(d)
    finally:
        lib.H5Fclose(f)
    return out
