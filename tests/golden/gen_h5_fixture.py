#!/usr/bin/env python3
"""Writes tests/golden/tiny_table.h5 (+ tiny_table_expected.npz) with the HDF5 C library itself
(ctypes on libhdf5 — h5py is not installed here), in the shape the reference's preprocessing
produces (data_preprocess/proc_avazu.py:284-288: datasets feat_ids / field_ids / type_ids /
labels via create_dataset(data=...)): contiguous int64 datasets, plus one chunked + gzip float32
dataset and one big-endian int16 dataset to pin the reader's type/layout handling.

    python tests/golden/gen_h5_fixture.py
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "map-code_amd"))
from mapx import h5lite  # noqa: E402

H5F_ACC_TRUNC = 2


def main():
    lib = h5lite.library()
    hid = lib.hid_t
    g = lambda name: hid.in_dll(lib, name).value          # noqa: E731  library-global type / class ids
    for fn, res, args in (("H5Fcreate", hid, [ctypes.c_char_p, ctypes.c_uint, hid, hid]),
                          ("H5Screate_simple", hid, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]),
                          ("H5Dcreate2", hid, [hid, ctypes.c_char_p, hid, hid, hid, hid, hid]),
                          ("H5Dwrite", ctypes.c_int, [hid, hid, hid, hid, hid, ctypes.c_void_p]),
                          ("H5Pcreate", hid, [hid]), ("H5Pclose", ctypes.c_int, [hid]),
                          ("H5Pset_chunk", ctypes.c_int, [hid, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
                          ("H5Pset_deflate", ctypes.c_int, [hid, ctypes.c_uint])):
        f = getattr(lib, fn)
        f.restype, f.argtypes = res, args
    rng = np.random.default_rng(7)
    N, F = 64, 5
    data = {
        "feat_ids": rng.integers(0, 1000, size=(N, F), dtype=np.int64),
        "field_ids": np.tile(np.arange(F, dtype=np.int64), (N, 1)),
        "type_ids": np.zeros((N, F), dtype=np.int64),
        "labels": rng.integers(0, 2, size=(N,), dtype=np.int64),
        "chunked_f32": rng.standard_normal((37, 3)).astype(np.float32),
        "be_i16": rng.integers(-300, 300, size=(11,), dtype=np.int16),
    }
    path = os.path.join(HERE, "tiny_table.h5")
    fid = lib.H5Fcreate(path.encode(), H5F_ACC_TRUNC, 0, 0)
    assert fid >= 0
    for name, arr in data.items():
        dims = (ctypes.c_uint64 * arr.ndim)(*arr.shape)
        space = lib.H5Screate_simple(arr.ndim, dims, None)
        mem = {"int64": "H5T_NATIVE_INT64_g", "float32": "H5T_NATIVE_FLOAT_g", "int16": "H5T_NATIVE_INT16_g"}[str(arr.dtype)]
        file_type = g("H5T_STD_I16BE_g") if name == "be_i16" else g(mem)
        dcpl = 0
        if name == "chunked_f32":
            dcpl = lib.H5Pcreate(g("H5P_CLS_DATASET_CREATE_ID_g"))
            chunk = (ctypes.c_uint64 * 2)(8, 3)
            assert lib.H5Pset_chunk(dcpl, 2, chunk) >= 0 and lib.H5Pset_deflate(dcpl, 4) >= 0
        d = lib.H5Dcreate2(fid, name.encode(), file_type, space, 0, dcpl, 0)
        assert d >= 0, name
        assert lib.H5Dwrite(d, g(mem), 0, 0, 0, np.ascontiguousarray(arr).ctypes.data) >= 0
        lib.H5Dclose(d)
        lib.H5Sclose(space)
        if dcpl:
            lib.H5Pclose(dcpl)
    lib.H5Fclose(fid)
    np.savez(os.path.join(HERE, "tiny_table_expected.npz"), **data)
    print("wrote", path, os.path.getsize(path), "bytes with libhdf5", ".".join(map(str, lib.version)))


if __name__ == "__main__":
    main()
