"""Test helper: drive the real libhdf5 (if the image has one) through ctypes, to check the
pure-Python reader/writer in deepptmpred_amd/hdf5_min.py against the reference library."""
import ctypes as C
import glob
import os

import numpy as np

_CANDIDATES = ["/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so"] + \
    sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))


def load():
    for p in _CANDIDATES:
        if os.path.exists(p):
            try:
                lib = C.CDLL(p)
            except OSError:
                continue
            lib.H5open()
            hid = C.c_int64
            for name, res, args in [
                ("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]),
                ("H5Fclose", C.c_int, [hid]), ("H5Gcreate2", hid, [hid, C.c_char_p, hid, hid, hid]),
                ("H5Gclose", C.c_int, [hid]), ("H5Screate_simple", hid, [C.c_int, C.c_void_p, C.c_void_p]),
                ("H5Sclose", C.c_int, [hid]), ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
                ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), ("H5Dclose", C.c_int, [hid]),
                ("H5Dopen2", hid, [hid, C.c_char_p, hid]), ("H5Dread", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
                ("H5Dget_space", hid, [hid]), ("H5Sget_simple_extent_ndims", C.c_int, [hid]),
                ("H5Sget_simple_extent_dims", C.c_int, [hid, C.c_void_p, C.c_void_p]),
                ("H5Tcopy", hid, [hid]), ("H5Tset_size", C.c_int, [hid, C.c_size_t]), ("H5Tclose", C.c_int, [hid]),
                ("H5Tget_size", C.c_size_t, [hid]),
                ("H5Acreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid]), ("H5Awrite", C.c_int, [hid, hid, C.c_void_p]),
                ("H5Aclose", C.c_int, [hid]), ("H5Aopen_by_name", hid, [hid, C.c_char_p, C.c_char_p, hid, hid]),
                ("H5Aread", C.c_int, [hid, hid, C.c_void_p]), ("H5Aget_type", hid, [hid]), ("H5Aget_space", hid, [hid]),
                ("H5Oopen", hid, [hid, C.c_char_p, hid]), ("H5Oclose", C.c_int, [hid]),
            ]:
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            lib.T_FLOAT = C.c_int64.in_dll(lib, "H5T_NATIVE_FLOAT_g").value
            lib.T_C_S1 = C.c_int64.in_dll(lib, "H5T_C_S1_g").value
            return lib
    return None


def write_tree(lib, path, datasets, str_attrs):
    """datasets: {path: float32 array}; str_attrs: {(obj_path, attr): [bytes, ...]} as fixed-length string arrays."""
    f = lib.H5Fcreate(path.encode(), 2, 0, 0)
    assert f >= 0
    made = set()

    def ensure_group(p):
        parts = [x for x in p.split("/") if x]
        cur = ""
        for part in parts:
            cur += "/" + part
            if cur not in made:
                g = lib.H5Gcreate2(f, cur.encode(), 0, 0, 0)
                assert g >= 0, cur
                lib.H5Gclose(g)
                made.add(cur)

    for p, arr in datasets.items():
        ensure_group(os.path.dirname(p))
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        dims = (C.c_uint64 * max(arr.ndim, 1))(*arr.shape)
        s = lib.H5Screate_simple(arr.ndim, dims, None)
        d = lib.H5Dcreate2(f, p.encode(), lib.T_FLOAT, s, 0, 0, 0)
        assert d >= 0, p
        assert lib.H5Dwrite(d, lib.T_FLOAT, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)) >= 0
        lib.H5Dclose(d)
        lib.H5Sclose(s)
    for (obj, name), values in str_attrs.items():
        ensure_group(obj)
        width = max(len(v) for v in values)
        arr = np.array(values, dtype=f"S{width}")
        t = lib.H5Tcopy(lib.T_C_S1)
        lib.H5Tset_size(t, width)
        dims = (C.c_uint64 * 1)(len(values))
        s = lib.H5Screate_simple(1, dims, None)
        o = lib.H5Oopen(f, (obj or "/").encode(), 0)
        a = lib.H5Acreate2(o, name.encode(), t, s, 0, 0)
        assert a >= 0
        assert lib.H5Awrite(a, t, arr.ctypes.data_as(C.c_void_p)) >= 0
        lib.H5Aclose(a)
        lib.H5Oclose(o)
        lib.H5Sclose(s)
        lib.H5Tclose(t)
    lib.H5Fclose(f)


def read_dataset(lib, path, name):
    f = lib.H5Fopen(path.encode(), 0, 0)
    assert f >= 0, "libhdf5 cannot open the file"
    d = lib.H5Dopen2(f, name.encode(), 0)
    assert d >= 0, f"libhdf5 cannot open dataset {name}"
    s = lib.H5Dget_space(d)
    nd = lib.H5Sget_simple_extent_ndims(s)
    dims = (C.c_uint64 * max(nd, 1))()
    lib.H5Sget_simple_extent_dims(s, dims, None)
    out = np.empty(tuple(dims[:nd]), dtype=np.float32)
    assert lib.H5Dread(d, lib.T_FLOAT, 0, 0, 0, out.ctypes.data_as(C.c_void_p)) >= 0
    lib.H5Sclose(s)
    lib.H5Dclose(d)
    lib.H5Fclose(f)
    return out


def read_str_attr(lib, path, obj, name):
    f = lib.H5Fopen(path.encode(), 0, 0)
    a = lib.H5Aopen_by_name(f, (obj or "/").encode(), name.encode(), 0, 0)
    assert a >= 0, f"libhdf5 cannot open attribute {obj}@{name}"
    t = lib.H5Aget_type(a)
    width = lib.H5Tget_size(t)
    s = lib.H5Aget_space(a)
    dims = (C.c_uint64 * 1)()
    lib.H5Sget_simple_extent_dims(s, dims, None)
    out = np.empty(dims[0], dtype=f"S{width}")
    assert lib.H5Aread(a, t, out.ctypes.data_as(C.c_void_p)) >= 0
    lib.H5Sclose(s)
    lib.H5Tclose(t)
    lib.H5Aclose(a)
    lib.H5Fclose(f)
    return [bytes(x) for x in out]
