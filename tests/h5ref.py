"""The REAL libhdf5 behind ctypes: the independent reader the hand-written HDF5 writer (covidseg_amd/hdf5_min.py) is pinned against.

Test infrastructure only.  The build image carries HDF5 1.10.6 under /opt/conda (library + h5dump / h5ls, no h5py); `available()` is
False where it does not, and the tests that need it skip.  `read_tree(path)` walks a file with the library's own calls (H5Lget_name_by_idx,
H5Oopen, H5Dread, H5Aread) and returns {"/a/b": ndarray, ...} and {("/a", "attr"): value, ...}.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_CANDIDATES = [os.environ.get("HDF5_LIB"), "/opt/conda/lib/libhdf5.so", "libhdf5.so", "libhdf5_serial.so"]
H5DUMP = next((p for p in ("/opt/conda/bin/h5dump", "/usr/bin/h5dump") if os.path.exists(p)), None)
H5LS = next((p for p in ("/opt/conda/bin/h5ls", "/usr/bin/h5ls") if os.path.exists(p)), None)
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    for p in _CANDIDATES:
        if not p:
            continue
        try:
            lib = C.CDLL(p)
            lib.H5open()
            _lib = lib
            return lib
        except OSError:
            continue
    _lib = False
    return False


def available() -> bool:
    return bool(_load())


hid = C.c_int64


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", C.c_uint64), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


def _fn(name, res, *args):
    f = getattr(_load(), name)
    f.restype = res
    f.argtypes = list(args)
    return f


def _g(name):
    return hid.in_dll(_load(), name).value


def version() -> str:
    a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
    _fn("H5get_libversion", C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint))(C.byref(a), C.byref(b), C.byref(c))
    return f"{a.value}.{b.value}.{c.value}"


def _shape(space):
    nd = _fn("H5Sget_simple_extent_ndims", C.c_int, hid)(space)
    if nd <= 0:
        return ()
    dims = (C.c_uint64 * nd)()
    _fn("H5Sget_simple_extent_dims", C.c_int, hid, C.POINTER(C.c_uint64), C.c_void_p)(space, dims, None)
    return tuple(int(d) for d in dims)


def _read_typed(reader, obj, ftype, space):
    """reader(obj, memtype, buf) for a dataset or an attribute; returns an ndarray (numbers), or bytes / list of bytes (strings)"""
    shape = _shape(space)
    n = int(np.prod(shape)) if shape else 1
    cls = _fn("H5Tget_class", C.c_int, hid)(ftype)
    size = _fn("H5Tget_size", C.c_size_t, hid)(ftype)
    if cls == 3:                                                     # H5T_STRING
        if _fn("H5Tis_variable_str", C.c_int, hid)(ftype) > 0:
            buf = (C.c_char_p * max(n, 1))()
            mt = _fn("H5Tcopy", hid, hid)(ftype)
            if n and reader(obj, mt, buf) < 0:
                raise RuntimeError("libhdf5 read (vlen str) failed")
            out = [bytes(buf[i]) if buf[i] is not None else b"" for i in range(n)]
        else:
            raw = C.create_string_buffer(max(n * size, 1))
            mt = _fn("H5Tcopy", hid, hid)(ftype)
            if n and reader(obj, mt, raw) < 0:
                raise RuntimeError("libhdf5 read (fixed str) failed")
            out = [raw.raw[i * size:(i + 1) * size].rstrip(b"\0") for i in range(n)]
        _fn("H5Tclose", C.c_int, hid)(mt)
        return out[0] if shape == () else out
    if cls == 1:                                                     # H5T_FLOAT
        dt, mt = (np.float32, _g("H5T_NATIVE_FLOAT_g")) if size == 4 else (np.float64, _g("H5T_NATIVE_DOUBLE_g"))
    elif cls == 0:                                                   # H5T_INTEGER: read widened to int64
        dt, mt = np.int64, _g("H5T_NATIVE_LLONG_g")
    else:
        raise RuntimeError(f"h5ref: datatype class {cls} not handled")
    a = np.zeros(max(n, 1), dt)
    if n and reader(obj, mt, a.ctypes.data_as(C.c_void_p)) < 0:
        raise RuntimeError("libhdf5 read failed")
    return a[:n].reshape(shape), (cls, size)


def read_tree(path):
    """-> (datasets {abs path: ndarray}, file-type facts {abs path: (class, size)}, attrs {(abs object path, name): value}, groups [abs path])"""
    if not available():
        raise RuntimeError("libhdf5 not found")
    _fn("H5Eset_auto2", C.c_int, hid, C.c_void_p, C.c_void_p)(0, None, None)          # errors come back as return codes
    f = _fn("H5Fopen", hid, C.c_char_p, C.c_uint, hid)(os.fsencode(path), 0, 0)
    if f < 0:
        raise RuntimeError(f"libhdf5 refuses to open {path}")
    dsets, facts, attrs, groups = {}, {}, {}, []
    Oopen = _fn("H5Oopen", hid, hid, C.c_char_p, hid)
    Oclose = _fn("H5Oclose", C.c_int, hid)
    Iget = _fn("H5Iget_type", C.c_int, hid)
    Ginfo = _fn("H5Gget_info", C.c_int, hid, C.POINTER(_GInfo))
    Lname = _fn("H5Lget_name_by_idx", C.c_ssize_t, hid, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_char_p, C.c_size_t, hid)
    Aopen = _fn("H5Aopen_by_idx", hid, hid, C.c_char_p, C.c_int, C.c_int, C.c_uint64, hid, hid)
    Aname = _fn("H5Aget_name", C.c_ssize_t, hid, C.c_size_t, C.c_char_p)
    Aread = _fn("H5Aread", C.c_int, hid, hid, C.c_void_p)
    Dread5 = _fn("H5Dread", C.c_int, hid, hid, hid, hid, hid, C.c_void_p)

    def obj_attrs(o, where):
        i = 0
        while True:
            a = Aopen(o, b".", 0, 0, i, 0, 0)                        # H5_INDEX_NAME, H5_ITER_INC
            if a < 0:
                break
            nb = C.create_string_buffer(1024)
            Aname(a, 1024, nb)
            t = _fn("H5Aget_type", hid, hid)(a)
            sp = _fn("H5Aget_space", hid, hid)(a)
            v = _read_typed(lambda ob, mt, buf: Aread(ob, mt, buf), a, t, sp)
            attrs[(where, nb.value.decode("utf8"))] = v[0] if isinstance(v, tuple) else v
            _fn("H5Sclose", C.c_int, hid)(sp); _fn("H5Tclose", C.c_int, hid)(t); _fn("H5Aclose", C.c_int, hid)(a)
            i += 1

    def walk(g, where):
        groups.append(where or "/")
        obj_attrs(g, where or "/")
        info = _GInfo()
        if Ginfo(g, C.byref(info)) < 0:
            raise RuntimeError("H5Gget_info failed")
        for i in range(info.nlinks):
            nb = C.create_string_buffer(1024)
            if Lname(g, b".", 0, 0, i, nb, 1024, 0) < 0:
                raise RuntimeError("H5Lget_name_by_idx failed")
            child = where + "/" + nb.value.decode("utf8")
            o = Oopen(g, nb.value, 0)
            if o < 0:
                raise RuntimeError(f"libhdf5 cannot open {child}")
            kind = Iget(o)
            if kind == 2:                                            # H5I_GROUP
                walk(o, child)
            elif kind == 5:                                          # H5I_DATASET
                t = _fn("H5Dget_type", hid, hid)(o)
                sp = _fn("H5Dget_space", hid, hid)(o)
                v = _read_typed(lambda ob, mt, buf: Dread5(ob, mt, 0, 0, 0, buf), o, t, sp)
                dsets[child], facts[child] = v if isinstance(v, tuple) else (v, (3, 0))
                obj_attrs(o, child)
                _fn("H5Sclose", C.c_int, hid)(sp); _fn("H5Tclose", C.c_int, hid)(t)
            Oclose(o)

    root = Oopen(f, b"/", 0)
    walk(root, "")
    Oclose(root)
    if _fn("H5Fclose", C.c_int, hid)(f) < 0:
        raise RuntimeError("H5Fclose failed")
    return dsets, facts, attrs, groups
