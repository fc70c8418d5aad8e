"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the C oracle (and of the
compiled reference core when ``oracle/_ref/libref.so`` exists).

Functions mirror the reference wrapper semantics
(KPConv-PyTorch/cpp_wrappers/cpp_subsampling/wrapper.cpp:62-333,
 KPConv-PyTorch/cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)


def build(ref=True):
    """Compile liboracle.so (always) and _ref/libref.so (when /root/reference is present)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.check_call(["make", "-s", "-C", _HERE] + targets)


def _load(path):
    return C.CDLL(path) if os.path.exists(path) else None


_ORC = None
_REF = None


def orc():
    global _ORC
    if _ORC is None:
        p = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(p):
            build(ref=False)
        _ORC = C.CDLL(p)
        _ORC.orc_grid_subsample_batch.restype = C.c_long
        _ORC.orc_grid_subsample.restype = C.c_long
        _ORC.orc_radius_neighbors_batch.restype = C.c_int
        _ORC.orc_knn_f64.restype = None
    return _ORC


def ref():
    """The compiled reference core, or None when it was not built (GPU box without prebuilt file)."""
    global _REF
    if _REF is None:
        _REF = _load(os.path.join(_HERE, "_ref", "libref.so"))
        if _REF is not None:
            _REF.ref_subsample_batch.restype = C.c_long
            _REF.ref_subsample.restype = C.c_long
            _REF.ref_radius_neighbors_batch.restype = C.c_int
    return _REF


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def subsample_batch(points, lens, features=None, labels=None, dl=0.1, max_p=0, impl="oracle"):
    pts = _f32(points)
    lens = _i32(lens)
    N, B = pts.shape[0], lens.shape[0]
    f = _f32(features) if features is not None else None
    l = _i32(labels) if labels is not None else None
    if l is not None and l.ndim == 1:
        l = l.reshape(-1, 1)
    fdim = f.shape[1] if f is not None else 0
    ldim = l.shape[1] if l is not None else 0
    op = np.empty((max(N, 1), 3), np.float32)
    of = np.empty((max(N, 1), fdim), np.float32) if f is not None else None
    ol = np.empty((max(N, 1), ldim), np.int32) if l is not None else None
    olen = np.empty((B,), np.int32)
    lib = orc() if impl == "oracle" else ref()
    fn = lib.orc_grid_subsample_batch if impl == "oracle" else lib.ref_subsample_batch
    M = fn(_ptr(pts, _fp), C.c_long(N), _ptr(f, _fp), C.c_int(fdim), _ptr(l, _ip), C.c_int(ldim),
           _ptr(lens, _ip), C.c_int(B), C.c_float(dl), C.c_int(max_p),
           _ptr(op, _fp), _ptr(of, _fp), _ptr(ol, _ip), _ptr(olen, _ip))
    out = [op[:M].copy(), olen]
    if f is not None:
        out.append(of[:M].copy())
    if l is not None:
        out.append(ol[:M].copy())
    return tuple(out)


def subsample(points, features=None, labels=None, dl=0.1, impl="oracle"):
    pts = _f32(points)
    N = pts.shape[0]
    f = _f32(features) if features is not None else None
    l = _i32(labels) if labels is not None else None
    if l is not None and l.ndim == 1:
        l = l.reshape(-1, 1)
    fdim = f.shape[1] if f is not None else 0
    ldim = l.shape[1] if l is not None else 0
    op = np.empty((max(N, 1), 3), np.float32)
    of = np.empty((max(N, 1), fdim), np.float32) if f is not None else None
    ol = np.empty((max(N, 1), ldim), np.int32) if l is not None else None
    lib = orc() if impl == "oracle" else ref()
    fn = lib.orc_grid_subsample if impl == "oracle" else lib.ref_subsample
    M = fn(_ptr(pts, _fp), C.c_long(N), _ptr(f, _fp), C.c_int(fdim), _ptr(l, _ip), C.c_int(ldim),
           C.c_float(dl), _ptr(op, _fp), _ptr(of, _fp), _ptr(ol, _ip))
    out = [op[:M].copy()]
    if f is not None:
        out.append(of[:M].copy())
    if l is not None:
        out.append(ol[:M].copy())
    return out[0] if len(out) == 1 else tuple(out)


def radius_neighbors_batch(queries, supports, q_lens, s_lens, radius, impl="oracle"):
    q, s = _f32(queries), _f32(supports)
    ql, sl = _i32(q_lens), _i32(s_lens)
    Nq, Ns, B = q.shape[0], s.shape[0], ql.shape[0]
    lib = orc() if impl == "oracle" else ref()
    fn = lib.orc_radius_neighbors_batch if impl == "oracle" else lib.ref_radius_neighbors_batch
    args = (_ptr(q, _fp), C.c_long(Nq), _ptr(s, _fp), C.c_long(Ns), _ptr(ql, _ip), _ptr(sl, _ip),
            C.c_int(B), C.c_float(radius))
    W = fn(*args, None)
    out = np.empty((Nq, W), np.int32)
    if W > 0:
        fn(*args, _ptr(out, _ip))
    return out


def knn_f64(queries, keys, k=3):
    q = np.ascontiguousarray(queries, dtype=np.float64)
    kk = np.ascontiguousarray(keys, dtype=np.float64)
    idx = np.empty((q.shape[0], k), np.int64)
    d2 = np.empty((q.shape[0], k), np.float64)
    orc().orc_knn_f64(_ptr(q, _dp), C.c_long(q.shape[0]), _ptr(kk, _dp), C.c_long(kk.shape[0]),
                      C.c_int(k), _ptr(idx, _lp), _ptr(d2, _dp))
    return idx, d2
