"""Drop-in for the reference CPython extension ``cpp_wrappers.cpp_subsampling.grid_subsampling``
(KPConv-PyTorch/cpp_wrappers/cpp_subsampling/wrapper.cpp): same function names, keyword-only
options, NumPy in / NumPy out, dtypes, return tuples and RuntimeError messages -- computed by the
HIP kernel (csrc/subsample.hip) through the C ABI. Inputs are staged to HBM and results copied
back, exactly once each; device-resident callers use ``ops.grid_subsample_batch`` directly."""
import numpy as np
import torch

try:
    from ..._native import ops
except ImportError:
    from _native import ops


def _as(obj, dtype, what):
    try:
        return np.ascontiguousarray(obj, dtype=dtype)
    except (TypeError, ValueError):
        raise RuntimeError("Error converting %s to numpy arrays of type %s" % (what, np.dtype(dtype).name))


def _check_and_stage(points, features, classes):
    pts = _as(points, np.float32, "input points")
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")          # wrapper.cpp:389
    N = pts.shape[0]
    f = l = None
    if features is not None:
        f = _as(features, np.float32, "input features")
        if f.ndim != 2:
            raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")
        if f.shape[0] != N:
            raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")
    if classes is not None:
        l = _as(classes, np.int32, "input classes")
        if l.ndim > 2:
            raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")
        if l.shape[0] != N:
            raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")
    dev = torch.device("cuda", torch.cuda.current_device())
    tp = torch.from_numpy(pts).to(dev)
    tf = torch.from_numpy(f).to(dev) if f is not None else None
    tl = torch.from_numpy(l.reshape(N, -1)).to(dev) if l is not None else None
    return tp, tf, tl


def subsample_batch(points, batches, *, features=None, classes=None, sampleDl=0.1, method='barycenters',
                    max_p=0, verbose=0):
    """(s_points f32 (M,3), s_len i32 (B,)[, s_features f32 (M,d)][, s_classes i32 (M,l)])."""
    if method not in ('barycenters', 'voxelcenters'):
        raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')
    lens = _as(batches, np.int32, "input batches")
    if lens.ndim > 1:
        raise RuntimeError("Wrong dimensions : batches.shape is not (B,) ")
    tp, tf, tl = _check_and_stage(points, features, classes)
    if int(lens.sum()) != tp.shape[0]:
        raise RuntimeError("Wrong batch lengths: the sum of batches is not the number of points")
    res = ops.grid_subsample_batch(tp, lens, features=tf, labels=tl, dl=sampleDl, max_p=max_p)
    if res[0].shape[0] < 1:
        raise RuntimeError("Error")                                                  # wrapper.cpp:277-281
    out = [res[0].cpu().numpy(), res[1]]
    out += [r.cpu().numpy() for r in res[2:]]
    return tuple(out)


def subsample(points, *, features=None, classes=None, sampleDl=0.1, method='barycenters', verbose=0):
    """s_points | (s_points, s_features) | (s_points, s_classes) | (s_points, s_features, s_classes)."""
    if method not in ('barycenters', 'voxelcenters'):
        raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')
    tp, tf, tl = _check_and_stage(points, features, classes)
    res = ops.grid_subsample_batch(tp, [tp.shape[0]], features=tf, labels=tl, dl=sampleDl)
    if res[0].shape[0] < 1:
        raise RuntimeError("Error")                                                  # wrapper.cpp:505-509
    out = [res[0].cpu().numpy()] + [r.cpu().numpy() for r in res[2:]]
    return out[0] if len(out) == 1 else tuple(out)
