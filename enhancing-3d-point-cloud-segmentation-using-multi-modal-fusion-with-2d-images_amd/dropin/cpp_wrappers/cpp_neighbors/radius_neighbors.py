"""Drop-in for ``cpp_wrappers.cpp_neighbors.radius_neighbors`` (reference
KPConv-PyTorch/cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238): ``batch_query`` with NumPy in /
NumPy int32 out and the reference's error messages, computed by csrc/neighbors.hip."""
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


def batch_query(queries, supports, q_batches, s_batches, *, radius=0.1):
    q = _as(queries, np.float32, "query points")
    s = _as(supports, np.float32, "support points")
    ql = _as(q_batches, np.int32, "query batches")
    sl = _as(s_batches, np.int32, "support batches")
    if q.ndim != 2 or q.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : query.shape is not (N, 3)")
    if s.ndim != 2 or s.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : support.shape is not (N, 3)")
    if ql.ndim > 1:
        raise RuntimeError("Wrong dimensions : queries_batches.shape is not (B,) ")
    if sl.ndim > 1:
        raise RuntimeError("Wrong dimensions : supports_batches.shape is not (B,) ")
    if ql.shape[0] != sl.shape[0]:
        raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
    dev = torch.device("cuda", torch.cuda.current_device())
    out = ops.radius_neighbors_batch(torch.from_numpy(q).to(dev), torch.from_numpy(s).to(dev), ql, sl, radius)
    if out.numel() < 1:
        raise RuntimeError("Error")                                                  # wrapper.cpp:201-205
    return out.cpu().numpy()
