"""MI355X-native MV-KPConv hot path (gfx950 HIP kernels behind a C ABI).

Sub-modules
    _lib     ctypes binding of libmvkpconv.so (include/mvkpconv.h), no fallback
    ops      torch.autograd wrappers over the C ABI (device pointers + current stream)
    dropin/  host-side mirror of the reference's module paths (cpp_wrappers.*, models.blocks,
             mvpnet.ops.group_points, mvpnet.models.mvpnet_3d, datasets.common pyramid)
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
