"""``group_points(points, index)`` with the reference's signature and autograd behaviour
(mvpnet/ops/group_points.py:5-31) on the HIP gather / scatter-add kernels (csrc/fusion.hip)."""
try:
    from ..._native import ops
except ImportError:
    from _native import ops


def group_points(points, index):
    """Gather points by index.

    Args:
        points (torch.Tensor): (batch_size, channels, num_points)
        index (torch.Tensor): (batch_size, num_centroids, num_neighbors) int64
    Returns:
        torch.Tensor: (batch_size, channels, num_centroids, num_neighbors)
    """
    return ops.group_points(points, index)
