"""Data-parallel sharding of spheres over the GPUs of one node.

Spheres are independent in every kernel of the path (subsampling, neighbours, k-NN, group_points,
KPConv never cross batch elements), so each rank processes its own spheres with no data-path
collective; the only exchange is the gradient: one flat fp32 bucket, all-reduced (sum) by RCCL over
xGMI and divided by the world size. The reference has no counterpart (single GPU).
BatchNorm statistics stay per rank (documented deviation from a single-GPU run over the same
spheres; parity is defined on one GPU)."""
import torch
import torch.distributed as dist


class FlatAllReduce:
    """Averages the gradients of `params` across ranks through one contiguous bucket."""

    def __init__(self, params, world=None):
        self.params = [p for p in params if p.requires_grad]
        self.world = world if world is not None else dist.get_world_size()
        self.flat = None

    def __call__(self, grads=None):
        """grads: the gradient tensors to average (default: the current .grad of every parameter; a
        captured graph passes the fixed tensors ITS backward writes)."""
        if grads is None:
            grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        sizes = [g.numel() for g in grads]
        n = sum(sizes)
        if self.flat is None or self.flat.numel() != n or self.flat.device != grads[0].device:
            self.flat = torch.empty(n, device=grads[0].device, dtype=torch.float32)
        torch._foreach_copy_(list(self.flat.split(sizes)), [g.reshape(-1) for g in grads])
        dist.all_reduce(self.flat)
        self.flat.div_(self.world)
        torch._foreach_copy_([g.view(-1) for g in grads], list(self.flat.split(sizes)))


def shard_spheres(n_total, rank, world):
    """Indices of the spheres rank `rank` owns when n_total spheres are dealt round-robin."""
    return list(range(rank, n_total, world))
