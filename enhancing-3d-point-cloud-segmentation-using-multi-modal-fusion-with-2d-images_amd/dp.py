"""Data-parallel sharding of spheres over the GPUs of one node.

Spheres are independent in every kernel of the path (subsampling, neighbours, k-NN, group_points,
KPConv never cross batch elements), so each rank processes its own spheres with no data-path
collective; the only exchange is the gradient: flat fp32 buckets, all-reduced (sum) by RCCL over
xGMI and divided by the world size. The reference has no counterpart (single GPU).
BatchNorm statistics stay per rank (documented deviation from a single-GPU run over the same
spheres; parity is defined on one GPU).

Overlap. xGMI is point to point (7 links x ~153 GB/s per GPU): a ring all-reduce of the 97.5 MB of KPFCNN
gradients costs ~1.1 ms, 15-20 % of a step, if it only starts after the backward. The backward runs from the
head through the decoder and then down the encoder from the coarsest level: the levels that hold almost all of
the PARAMETERS (levels 2-4: 15 x 512 x 512 kernels) are finished long before the levels that hold almost all of
the POINTS (levels 0-1), which take about half of the backward time and own a few percent of the bytes.
`two_stage_backward` therefore cuts the backward at the entry of encoder level `cut_layer`: stage 1 produces the
gradients of everything above the cut (bucket 0, ~95 % of the bytes), whose all-reduce is started asynchronously
and runs on RCCL's stream while stage 2 (levels below the cut) is still computing; bucket 1 follows.
"""
import ctypes
import os

import torch
import torch.distributed as dist


class RcclCommunicator:
    """An RCCL communicator of our own over the ranks of the default process group, driven through librccl's C API.

    Why not the process group: its collectives cannot be captured in a hipGraph on this runtime (the group's watchdog
    thread polls the event of a work recorded during the capture and aborts with hipErrorCapturedEvent,
    tools/capture_collective_probe.py); an all-reduce issued straight through `ncclAllReduce` on a stream of ours
    captures and replays (tools/capture_rccl_direct_probe.py), which puts the gradient exchange INSIDE the step's graph
    as one more branch. The process group is only used once, to hand rank 0's unique id to the other ranks."""
    _FLOAT32, _SUM = 7, 0                   # ncclFloat32, ncclSum (rccl.h)

    class _UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_byte * 128)]

    def __init__(self, device):
        lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(self._UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, self._UniqueId, ctypes.c_int]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        self.lib, self.device = lib, torch.device(device)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        uid = self._UniqueId()
        if self.rank == 0:
            self._ck(lib.ncclGetUniqueId(ctypes.byref(uid)))
        box = [bytes(bytearray(uid.internal)) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctypes.memmove(uid.internal, box[0], 128)
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._ck(lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank))

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("RCCL: %s" % self.lib.ncclGetErrorString(rc).decode())

    def all_reduce_(self, t, stream):
        """In-place sum of the contiguous fp32 device tensor `t` over the ranks, enqueued on `stream`."""
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
            raise ValueError("RcclCommunicator.all_reduce_: contiguous float32 tensor on %s expected" % self.device)
        self._ck(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), self._FLOAT32, self._SUM, self.comm,
                                        stream.cuda_stream))

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()


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


class BucketedAllReduce:
    """Gradient exchange in buckets that can be started one by one while the backward is still running.

    pack(k) / unpack(k) are plain device copies (capturable in a hipGraph); launch(k) issues the asynchronous
    all-reduce of bucket k (an eager RCCL call: it runs on the process group's own stream, ordered after the work
    already enqueued on the current stream); wait() blocks the current stream until every launched bucket is done.

    With `comm` (an RcclCommunicator) the all-reduces go straight to librccl on a stream of this object, forked from
    and joined to the current stream: the whole exchange is then capturable (`capturable` is True) and becomes a
    branch of the step's graph.

    prescaled=True: the backward was seeded with 1 / world (`seed()`; exact for the power-of-two world sizes of a
    node), so the summed bucket already holds the mean and unpack(k) only re-points every `.grad` at its slice of the
    bucket -- no division pass and no copy back (two passes over the ~100 MB of KPFCNN gradients per step)."""

    def __init__(self, buckets, world=None, comm=None, prescaled=False):
        self.buckets = [[p for p in b if p.requires_grad] for b in buckets]
        self.world = world if world is not None else dist.get_world_size()
        self.flat = [None] * len(self.buckets)
        self.grads = [None] * len(self.buckets)
        self.owners = [None] * len(self.buckets)
        self.work = []
        self.prescaled = bool(prescaled)
        self._seed = None
        self.comm = comm
        self.capturable = comm is not None
        self.comm_stream = torch.cuda.Stream(device=comm.device) if comm is not None else None
        self._forked = False

    def seed(self, like):
        """The gradient to start the backward with: 1 / world when prescaled, else None (= 1)."""
        if not self.prescaled:
            return None
        if self._seed is None or self._seed.device != like.device or self._seed.dtype != like.dtype:
            self._seed = torch.full((), 1.0 / self.world, device=like.device, dtype=like.dtype)
        return self._seed

    def pack(self, k, grads=None):
        if grads is None:
            self.owners[k] = [p for p in self.buckets[k] if p.grad is not None]
            grads = [p.grad for p in self.owners[k]]
        else:
            self.owners[k] = None
        self.grads[k] = grads
        if not grads:
            return
        n = sum(g.numel() for g in grads)
        if self.flat[k] is None or self.flat[k].numel() != n or self.flat[k].device != grads[0].device:
            self.flat[k] = torch.empty(n, device=grads[0].device, dtype=torch.float32)
        torch._foreach_copy_(list(self.flat[k].split([g.numel() for g in grads])), [g.reshape(-1) for g in grads])

    def launch(self, k):
        if not self.grads[k]:
            return
        if self.comm is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            self.comm.all_reduce_(self.flat[k], self.comm_stream)
            self._forked = True
            return
        if dist.get_backend() == "nccl":        # RCCL: asynchronous on the group's stream, overlaps the rest of the backward
            self.work.append(dist.all_reduce(self.flat[k], async_op=True))
        else:                                   # gloo (CPU rehearsal): its asynchronous path on device tensors is far slower
            dist.all_reduce(self.flat[k])

    def wait(self):
        if self._forked:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            self._forked = False
        for w in self.work:
            w.wait()
        self.work = []

    def unpack(self, k):
        grads = self.grads[k]
        if not grads:
            return
        if self.prescaled:
            if self.owners[k] is None:
                raise RuntimeError("BucketedAllReduce(prescaled=True): pack() must collect the gradients itself")
            for p, v in zip(self.owners[k], self.flat[k].split([g.numel() for g in grads])):
                p.grad = v.view_as(p)
            return
        self.flat[k].div_(self.world)
        torch._foreach_copy_([g.view(-1) for g in grads], list(self.flat[k].split([g.numel() for g in grads])))

    def __call__(self):
        """Everything at once (no overlap): the FlatAllReduce behaviour with several buckets."""
        for k in range(len(self.buckets)):
            self.pack(k)
            self.launch(k)
        self.wait()
        for k in range(len(self.buckets)):
            self.unpack(k)


def split_parameters_at(net, cut_block):
    """(late, early) parameter lists of a KPFCNN-style network for a cut at encoder block `cut_block`:
    early = parameters of net.encoder_blocks[:cut_block]; late = every other trainable parameter."""
    early = [p for blk in list(net.encoder_blocks)[:cut_block] for p in blk.parameters() if p.requires_grad]
    ids = {id(p) for p in early}
    late = [p for p in net.parameters() if p.requires_grad and id(p) not in ids]
    return late, early


def cut_block_of_layer(architecture, layer):
    """Index of the first encoder block that runs at pyramid layer `layer` (blocks after the layer-th strided /
    pooling block); None when the architecture is shallower."""
    seen = 0
    for i, name in enumerate(architecture):
        if 'upsample' in name or 'global' in name:
            return None
        if seen == layer:
            return i
        if 'strided' in name or 'pool' in name:
            seen += 1
    return None


def deformable_below(architecture, cut_block):
    """True when a deformable block lies below the cut: the regulariser then reaches its KPConv node in stage 1 (through
    min_d2) and the feature path reaches the SAME node again in stage 2 -- stage 1 must keep the graph
    (two_stage_backward(retain_graph=True)). The shipped cuts (entry of level 2) lie below every deformable block."""
    return any('deformable' in b for b in list(architecture)[:cut_block])


def two_stage_backward(loss, cut_tensors, between=None, backward_scope=None, seed=None, retain_graph=False):
    """loss.backward() in two pieces around a severed graph: `cut_tensors` = (originals, leaves) as recorded by
    run_encoder_decoder when net.backward_cut is set (everything downstream of the cut was computed from the
    detached leaves). Stage 1 = loss.backward(): gradients of the parameters above the cut and of the leaves;
    `between()` (e.g. start the all-reduce of the late bucket); stage 2 = backward of the originals with the leaves'
    gradients: the parameters below the cut. The sum of both stages is exactly what an unsevered loss.backward()
    computes. backward_scope: context manager factory wrapped around each stage (ops.overlap_weight_grads).
    seed: gradient of the loss to start from (BucketedAllReduce.seed). retain_graph: see deformable_below()."""
    import contextlib
    scope = backward_scope if backward_scope is not None else contextlib.nullcontext
    orig, leaves = cut_tensors
    with scope():
        loss.backward(seed, retain_graph=bool(retain_graph))
    if between is not None:
        between()
    pairs = [(t, l.grad) for t, l in zip(orig, leaves) if t.requires_grad and l.grad is not None]
    if pairs:
        with scope():
            torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])


def shard_spheres(n_total, rank, world):
    """Indices of the spheres rank `rank` owns when n_total spheres are dealt round-robin."""
    return list(range(rank, n_total, world))
