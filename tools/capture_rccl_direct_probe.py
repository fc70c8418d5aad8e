"""Development-only: an RCCL all-reduce issued straight through librccl (own communicator, no torch process group)
captured in a hipGraph next to other work -- the route left for putting the gradient exchange inside the step's graph
(the process group's watchdog aborts on works recorded during a capture, tools/capture_collective_probe.py)."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))


class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                              ctypes.c_void_p, ctypes.c_void_p]
lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
lib.ncclGetErrorString.restype = ctypes.c_char_p


def ck(rc):
    if rc != 0:
        raise RuntimeError("rccl: %s" % lib.ncclGetErrorString(rc).decode())


dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
uid = UniqueId()
ck(lib.ncclGetUniqueId(ctypes.byref(uid)))
comm = ctypes.c_void_p()
ck(lib.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0))
main, side = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.set_stream(main)
x = torch.ones(1 << 20, device=dev)
y = torch.zeros_like(x)
NCCL_FLOAT, NCCL_SUM = 7, 0


def allreduce(t, stream):
    ck(lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), NCCL_FLOAT, NCCL_SUM, comm, stream.cuda_stream))


allreduce(x, main)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
    y.add_(x)
    side.wait_stream(main)
    allreduce(y, side)                # the ring on its own branch of the graph
    x.mul_(1.0)                       # work that may overlap it
    main.wait_stream(side)
    y.mul_(0.5)
print("captured", flush=True)
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("replayed, y[0] =", y[0].item(), "(expected %.6f)" % (1.0 - 0.5 ** 10), flush=True)
ck(lib.ncclCommDestroy(comm))
print("PROBE OK", flush=True)
