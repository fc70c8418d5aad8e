"""The gradient exchange through librccl inside a hipGraph, on a one-rank communicator (run by tests/test_gpu_parity.py
in a child process: it owns a process group). Two buckets packed, all-reduced on the exchange branch while other work
runs on the capturing stream, unpacked; ten replays with changing gradients must reproduce the gradients (world 1:
sum / 1)."""
import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
dp = mvkpconv.sub("dp")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
main = torch.cuda.Stream()
torch.cuda.set_stream(main)
g0 = torch.Generator(device="cpu").manual_seed(5)
a = [torch.nn.Parameter(torch.zeros(n, device=dev)) for n in (1 << 20, 4097, 3)]
b = [torch.nn.Parameter(torch.zeros(n, device=dev)) for n in (515, 64 * 64)]
src = [torch.randn(p.numel(), generator=g0).to(dev) for p in a + b]
for p in a + b:
    p.grad = torch.zeros_like(p)
comm = dp.RcclCommunicator(dev)
red = dp.BucketedAllReduce([a, b], world=1, comm=comm)
assert red.capturable
scale = torch.ones((), device=dev)
busy = torch.zeros(1 << 22, device=dev)


def step():
    for p, s in zip(a + b, src):
        p.grad.copy_(s * scale)
    red.pack(0)
    red.launch(0)
    busy.add_(1.0)                     # stands for the backward below the cut
    red.pack(1)
    red.launch(1)
    red.wait()
    red.unpack(0)
    red.unpack(1)


step()                                 # eager once (allocates the flat buckets)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=main, capture_error_mode="thread_local"):
    step()
for k in range(10):
    scale.fill_(float(k + 1))
    graph.replay()
    torch.cuda.synchronize()
    for p, s in zip(a + b, src):
        assert torch.equal(p.grad, s * float(k + 1)), "replay %d: gradient changed by the exchange" % k
assert busy[0].item() == 11.0
dist.barrier()
comm.close()
dist.destroy_process_group()
print("DP GRAPH EXCHANGE OK")
