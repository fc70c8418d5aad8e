"""Development-only: can an RCCL all-reduce of a one-rank process group be captured in a hipGraph on this runtime?
usage: python tools/capture_collective_probe.py [thread_local|global|relaxed]"""
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
mode = sys.argv[1] if len(sys.argv) > 1 else "thread_local"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
main = torch.cuda.Stream()
torch.cuda.set_stream(main)
x = torch.ones(1 << 20, device=dev)
y = torch.zeros_like(x)
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=main, capture_error_mode=mode):
    y.add_(x)
    w = dist.all_reduce(y, async_op=True)
    x.mul_(1.0)                       # work of the capturing stream that may overlap the ring
    w.wait()
    y.mul_(0.5)
print("captured", flush=True)
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("replayed, y[0] =", y[0].item(), flush=True)
for _ in range(3):
    dist.all_reduce(x)
dist.barrier()
torch.cuda.synchronize()
print("eager collectives after the capture ok", flush=True)
dist.destroy_process_group()
print("PROBE OK", flush=True)
