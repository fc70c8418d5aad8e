"""Development-only (GPU box): the masked BatchNorm + LeakyReLU of the big levels, forward and backward, device time of graph-
captured launches (round 5: used to compare two builds of csrc/bn.hip, DESIGN 4.12 h).  usage: python tools/bn_bench.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops = mvkpconv.sub("ops")
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


for (R, D) in ((19464, 32), (19464, 64), (19464, 128), (55070, 32), (55070, 64), (55070, 128), (4986, 128), (4986, 256), (171123, 64)):
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm1d(D).to(dev)
    x = torch.randn(R, D, device=dev, requires_grad=True)
    nv = torch.tensor([R], dtype=torch.int32, device=dev)
    g = torch.randn(R, D, device=dev)
    fwd = timeit(lambda: ops.bn_lrelu(x.detach(), nv, bn, slope=0.1))

    def both():
        y = ops.bn_lrelu(x, nv, bn, slope=0.1)
        y.backward(g)
        x.grad = None
    tot = timeit(both)
    mb = R * D * 4 / 1e6
    print("%7d x %3d (%5.1f MB): forward %6.1f us (%.2f TB/s of 2 passes + 1 write) | forward + backward %6.1f us" % (
        R, D, mb, fwd, 3 * mb / fwd, tot), flush=True)
