"""Development-only: masked BatchNorm + LeakyReLU forward and backward at the coarse-level shapes (graph-timed)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops = mvkpconv.sub("ops")
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
for R, D in ((225, 256), (225, 64), (923, 128), (923, 32), (923, 512), (3986, 64)):
    x = torch.randn(R, D, device="cuda", requires_grad=True)
    bn = torch.nn.BatchNorm1d(D).cuda()
    nv = torch.tensor([R], dtype=torch.int32, device="cuda")
    go = torch.randn(R, D, device="cuda")
    def fwd():
        with torch.no_grad():
            return ops.bn_lrelu(x, nv, bn, slope=0.1)
    def both():
        y = ops.bn_lrelu(x, nv, bn, slope=0.1)
        torch.autograd.grad(y, [x, bn.weight, bn.bias], go)
    print("R %5d D %4d : fwd %.1f us  fwd+bwd %.1f us" % (R, D, timeit(fwd), timeit(both)), flush=True)
