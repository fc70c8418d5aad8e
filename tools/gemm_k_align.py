"""Development-only: does the 8-byte row alignment of K*Cin = 990 cost the first-layer contraction anything?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
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
for K in (990, 992, 960, 1024):
    for sk in (1, 2, 3):
        A = torch.randn(19464, K, device="cuda"); B = torch.randn(K, 64, device="cuda"); out = torch.zeros(19464, 64, device="cuda")
        os.environ["MVK_GEMM_FORCE"] = "2,1,%d" % sk
        print("K %4d split %d: %.1f us" % (K, sk, timeit(lambda: ops.gemm(A, B, out=out))), flush=True)
