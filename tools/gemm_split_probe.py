"""Development-only (GPU box): the split coarse-level products over the split count, atomic epilogue
against the ordered hand-off (ops.set_deterministic), device time of graph-captured launches.
usage: python tools/gemm_split_probe.py"""
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


for (M, N, K) in ((85, 512, 7680), (330, 256, 3840), (1300, 128, 1920), (4986, 64, 960)):
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev)
    out = torch.zeros(M, N, device=dev)
    floor = 2.0 * M * N * K / 157.3e6
    row = "%-18s MFMA floor %4.1f us |" % ((M, N, K), floor)
    for det in (False, True):
        ops.set_deterministic(det)
        cells = []
        for sk in (1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 60):
            if K // 32 // sk < 2: continue
            os.environ["MVK_GEMM_FORCE"] = "2,1,%d" % sk
            try:
                cells.append("%d:%.1f" % (sk, timeit(lambda: ops.gemm(A, B, out=out))))
            except Exception as e:
                cells.append("%d:ERR" % sk)
        row += (" ordered " if det else " atomic ") + " ".join(cells) + " |"
    ops.set_deterministic(False)
    os.environ.pop("MVK_GEMM_FORCE", None)
    print(row, flush=True)
# launch floor: an empty-ish kernel in the same graph form
x = torch.zeros(64, device=dev)
print("smallest launch (x.add_(1) on 64 floats): %.1f us" % timeit(lambda: x.add_(1.0)))
