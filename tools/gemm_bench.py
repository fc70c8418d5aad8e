"""Development-only: gemm_f32_mfma over the product shapes of the early-fusion net (plan chosen by the library,
and forced plans via MVK_GEMM_FORCE=pm,qn,split), against torch.matmul."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (tools/ sits next to the package)
sys.path.insert(0, ROOT)
import mvkpconv
ops = mvkpconv.sub("ops")
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    """us per launch, device side: n launches captured in one hipGraph (the Python -> ctypes -> hipLaunchKernel path
    costs ~10 us per call, more than most of these kernels)."""
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
sweep = "--sweep" in sys.argv
# (M, N, K, transA, transB): forward contractions / unary layers (NN, NT), dA (NT), dW (TN)
shapes = [(19464, 64, 990, 0, 0), (19464, 32, 480, 0, 0), (4986, 64, 960, 0, 0), (1300, 128, 1920, 0, 0), (330, 256, 3840, 0, 0), (85, 512, 7680, 0, 0),
          (19464, 128, 64, 0, 1), (19464, 32, 128, 0, 1), (19464, 128, 32, 0, 1), (4986, 256, 128, 0, 1), (19464, 128, 384, 0, 1), (19464, 20, 128, 0, 1),
          (19464, 990, 64, 0, 1), (4986, 960, 64, 0, 1), (85, 7680, 512, 0, 1),
          (990, 64, 19464, 1, 0), (960, 64, 4986, 1, 0), (7680, 512, 85, 1, 0), (128, 64, 19464, 1, 0), (32, 128, 19464, 1, 0)]
for (M, N, K, ta, tb) in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    fl = 2.0 * M * N * K
    os.environ.pop("MVK_GEMM_FORCE", None)
    sp, rows = ops.gemm_plan(M, N, K, None, False)
    out = torch.zeros(M, N, device=dev)
    us = timeit(lambda: ops.gemm(A, B, transA=bool(ta), transB=bool(tb), out=out))
    ust = timeit(lambda: torch.matmul(A.t() if ta else A, B.t() if tb else B))
    line = "%-26s plan split %2d | %6.1f us %5.1f TF | torch %6.1f us" % ((M, N, K, "TN"[1 - ta] + "TN"[1 - tb]), sp, us, fl / us / 1e6, ust)
    if sweep:
        best = []
        for pm in (1, 2, 3, 4, 5, 6, 8):
            for qn in ((1, 2) if N > 64 else (1,)):
                if pm * qn > 12 or (N <= 32 and pm > 2): continue
                for sk in (1, 2, 3, 4, 6, 8, 12, 16, 32):
                    if K // 32 // sk < 2 and sk > 1: continue
                    os.environ["MVK_GEMM_FORCE"] = "%d,%d,%d" % (pm, qn, sk)
                    best.append((timeit(lambda: ops.gemm(A, B, transA=bool(ta), transB=bool(tb), out=out)), pm, qn, sk))
        best.sort()
        line += " | best " + "  ".join("%.1fus(pm%d qn%d sk%d)" % b for b in best[:4])
    print(line, flush=True)
