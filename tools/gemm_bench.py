import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (tools/ sits next to the package)
sys.path.insert(0, ROOT)
import mvkpconv
ops = mvkpconv.sub("ops")
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(19464, 64, 990), (19464, 32, 480), (3986, 64, 960), (923, 128, 1920), (225, 256, 3840), (65, 512, 7680)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev)
    fl = 2.0 * M * N * K
    res = []
    for sk in (1, 2, 3, 4, 6, 8, 16, 32):
        if K // sk < 64: continue
        out = torch.zeros(M, N, device=dev)
        us = timeit(lambda: ops.gemm(A, B, out=out, split_k=sk))
        res.append("sk%d %.0fus %.0fTF" % (sk, us, fl / us / 1e6))
    us = timeit(lambda: torch.matmul(A, B))
    print((M, N, K), " | ".join(res), "| torch.matmul %.0fus %.0fTF" % (us, fl / us / 1e6))
# backward shapes: dA = g @ W^T (NT): M=Nq, N=K*Cin, Kd=Cout ; dW = A^T g (TN): M=K*Cin, N=Cout, Kd=Nq
for (Nq, Cout, KC) in [(19464, 64, 990), (3986, 64, 960), (65, 512, 7680)]:
    g = torch.randn(Nq, Cout, device=dev); W = torch.randn(KC, Cout, device=dev); A = torch.randn(Nq, KC, device=dev)
    fl = 2.0 * Nq * Cout * KC
    us1 = timeit(lambda: ops.gemm(g, W, transB=True))
    res = []
    for sk in (1, 4, 16, 32, 64):
        out = torch.zeros(KC, Cout, device=dev)
        res.append("sk%d %.0fus" % (sk, timeit(lambda: ops.gemm(A, g, transA=True, out=out, split_k=sk))))
    print("bwd", (Nq, Cout, KC), "NT %.0fus %.0fTF" % (us1, fl / us1 / 1e6), "| TN", " ".join(res))
