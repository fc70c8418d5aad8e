"""Development-only: the forward K x Cin x Cout contractions of the early-fusion net, for an MFMA-busy PMC pass."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops = mvkpconv.sub("ops")
dev = torch.device("cuda:0")
for (M, N, K) in [(19464, 64, 990), (19464, 32, 480), (3986, 64, 960), (923, 128, 1920), (225, 256, 3840), (65, 512, 7680)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev)
    for _ in range(10):
        ops.gemm(A, B)
torch.cuda.synchronize()
