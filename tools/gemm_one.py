"""Development-only: one GEMM shape, a few launches (for rocprofv3 --pmc passes). usage: gemm_one.py M N K ta tb"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops = mvkpconv.sub("ops")
M, N, K, ta, tb = (int(v) for v in sys.argv[1:6])
A = torch.randn((K, M) if ta else (M, K), device="cuda"); B = torch.randn((N, K) if tb else (K, N), device="cuda")
out = torch.zeros(M, N, device="cuda")
for _ in range(10):
    ops.gemm(A, B, transA=bool(ta), transB=bool(tb), out=out)
torch.cuda.synchronize()
