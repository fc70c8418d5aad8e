"""Development: where does a split product go wrong? (ordered split reduction)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import mvkpconv
ops = mvkpconv.sub("ops")
torch.manual_seed(0)
for (M, N, K, split) in ((4096, 64, 960, 4), (4096, 64, 960, 2), (256, 64, 960, 4), (4096, 32, 480, 3), (19464, 32, 256, 2)):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda")
    ref = (A.double() @ B.double()).float()
    for rep in range(3):
        y = ops.gemm(A, B, split_k=split)
        torch.cuda.synchronize()
        bad = ((y - ref).abs() > 1e-2 * ref.abs().max()) | ~torch.isfinite(y)
        rows = bad.any(1).nonzero().flatten().cpu().numpy()
        cols = bad.any(0).nonzero().flatten().cpu().numpy()
        print(M, N, K, split, "rep", rep, "bad elements", int(bad.sum()), "rows", rows[:8], "...", rows[-4:] if len(rows) else "", "n rows", len(rows), "cols", cols[:8], "n cols", len(cols))
        if bad.any():
            r = int(rows[0]); c = int(bad[r].nonzero()[0])
            print("   sample", r, c, float(y[r, c]), float(ref[r, c]), "ratio", float(y[r, c] / ref[r, c]))
            # per 32-row tile histogram
            t = (bad.view(-1, 32 if N > 32 else 64, N).any(2).any(1)).nonzero().flatten().cpu().numpy() if M % (32 if N > 32 else 64) == 0 else []
            print("   bad row tiles", list(t[:20]), len(t))
