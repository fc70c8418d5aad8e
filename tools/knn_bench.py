import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (tools/ sits next to the package); sys.path.insert(0, ROOT)
import mvkpconv
ops = mvkpconv.sub("ops")
dev = torch.device("cuda:0")
torch.manual_seed(0)
q = torch.rand(19464, 3, device=dev); keys = torch.rand(3, 120, 160, 3, device=dev, dtype=torch.float64); mask = torch.rand(3, 120, 160, device=dev) > 0.05
for _ in range(3): out = ops.knn_pixels(q, keys, mask, 3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): out = ops.knn_pixels(q, keys, mask, 3)
e1.record(); torch.cuda.synchronize()
print("knn 19464 x 57600: %.0f us" % (e0.elapsed_time(e1) / 10 * 1e3))
# check vs torch brute force on a subset
kk = keys.reshape(-1, 3)[mask.reshape(-1)]; ind = torch.nonzero(mask.reshape(-1))[:, 0]
d = ((q[:500, None, :].double() - kk[None]) ** 2).sum(-1)
ref = ind[d.topk(3, largest=False).indices]
print("match", torch.equal(ref, out[:500]))
