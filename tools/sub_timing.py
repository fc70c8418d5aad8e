"""Development: phase times of subsample_cloud_kernel (MVK_SUB_TIMING=1) for the four pyramid levels of a 19 k-point sphere."""
import os, sys
os.environ["MVK_SUB_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import mvkpconv
syn, ops = mvkpconv.sub("synthetic"), mvkpconv.sub("ops")
dev = torch.device("cuda:0")
sph = syn.raw_sphere(seed=0)
st = syn.stage_spheres([sph], dev, None)
p = st['points'][0] - st['center'][0]
dl = 0.08
for rep in range(2):
    q = p
    for lvl in range(4):
        out = ops.grid_subsample_batch(q, [q.shape[0]], dl=dl * (2 ** lvl))
        q = out[0]
    torch.cuda.synchronize()
