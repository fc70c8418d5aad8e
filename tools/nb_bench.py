"""Development-only: timing of the level-0 neighbour searches of the synthetic sphere."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops, syn = mvkpconv.sub("ops"), mvkpconv.sub("synthetic")
dev = torch.device("cuda:0")
staged = syn.stage_spheres([syn.raw_sphere(seed=0)], dev, None)
p = staged['points'][0] - staged['center'][0]
l = [p.shape[0]]
sub, ls = ops.grid_subsample_batch(p, l, dl=0.08)
status = torch.zeros(2, dtype=torch.int32, device=dev)


def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("N0 %d N1 %d" % (p.shape[0], sub.shape[0]))
print("conv  L0 (19k x 19k, r .1, limit 58) build+query %.0f us" % t(lambda: ops.radius_neighbors_batch(p, p, l, l, 0.1, limit=58, status=status)))
print("conv  L0 query only (grid reused)                %.0f us" % t(lambda: ops.radius_neighbors_batch(p, p, l, l, 0.1, limit=58, status=status, reuse_grid=True)))
print("pool  L0 (4k x 19k, r .1)  query only            %.0f us" % t(lambda: ops.radius_neighbors_batch(sub, p, ls, l, 0.1, limit=58, status=status, reuse_grid=True)))
ops.radius_neighbors_batch(sub, sub, ls, ls, 0.2, limit=53, status=status)
print("up    L0 (19k x 4k, r .2)  query only            %.0f us" % t(lambda: ops.radius_neighbors_batch(p, sub, l, ls, 0.2, limit=53, status=status, reuse_grid=True)))
print("conv  L1 (4k x 4k, r .2)   query only            %.0f us" % t(lambda: ops.radius_neighbors_batch(sub, sub, ls, ls, 0.2, limit=53, status=status, reuse_grid=True)))
print("sub   L0->L1                                     %.0f us" % t(lambda: ops.grid_subsample_batch(p, l, dl=0.08)))
