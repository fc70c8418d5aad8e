"""Development-only (GPU box): mean fill of the reverse lists a batch carries against the mean fill of the forward neighbour
matrices (they hold the same pairs), for the eager pyramid (syn.build_batch) and the sync-free input chain."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops, syn = mvkpconv.sub("ops"), mvkpconv.sub("synthetic")
dev = torch.device("cuda:0")
cfg = syn.make_config("early")
staged = syn.stage_spheres([syn.raw_sphere(seed=0)], dev, [syn.sphere_views(syn.raw_sphere(seed=0), nv=3, h=120, w=160)])
limits = syn.calibrate_limits(cfg, staged)
b, _ = syn.build_batch(cfg, staged, limits, torch.int32)
for l in range(len(b.points)):
    n = b.points[l].shape[0]
    nb, rv = b.neighbors[l], b.rev_neighbors[l]
    if rv is None:
        continue
    print("level", l, "N", n, "fwd width", nb.shape[1], "fwd mean", float((nb < n).sum()) / n, "| rev width", rv.shape[1],
          "rev mean (< N)", float((rv < n).sum()) / n, "rev min/max value", int(rv.min()), int(rv.max()))
