"""Development-only: level sizes (points, neighbour columns) of a synthetic workload."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
syn = mvkpconv.sub("synthetic")
wl = sys.argv[1] if len(sys.argv) > 1 else "middle"
deform = len(sys.argv) > 2 and sys.argv[2] == "deformable"
dev = torch.device("cuda", 0)
cfg = syn.make_config(wl, deformable=deform, modulated=False)
spheres = [syn.raw_sphere(seed=1000, radius=2.0)]
views = [syn.sphere_views(s, nv=5) for s in spheres] if wl != "baseline" else None
staged = syn.stage_spheres(spheres, dev, views)
limits = syn.calibrate_limits(cfg, staged)
batch, lens = syn.build_batch(cfg, staged, limits, torch.int32)
print("limits", limits)
for name in ("points", "neighbors", "pools", "upsamples"):
    v = getattr(batch, name, None)
    if v is not None:
        print(name, [tuple(t.shape) for t in v])
for i, nb in enumerate(batch.neighbors):
    n = nb.shape[0]
    print("level", i, "mean real neighbours %.1f" % ((nb < n).sum(1).float().mean().item()))
print(cfg.architecture)
