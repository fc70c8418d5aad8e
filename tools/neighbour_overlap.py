"""Development-only: how much the neighbour sets of G consecutive points of a level overlap (rows of dx a workgroup
of the scatter could combine in LDS before going to global atomics)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
syn = mvkpconv.sub("synthetic")
wl = sys.argv[1] if len(sys.argv) > 1 else "early"
dev = torch.device("cuda", 0)
cfg = syn.make_config(wl, deformable=False, modulated=False)
spheres = [syn.raw_sphere(seed=1000, radius=2.0)]
views = [syn.sphere_views(s, nv=3) for s in spheres] if wl != "baseline" else None
staged = syn.stage_spheres(spheres, dev, views)
limits = syn.calibrate_limits(cfg, staged)
batch, lens = syn.build_batch(cfg, staged, limits, torch.int32)
for name, mats in (("neighbors", batch.neighbors), ("pools", batch.pools[:-1]), ("upsamples", batch.upsamples[:-1])):
    for l, nb in enumerate(mats):
        ns = int(nb.max().item())           # shadow index = number of support rows
        for G in (4, 16, 64):
            n = (nb.shape[0] // G) * G
            if n == 0:
                continue
            g = nb[:n].reshape(n // G, -1).long()
            real = (g < ns).sum().item()
            s, _ = torch.sort(g, 1)
            uniq = ((s[:, 1:] != s[:, :-1]) & (s[:, 1:] < ns)).sum().item() + (s[:, 0] < ns).sum().item()
            print("%-9s level %d  rows %6d x %3d  G=%2d  entries/group %.0f  distinct rows/group %.0f  ratio %.2f"
                  % (name, l, nb.shape[0], nb.shape[1], G, real / (n // G), uniq / (n // G), real / max(uniq, 1)))
