"""Development-only: 3-NN timing on the synthetic early-fusion sphere (pruned vs brute force)."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops, syn = mvkpconv.sub("ops"), mvkpconv.sub("synthetic")
dev = torch.device("cuda:0")
sph = [syn.raw_sphere(seed=0)]
staged = syn.stage_spheres(sph, dev, [syn.sphere_views(s) for s in sph])
pw = staged['points'][0]
xyz, valid = ops.unproject_depth(staged['depth'][0], staged['cam'][0], staged['poses'][0])
print("queries", tuple(pw.shape), "keys", tuple(xyz.shape), "valid", int(valid.sum()))


def run(tag):
    for _ in range(3):
        out = ops.knn_pixels(pw, xyz, valid, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ops.knn_pixels(pw, xyz, valid, 3)
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.0f us" % (tag, e0.elapsed_time(e1) / 20 * 1e3))
    return out


a = run("pruned")
os.environ["MVK_KNN_BRUTE"] = "1"
b = run("brute ")
print("equal", torch.equal(a, b))
