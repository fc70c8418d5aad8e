"""Development-only: how close do the level sizes get to the captured capacities over many random grid orientations?"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops, syn = mvkpconv.sub("ops"), mvkpconv.sub("synthetic")
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
variant = sys.argv[1] if len(sys.argv) > 1 else "early"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cfg = syn.make_config(variant)
sph = [syn.raw_sphere(seed=0)]
staged = syn.stage_spheres(sph, dev, [syn.sphere_views(s) for s in sph] if variant != "baseline" else None)
limits = syn.calibrate_limits(cfg, staged)
np.random.seed(1234)
batch, _ = syn.build_batch(cfg, staged, limits, torch.int32)
static = syn.StaticBatch(batch, limits)
chain = syn.DeviceInputChain(cfg, staged, limits, static)
first = [int(p.shape[0]) for p in batch.points]
mx = np.zeros(len(static.caps), np.int64)
mn = np.full(len(static.caps), 1 << 30, np.int64)
for it in range(n):
    chain.draw_rotations()
    chain.build(static)
    c = np.array([int(x) for x in torch.cat(static._counts).cpu()])
    mx, mn = np.maximum(mx, c), np.minimum(mn, c)
print("first batch sizes", first)
print("capacities      ", static.caps)
print("min over %d     " % n, mn.tolist())
print("max over %d     " % n, mx.tolist(), "-> headroom", [round(float(c) / m, 3) for c, m in zip(static.caps, mx)])
print("status", chain.status.cpu().tolist())
