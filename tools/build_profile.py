"""Development-only: host-side cProfile of the input chain (build_batch) in enqueue-only mode."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
syn, ops = mvkpconv.sub("synthetic"), mvkpconv.sub("ops")
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
cfg = syn.make_config("early")
sph = [syn.raw_sphere(seed=0)]
staged = syn.stage_spheres(sph, dev, [syn.sphere_views(s) for s in sph])
limits = syn.calibrate_limits(cfg, staged)
status = torch.zeros(2, dtype=torch.int32, device=dev)
for _ in range(5):
    syn.build_batch(cfg, staged, limits, torch.int32, status=status)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    syn.build_batch(cfg, staged, limits, torch.int32, status=status)
torch.cuda.synchronize()
print("build_batch %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    syn.build_batch(cfg, staged, limits, torch.int32, status=status)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
