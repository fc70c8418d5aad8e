"""Development-only: the gather launch of a first layer with <= 4 channels (baseline: (1, z); middle / late: 4)."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops, syn = mvkpconv.sub("ops"), mvkpconv.sub("synthetic")
common = mvkpconv.sub("dropin.datasets.common")
kpmod = mvkpconv.sub("dropin.kernels.kernel_points")
dev = torch.device("cuda:0")
cfg = syn.make_config("baseline")
staged = syn.stage_spheres([syn.raw_sphere(seed=0)], dev, None)
limits = syn.calibrate_limits(cfg, staged)
p = staged['points'][0] - staged['center'][0]
pyr = common.segmentation_inputs_sphere(cfg, p, np.asarray([p.shape[0]], np.int32), limits, torch.int32)
kp = torch.from_numpy(kpmod.load_kernels(0.1, 15, dimension=3, fixed='center').astype(np.float32)).to(dev)
pts, nb = pyr['points'][0], pyr['neighbors'][0]
for cin in (2, 4, 1, 5, 8):
    x = torch.randn(pts.shape[0], cin, device=dev)
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.kpconv_gather(pts, pts, nb, x, kp, 0.048); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): ops.kpconv_gather(pts, pts, nb, x, kp, 0.048)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("Cin %d  N %d H %d : %.1f us" % (cin, pts.shape[0], nb.shape[1], e0.elapsed_time(e1) / 100 * 1e3), flush=True)
