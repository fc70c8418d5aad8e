"""Development-only: the rigid KPConv scatter (dx) launches of level 0 / 1 of the synthetic sphere."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops, syn = mvkpconv.sub("ops"), mvkpconv.sub("synthetic")
common = mvkpconv.sub("dropin.datasets.common")
kpmod = mvkpconv.sub("dropin.kernels.kernel_points")
dev = torch.device("cuda:0")
cfg = syn.make_config("early")
staged = syn.stage_spheres([syn.raw_sphere(seed=0)], dev, None)
limits = syn.calibrate_limits(cfg, staged)
p = staged['points'][0] - staged['center'][0]
pyr = common.segmentation_inputs_sphere(cfg, p, np.asarray([p.shape[0]], np.int32), limits, torch.int32)
kp = torch.from_numpy(kpmod.load_kernels(0.1, 15, dimension=3, fixed='center').astype(np.float32)).to(dev)
for lvl, cins in ((0, (66, 32, 64)), (1, (64, 128))):
    pts, nb = pyr['points'][lvl], pyr['neighbors'][lvl]
    for cin in cins:
        dA = torch.randn(pts.shape[0], 15, cin, device=dev)
        scale = 2.0 ** lvl
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ops.kpconv_scatter(pts, pts, nb, dA, kp * scale, 0.048 * scale); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(10): ops.kpconv_scatter(pts, pts, nb, dA, kp * scale, 0.048 * scale)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        print("level %d N %5d H %d Cin %3d : %.1f us (incl. the zero fill of dx)" % (lvl, pts.shape[0], nb.shape[1], cin, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
