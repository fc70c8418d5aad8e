"""Development-only: the KPConv gather launches of level 0 / 1 of the synthetic sphere, timed with HIP events."""
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


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for lvl, cins in ((0, (66, 32, 64)), (1, (64, 128))):
    pts, nb = pyr['points'][lvl], pyr['neighbors'][lvl]
    for cin in cins:
        x = torch.randn(pts.shape[0], cin, device=dev)
        scale = 2.0 ** lvl
        us = t(lambda: ops.kpconv_gather(pts, pts, nb, x, kp * scale, 0.048 * scale))
        heff = float((nb < pts.shape[0]).sum(1).float().mean())
        by = pts.shape[0] * heff * (cin * 4 + 16) + pts.shape[0] * 12 + pts.shape[0] * 15 * cin * 4
        print("level %d  N %5d  H %2d (eff %.1f)  Cin %3d : %6.1f us  %.2f TB/s algorithmic" % (
            lvl, pts.shape[0], nb.shape[1], heff, cin, us, by / us / 1e6))

A = torch.randn(19464, 990, device=dev); W = torch.randn(990, 64, device=dev)
for name, fn in (("f32 MFMA 19464x990x64", lambda: ops.gemm(A, W)),):
    us = t(fn)
    print("%s: %.1f us  %.1f TFLOP/s" % (name, us, 2 * 19464 * 990 * 64 / us / 1e6))
