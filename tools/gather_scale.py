"""Development-only: level-0 gather time against the number of query points (are launches bound by rounds of waves?)."""
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
pts, nb = pyr['points'][0], pyr['neighbors'][0]
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(pts.shape[0], cin, device=dev)
for nq in (2048, 4096, 8192, 12288, 16000, 16384, 16800, 18000, 19464):
    q, idx = pts[:nq].contiguous(), nb[:nq].contiguous()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.kpconv_gather(q, pts, idx, x, kp, 0.048)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                ops.kpconv_gather(q, pts, idx, x, kp, 0.048)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print("Cin %d  Nq %5d : %6.1f us" % (cin, nq, e0.elapsed_time(e1) / 100 * 1e3), flush=True)
