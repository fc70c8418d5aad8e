"""Development-only (GPU box): the rigid KPConv gather launches of the synthetic sphere's pyramid, device time of graph-
captured launches (20 per replay), for the kernel the library picks under the current environment -- run it once with
MVK_GATHER_MFMA=1 (kpconv_gather_mfma, round 5) and once with MVK_GATHER_MFMA=0 (kpconv_gather_vec). With a work list
(cell order) like the network uses. Checks 256 rows per layer against the float64 NumPy restatement.
usage: python tools/gather_mfma_bench.py [tag]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv  # noqa: E402

ops, syn = mvkpconv.sub("ops"), mvkpconv.sub("synthetic")
common = mvkpconv.sub("dropin.datasets.common")
kpmod = mvkpconv.sub("dropin.kernels.kernel_points")
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
cfg = syn.make_config("early")
staged = syn.stage_spheres([syn.raw_sphere(seed=0)], dev, None)
limits = syn.calibrate_limits(cfg, staged)
p = staged['points'][0] - staged['center'][0]
pyr = common.segmentation_inputs_sphere(cfg, p, np.asarray([p.shape[0]], np.int32), limits, torch.int32)
kp = torch.from_numpy(kpmod.load_kernels(0.1, 15, dimension=3, fixed='center').astype(np.float32)).to(dev)


def graph_time(fn, per=20, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
        for _ in range(per):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * per) * 1e3


tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("MVK_GATHER_MFMA", "1")
outs = {}
for lvl, cins in ((0, (66, 32, 64)), (1, (64, 128)), (2, (128, 256)), (3, (256,)), (4, (512,))):
    pts, nb = pyr['points'][lvl], pyr['neighbors'][lvl]
    order = ops.work_order_for(pts)
    for cin in cins:
        torch.manual_seed(cin)
        x = torch.randn(pts.shape[0], cin, device=dev)
        scale = 2.0 ** lvl
        kps = (kp * scale).contiguous()
        us = graph_time(lambda: ops.kpconv_gather(pts, pts, nb, x, kps, 0.048 * scale, order=order))
        A = ops.kpconv_gather(pts, pts, nb, x, kps, 0.048 * scale, order=order)[0]
        outs[(lvl, cin)] = A.cpu()
        heff = float((nb < pts.shape[0]).sum(1).float().mean())
        by = pts.shape[0] * heff * (cin * 4 + 16) + pts.shape[0] * 12 + pts.shape[0] * 15 * cin * 4
        print("[%s] level %d  N %5d  H %2d (eff %.1f)  Cin %3d : %6.1f us  %.2f TB/s algorithmic  (work list: %s)" % (
            tag, lvl, pts.shape[0], nb.shape[1], heff, cin, us, by / us / 1e6, order is not None), flush=True)
# correctness of the kernel under test on the real pyramid: 256 random rows per level against the float64 NumPy restatement
from oracle import npref  # noqa: E402
rng = np.random.default_rng(0)
for (lvl, cin), A in outs.items():
    pts, nb = pyr['points'][lvl].cpu().numpy(), pyr['neighbors'][lvl].cpu().numpy()
    rows = np.unique(rng.integers(0, pts.shape[0], 256))
    torch.manual_seed(cin)
    x = torch.randn(pts.shape[0], cin, device=dev).cpu().numpy()
    scale = 2.0 ** lvl
    _, want, _ = npref.kpconv_forward(pts[rows].astype(np.float64), pts.astype(np.float64), nb[rows].astype(np.int64),
                                      x.astype(np.float64), (kp.cpu().numpy() * scale).astype(np.float64),
                                      np.zeros((15, cin, 1)), 0.048 * scale, return_A=True)
    err = np.abs(A.numpy()[rows] - want).max() / np.abs(want).max()
    print("[%s] level %d Cin %3d: 256 rows vs float64 oracle %.2e" % (tag, lvl, cin, err))
