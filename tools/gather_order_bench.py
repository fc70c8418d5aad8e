"""Development-only: the level-0 gather with a work list (mvk_kpconv_gather_fwd_ordered) -- launch time inside a graph
for the row order, a Morton order, the cell order the neighbour search builds and a random order; 1 and 8 spheres.
  python tools/gather_order_bench.py [cin] [pmc]     pmc: three plain launches per order, for a rocprofv3 --pmc pass
With MVK_GATHER_SPLIT=0 (no sharing workgroups) every order must give the same bits; the tool prints the differing rows."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops, syn = mvkpconv.sub("ops"), mvkpconv.sub("synthetic")
common = mvkpconv.sub("dropin.datasets.common")
kpmod = mvkpconv.sub("dropin.kernels.kernel_points")
dev = torch.device("cuda:0")
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 66
pmc = len(sys.argv) > 2 and sys.argv[2] == "pmc"
cfg = syn.make_config("early")
kp = torch.from_numpy(kpmod.load_kernels(0.1, 15, dimension=3, fixed='center').astype(np.float32)).to(dev)


def part1by2(v):
    v = v.astype(np.uint64) & 0x1fffff
    v = (v | (v << 32)) & 0x1f00000000ffff
    v = (v | (v << 16)) & 0x1f0000ff0000ff
    v = (v | (v << 8)) & 0x100f00f00f00f00f
    v = (v | (v << 4)) & 0x10c30c30c30c30c3
    v = (v | (v << 2)) & 0x1249249249249249
    return v


def orders(pts_np, lens, cell):
    """per-cloud orders, concatenated with the cloud's first row added"""
    out = {"rows": [], "morton": [], "cells": [], "random": []}
    rng = np.random.default_rng(0)
    off = 0
    for n in lens:
        p = pts_np[off:off + n]
        c = np.floor((p - p.min(0)) / cell).astype(np.int64)
        d = c.max(0) + 1
        out["rows"].append(off + np.arange(n))
        out["morton"].append(off + np.argsort(part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2), kind="stable"))
        out["cells"].append(off + np.argsort((c[:, 2] * d[1] + c[:, 1]) * d[0] + c[:, 0], kind="stable"))
        out["random"].append(off + rng.permutation(n))
        off += n
    return {k: torch.from_numpy(np.concatenate(v).astype(np.int32)).to(dev) for k, v in out.items()}


def timed(fn, reps=20):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


for nsph in (1, 8):
    staged = syn.stage_spheres([syn.raw_sphere(seed=i) for i in range(nsph)], dev, None)
    limits = syn.calibrate_limits(cfg, staged)
    ps = [staged['points'][i] - staged['center'][i] for i in range(nsph)]
    lens = np.asarray([p.shape[0] for p in ps], np.int32)
    pyr = common.segmentation_inputs_sphere(cfg, torch.cat(ps, 0), lens, limits, torch.int32)
    pts, nb = pyr['points'][0], pyr['neighbors'][0]
    l0 = [int(v) for v in pyr['lengths'][0].cpu().tolist()]
    x = torch.randn(pts.shape[0], cin, device=dev)
    od = orders(pts.cpu().numpy(), l0, 0.1)
    ref = None
    for name in ("rows", "morton", "cells", "random"):
        o = None if name == "rows" else od[name]
        fn = lambda: ops.kpconv_gather(pts, pts, nb, x, kp, 0.048, order=o)
        if pmc:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            continue
        A = fn()[0]
        torch.cuda.synchronize()
        if ref is None:
            ref = A.clone()
        bad = int((A != ref).any(-1).any(-1).sum())
        print("spheres %d  Nq %6d  H %d  Cin %d  order %-7s : %7.1f us   rows differing from the row order: %d (max %.1e)"
              % (nsph, pts.shape[0], nb.shape[1], cin, name, timed(fn), bad, float((A - ref).abs().max())), flush=True)
