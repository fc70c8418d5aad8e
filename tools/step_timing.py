"""Development-only: wall-clock split of one bench step (with syncs between stages)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (tools/ sits next to the package)
sys.path.insert(0, ROOT)
import mvkpconv
syn, ops = mvkpconv.sub("synthetic"), mvkpconv.sub("ops")
dev = torch.device("cuda:0")
variant = sys.argv[1] if len(sys.argv) > 1 else "early"
torch.manual_seed(0); np.random.seed(0)
cfg = syn.make_config(variant)
sph = [syn.raw_sphere(seed=0)]
views = [syn.sphere_views(s) for s in sph] if variant != "baseline" else None
staged = syn.stage_spheres(sph, dev, views)
limits = syn.calibrate_limits(cfg, staged)
net = syn.build_model(cfg, dev); net.train()
if hasattr(net, "net_2d"):
    for m in net.net_2d._modules.values(): m.train(False)
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-2, momentum=0.98, weight_decay=1e-3)
def S(): torch.cuda.synchronize(); return time.perf_counter()
acc = {}
def add(k, v): acc[k] = acc.get(k, 0.0) + v
from mvkpconv import sub
common = sub("dropin.datasets.common")
for it in range(13):
    t0 = S()
    batch, lens = syn.build_batch(cfg, staged, limits, torch.int32); t1 = S()
    opt.zero_grad(set_to_none=True)
    out = net(batch, cfg); t2 = S()
    loss = net.loss(out, batch.labels); t3 = S()
    loss.backward(); t4 = S()
    torch.nn.utils.clip_grad_value_(params, 100.0); opt.step(); t5 = S()
    if it >= 3:
        add("batch", t1 - t0); add("fwd", t2 - t1); add("loss", t3 - t2); add("bwd", t4 - t3); add("opt", t5 - t4)
print({k: round(v / 10 * 1e3, 2) for k, v in acc.items()}, "ms; total", round(sum(acc.values()) / 10 * 1e3, 2))
# finer: pyramid only vs fusion inputs
t0 = S()
for _ in range(10):
    pts = [p - c for p, c in zip(staged['points'], staged['center'])]
    pyr = common.segmentation_inputs_sphere(cfg, torch.cat(pts, 0), np.asarray([pts[0].shape[0]], np.int32), limits, torch.int32)
t1 = S(); print("pyramid only ms", (t1 - t0) / 10 * 1e3)
if variant != "baseline":
    t0 = S()
    for _ in range(10):
        xyz, valid = ops.unproject_depth(staged['depth'][0], staged['cam'][0], staged['poses'][0])
        knn = ops.knn_pixels(staged['points'][0], xyz, valid, k=3)
    t1 = S(); print("unproject+knn ms", (t1 - t0) / 10 * 1e3)
    with torch.no_grad():
        t0 = S()
        for _ in range(10):
            f = net.net_2d({'image': batch.images.reshape(-1, 3, 120, 160)})['feature']
        t1 = S(); print("unet fwd ms", (t1 - t0) / 10 * 1e3)
# python profile of fwd+bwd
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    batch, lens = syn.build_batch(cfg, staged, limits, torch.int32)
    opt.zero_grad(set_to_none=True); out = net(batch, cfg); loss = net.loss(out, batch.labels); loss.backward(); opt.step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).strip_dirs().sort_stats("tottime").print_stats(45)
