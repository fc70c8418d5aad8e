"""Development-only: device time of the grouped weight-gradient launch of one backward pass (the real product list of a
workload, recorded from one step) under the split policy selected by MVK_DW_GROUP_KTILES.
usage: python tools/dw_group_bench.py [early|baseline|...] [spheres]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
syn, ops = mvkpconv.sub("synthetic"), mvkpconv.sub("ops")
dev = torch.device("cuda:0")
variant = sys.argv[1] if len(sys.argv) > 1 else "early"
nsph = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0); np.random.seed(0)
cfg = syn.make_config(variant)
sph = [syn.raw_sphere(seed=i) for i in range(nsph)]
views = [syn.sphere_views(s) for s in sph] if variant != "baseline" else None
staged = syn.stage_spheres(sph, dev, views)
limits = syn.calibrate_limits(cfg, staged)
net = syn.build_model(cfg, dev); net.train()
if hasattr(net, "net_2d"):
    for m in net.net_2d._modules.values(): m.train(False)
batch, lens = syn.build_batch(cfg, staged, limits, torch.int32)
shapes = []
orig = ops._flush_deferred


def spy(items=None):
    for A, B, *_ in ops._DEFER["items"]:
        shapes.append((A.shape[1], B.shape[1], A.shape[0]))
    orig()


ops._flush_deferred = spy
loss = net.loss(net(batch, cfg), batch.labels)
with ops.defer_weight_grads():
    loss.backward()
torch.cuda.synchronize()
ops._flush_deferred = orig
del loss, net, batch
torch.cuda.empty_cache()
flops = sum(2.0 * m * n * k for m, n, k in shapes)
abytes = sum(4.0 * (m + n) * k for m, n, k in shapes)
cbytes = sum(4.0 * m * n for m, n, k in shapes)
L = ops.lib()
sp = [L.mvk_gemm_f32_tn_grouped_split(m, n, k) for m, n, k in shapes]
atom = sum(4.0 * m * n * s for (m, n, k), s in zip(shapes, sp) if s > 1)
print("%d products, %.2f GFLOP, operands %.0f MB, outputs %.0f MB; split bounds: max %d, atomic bytes <= %.0f MB, policy %s"
      % (len(shapes), flops / 1e9, abytes / 1e6, cbytes / 1e6, max(sp), atom / 1e6, os.environ.get("MVK_DW_GROUP_KTILES", "by group size")))
As = [torch.randn(k, m, device=dev) for m, n, k in shapes]
Bs = [torch.randn(k, n, device=dev) for m, n, k in shapes]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for it in range(12):
    ops.step_begin() if hasattr(ops, "step_begin") else None
    torch.cuda.synchronize()
    with ops.defer_weight_grads():
        outs = [ops._dw_gemm(a, b) for a, b in zip(As, Bs)]
        torch.cuda.synchronize()
        e0.record()
    e1.record()
    torch.cuda.synchronize()
    if it >= 2:
        ts.append(e0.elapsed_time(e1) * 1e3)
ref = As[0].t() @ Bs[0]
err = (outs[0] - ref).abs().max().item() / ref.abs().max().item()
print("grouped launch (table copy + kernels): median %.1f us, min %.1f us | %.1f TFLOP/s | rel err of product 0: %.1e"
      % (float(np.median(ts)), min(ts), flops / float(np.median(ts)) / 1e6, err))
