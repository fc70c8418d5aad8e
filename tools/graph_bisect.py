"""Development-only: which piece breaks hipGraph capture? Each case runs in its own process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (tools/ sits next to the package)
CASE = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import mvkpconv
syn, ops = mvkpconv.sub("synthetic"), mvkpconv.sub("ops")
dev = torch.device("cuda:0")
case = sys.argv[1]
torch.manual_seed(0); np.random.seed(0)
def capture(fn, warm=2):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): out = fn()
    g.replay(); torch.cuda.synchronize()
    return out
if case == "gemm":
    A = torch.randn(500, 960, device=dev); B = torch.randn(960, 64, device=dev)
    print(case, capture(lambda: ops.gemm(A, B)).sum().item())
elif case == "gemm_split":
    A = torch.randn(50, 7680, device=dev); B = torch.randn(7680, 64, device=dev)
    print(case, capture(lambda: ops.gemm(A, B)).sum().item())
elif case == "bn":
    x = torch.randn(1000, 64, device=dev, requires_grad=True); bn = torch.nn.BatchNorm1d(64).to(dev)
    nv = torch.tensor([900], dtype=torch.int32, device=dev)
    def f():
        y = ops.bn_lrelu(x, nv, bn, 0.1); y.sum().backward(); return y
    print(case, capture(f).sum().item())
elif case in ("kpconv_fwd", "kpconv_fwdbwd"):
    q = torch.rand(2000, 3, device=dev) * 0.5; idx = torch.randint(0, 2001, (2000, 30), device=dev, dtype=torch.int32)
    x = torch.randn(2000, 64, device=dev, requires_grad=True); kp = torch.randn(15, 3, device=dev) * 0.05
    W = torch.randn(15, 64, 64, device=dev, requires_grad=True)
    def f():
        y, _ = ops.kpconv(q, q, idx, x, kp, W, 0.06)
        if case == "kpconv_fwdbwd": y.sum().backward()
        return y
    print(case, capture(f).sum().item())
elif case == "pools":
    x = torch.randn(2000, 64, device=dev, requires_grad=True); idx = torch.randint(0, 2001, (500, 30), device=dev, dtype=torch.int32)
    def f():
        y = ops.max_pool(x, idx) + ops.closest_pool(x, idx); y.sum().backward(); return y
    print(case, capture(f).sum().item())
else:
    variant, what = case.split(":")
    cfg = syn.make_config(variant)
    import os
    big = os.environ.get("BIG") == "1"
    sph = [syn.raw_sphere(seed=0)] if big else [syn.raw_sphere(seed=0, radius=0.7, density=2500.0)]
    views = ([syn.sphere_views(s) for s in sph] if big else [syn.sphere_views(s, nv=3, h=60, w=80) for s in sph]) if variant != "baseline" else None
    staged = syn.stage_spheres(sph, dev, views); limits = syn.calibrate_limits(cfg, staged)
    batch, lens = syn.build_batch(cfg, staged, limits, torch.int32)
    net = syn.build_model(cfg, dev); net.train()
    if hasattr(net, "net_2d"):
        for m in net.net_2d._modules.values(): m.train(False)
    params = [p for p in net.parameters() if p.requires_grad]
    if os.environ.get("GROUPS") == "1":
        opt = torch.optim.SGD([{"params": params}, {"params": [], "lr": 1e-3}], lr=1e-2, momentum=0.98, weight_decay=1e-3)
    else:
        opt = torch.optim.SGD(params, lr=1e-2, momentum=0.98, weight_decay=1e-3)
    if os.environ.get("EAGER_FIRST") == "1":
        for _ in range(2):
            b2, _ = syn.build_batch(cfg, staged, limits, torch.int32)
            opt.zero_grad(set_to_none=True); l = net.loss(net(b2, cfg), b2.labels); l.backward(); opt.step()
    static = syn.StaticBatch(batch, limits); ops.set_row_counts(static.valid)
    def f():
        if what == "unet":
            b, nv, _, h, w = static.images.shape
            return net.net_2d({"image": static.images.reshape(-1, 3, h, w)})["feature"]
        out = net(static, cfg)
        if what == "fwd": return out
        loss = net.loss(out, static.labels)
        if what == "loss": return loss
        loss.backward()
        if what == "bwd": return loss
        torch.nn.utils.clip_grad_value_(params, 100.0)
        if what == "clip": return loss
        opt.step(); return loss
    if what in ("bwd", "clip", "step"): opt.zero_grad(set_to_none=True)
    print(case, capture(f).sum().item())
''' % ROOT
cases = sys.argv[1:] or ["gemm", "gemm_split", "bn", "kpconv_fwd", "kpconv_fwdbwd", "pools", "baseline:fwd", "baseline:loss",
                         "baseline:bwd", "baseline:clip", "baseline:step", "early:unet", "early:fwd", "early:step"]
for c in cases:
    r = subprocess.run([sys.executable, "-c", CASE, c], capture_output=True, text=True, timeout=280)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.strip().splitlines() if "Error" in l or "error" in l][-2:]
    print("%-16s rc=%4d %s %s" % (c, r.returncode, tail[:80], " | ".join(e[:160] for e in err)))
