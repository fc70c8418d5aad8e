import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, contextlib
import mvkpconv
syn, ops, dp = mvkpconv.sub("synthetic"), mvkpconv.sub("ops"), mvkpconv.sub("dp")
dev = torch.device("cuda:0")
torch.manual_seed(0); np.random.seed(0)
cfg = syn.make_config("baseline", deformable=True)
sph = [syn.raw_sphere(seed=3, radius=0.6, density=2500.0)]
staged = syn.stage_spheres(sph, dev, None)
limits = syn.calibrate_limits(cfg, staged)
batch, _ = syn.build_batch(cfg, staged, limits, torch.int32)
net = syn.build_model(cfg, dev); net.train()
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith("offset_bias"): p.normal_(0, 0.05)
first_deform = min(i for i, b in enumerate(cfg.architecture) if "deformable" in b)
sd = {k: v.clone() for k, v in net.state_dict().items()}
def grads(cut, scope, retain):
    net.load_state_dict(sd); net.zero_grad(set_to_none=True)
    net.backward_cut = cut
    loss = net.loss(net(batch, cfg), batch.labels)
    if cut is not None:
        dp.two_stage_backward(loss, net.cut_tensors, backward_scope=scope, retain_graph=retain)
    else:
        with (scope or contextlib.nullcontext)():
            loss.backward()
    torch.cuda.synchronize(); net.backward_cut = None
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
def cmp(a, b, tag):
    scale = max(v.abs().max().item() for v in a.values())
    worst = max(((b[n] - a[n]).abs().max().item() / max(a[n].abs().max().item(), 1e-3 * scale), n) for n in a)
    print("%-60s worst %.3e %s" % (tag, worst[0], worst[1]))
    if worst[0] > 1e-4:
        rows = sorted((((b[n] - a[n]).abs().max().item() / max(a[n].abs().max().item(), 1e-3 * scale), n) for n in a), reverse=True)[:6]
        for e, n in rows:
            d = (b[n] - a[n]).abs()
            nbad = int((d > 1e-4 * a[n].abs().max()).sum())
            idx = torch.nonzero(d > 1e-4 * a[n].abs().max())[:4].tolist()
            print("      %.3e %-45s shape %s bad elements %d first %s" % (e, n, tuple(a[n].shape), nbad, idx))
want = grads(None, None, False)
cmp(want, grads(None, None, False), "plain again")
for r in range(6):
    cmp(want, grads(None, None, False), "plain again (%d)" % r)
if os.environ.get("QUICK"):
    sys.exit(0)
cmp(want, grads(None, ops.defer_weight_grads, False), "plain + defer")
cmp(want, grads(first_deform, None, False), "cut below deformables, no scope")
cmp(want, grads(first_deform, ops.defer_weight_grads, False), "cut below deformables, defer")
cmp(want, grads(first_deform + 2, None, True), "cut inside deformables, retain, no scope")
cmp(want, grads(first_deform + 2, ops.defer_weight_grads, True), "cut inside deformables, retain, defer")
