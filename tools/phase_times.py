"""Development-only (GPU box): device time of the two kinds of captured steps (with / without the two-batch encoder call),
HIP events around every replay.  usage: python tools/phase_times.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops, syn, stepmod, optim = mvkpconv.sub("ops"), mvkpconv.sub("synthetic"), mvkpconv.sub("step"), mvkpconv.sub("optim")
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
cfg = syn.make_config("early")
sph = [syn.raw_sphere(seed=0)]
staged = syn.stage_spheres(sph, dev, [syn.sphere_views(s, nv=3, h=120, w=160) for s in sph])
limits = syn.calibrate_limits(cfg, staged)
net = syn.build_model(cfg, dev)
net.train()
for m in net.net_2d._modules.values():
    m.train(False)
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=1e-3)
batch, _ = syn.build_batch(cfg, staged, limits, torch.int32)
stepmod.net_step_captured(net, batch, cfg, params, opt, None)
step = stepmod.GraphStep(net, cfg, opt, staged, limits)
for _ in range(6):
    step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    step()
    ev[i + 1].record()
torch.cuda.synchronize()
t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(40)])
print("even steps %.3f ms, odd steps %.3f ms, mean %.3f" % (t[0::2].mean(), t[1::2].mean(), t.mean()))
