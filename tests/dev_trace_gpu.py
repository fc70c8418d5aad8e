"""Development-only: block-by-block comparison of the product model (GPU) and the CPU port."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
from oracle import torch_port
syn = mvkpconv.sub("synthetic")
dev = torch.device("cuda:0")
variant = sys.argv[1] if len(sys.argv) > 1 else "baseline"
torch.manual_seed(0); np.random.seed(0)
cfg = syn.make_config(variant)
sphere = syn.raw_sphere(seed=0, radius=0.6, density=2500.0)
views = syn.sphere_views(sphere, nv=3, h=60, w=80) if variant != "baseline" else None
staged = syn.stage_spheres([sphere], dev, [views] if views else None)
limits = syn.calibrate_limits(cfg, staged)
batch, lens = syn.build_batch(cfg, staged, limits, torch.int64)
net = syn.build_model(cfg, dev); net.train()
if hasattr(net, "net_2d"):
    for m in net.net_2d._modules.values(): m.train(False)
got = {}
for name, mod in net.named_modules():
    if name.count('.') == 1 and (name.startswith("encoder_blocks") or name.startswith("decoder_blocks")):
        mod.register_forward_hook(lambda m, i, o, name=name: got.__setitem__(name, o.detach().cpu()))
seen = {}
if hasattr(net, "net_2d"):
    net.net_2d.register_forward_hook(lambda m, i, o: seen.__setitem__("feature", o["feature"].detach().cpu()))
if hasattr(net, "feat_aggreg"):
    net.feat_aggreg.register_forward_hook(lambda m, i, o: got.__setitem__("feat_aggreg", (o.detach().cpu(), [t.detach().cpu() for t in i])))
out = net(batch, cfg)
sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
cb = torch_port.batch_to_cpu(batch)
if variant != "baseline":
    cb['feature_2d'] = seen["feature"]
trace = {}
ref, _ = torch_port.forward(sd, cfg, cb, None, True, trace)
if "feat_aggreg" in got:
    o, ins = got.pop("feat_aggreg")
    f = torch_port.lift_2d(sd, cb, None, True)
    print("feat_aggreg out rel err", ((o.permute(0, 2, 1).reshape(-1, 64) - f).abs().max() / f.abs().max()).item())
    # inputs: src_xyz, tgt_xyz, feature
    b_, nv_, _, h_, w_ = cb['images'].shape
    f2d = cb['feature_2d'].reshape(b_, nv_, -1, h_, w_).transpose(1, 2).contiguous().reshape(b_, -1, nv_ * h_ * w_)
    xyz = cb['image_xyz'].permute(0, 4, 1, 2, 3).reshape(b_, 3, nv_ * h_ * w_)
    knn = cb['knn_list'][0].long()
    print("grouped feature equal", torch.equal(ins[2], torch_port.group_points(f2d[0:1], knn)),
          "grouped xyz equal", torch.equal(ins[0], torch_port.group_points(xyz[0:1], knn)),
          "tgt equal", torch.equal(ins[1], cb['feat_aggre_points'].transpose(1, 2)))
    print("src range", ins[0].abs().max().item(), "diff range", (ins[0] - ins[1].unsqueeze(-1)).abs().max().item())
print("points per level", [p.shape[0] for p in batch.points], "limits", limits)
for k in got:
    if k in trace:
        a, b = got[k], trace[k]
        print("%-22s shape %-14s rel err %.3e" % (k, tuple(a.shape), ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()))
print("logits rel err", ((out.detach().cpu() - ref).abs().max() / ref.abs().max()).item())
