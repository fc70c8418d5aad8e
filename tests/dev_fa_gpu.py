import os, sys, numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mvkpconv
fa_mod = mvkpconv.sub("dropin.mvpnet.models.mvpnet_3d")
torch.manual_seed(0)
fa = fa_mod.FeatureAggregation(64)
fa.train()
np_, k = 2311, 3
src = torch.rand(1, 3, np_, k) * 1.2; tgt = src.mean(3) + 0.01 * torch.randn(1, 3, np_); feat = torch.randn(1, 64, np_, k) * 0.1 + 0.07
import copy
fg = copy.deepcopy(fa).cuda()
rec_c, rec_g = {}, {}
for n, m in fa.named_modules():
    if n.endswith("conv") or n.endswith("bn"): m.register_forward_hook(lambda m, i, o, n=n: rec_c.__setitem__(n, o.detach().clone()))
for n, m in fg.named_modules():
    if n.endswith("conv") or n.endswith("bn"): m.register_forward_hook(lambda m, i, o, n=n: rec_g.__setitem__(n, o.detach().cpu().clone()))
oc = fa(src, tgt, feat); og = fg(src.cuda(), tgt.cuda(), feat.cuda())
for n in rec_c:
    print(n, ((rec_c[n] - rec_g[n]).abs().max() / rec_c[n].abs().max()).item(), "absmax", rec_c[n].abs().max().item())
print("out", ((oc - og.cpu()).abs().max() / oc.abs().max()).item())
# BN alone, 4-D vs 2-D layouts
x = rec_c["mlp.0.conv"]
bn = torch.nn.BatchNorm2d(64); bn.train(); bg = copy.deepcopy(bn).cuda()
print("bn2d alone", ((bn(x) - bg(x.cuda()).cpu()).abs().max()).item())
torch.backends.cudnn.enabled = False
print("bn2d miopen off", ((bn(x) - bg(x.cuda()).cpu()).abs().max()).item())
