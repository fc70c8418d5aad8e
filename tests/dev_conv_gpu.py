import torch, torch.nn.functional as F
torch.manual_seed(0)
x = torch.randn(1, 68, 2311, 3); w = torch.randn(64, 68, 1, 1) * 0.1
ref = F.conv2d(x.double(), w.double())
def err(y): return ((y.double().cpu() - ref).abs().max() / ref.abs().max()).item()
print("cpu f32", err(F.conv2d(x, w)))
xg, wg = x.cuda(), w.cuda()
for flag in (True, False):
    torch.backends.cudnn.allow_tf32 = flag
    print("gpu conv2d allow_tf32=%s" % flag, err(F.conv2d(xg, wg)), err(F.conv2d(xg, wg)))
torch.backends.cudnn.allow_tf32 = True
print("gpu conv as matmul", err(torch.einsum('oc,bcnk->bonk', wg[:, :, 0, 0], xg)))
x3 = torch.randn(3, 64, 120, 160); w3 = torch.randn(64, 64, 3, 3) * 0.05
r3 = F.conv2d(x3.double(), w3.double(), padding=1)
for flag in (True, False):
    torch.backends.cudnn.allow_tf32 = flag
    y = F.conv2d(x3.cuda(), w3.cuda(), padding=1)
    print("gpu conv3x3 allow_tf32=%s" % flag, ((y.double().cpu() - r3).abs().max() / r3.abs().max()).item())
print(torch.backends.cudnn.version() if torch.backends.cudnn.is_available() else None)
