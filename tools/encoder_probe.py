"""Development-only: the frozen 2D encoder alone (3 views of 120 x 160), graph-captured: time per call (round 3: 1.14 ms;
with torch.miopen_convolution_relu / _add_relu in place of conv + bias_act_nhwc: 199 ms -- the library's fused plans fall
on naive kernels for these f32 channels-last shapes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
syn = mvkpconv.sub("synthetic")
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg = syn.make_config("early")
torch.manual_seed(0)
net = syn.build_model(cfg, dev)
net.net_2d.eval()
NV = int(sys.argv[1]) if len(sys.argv) > 1 else 3      # views per call (6 = the views of two steps in one call)
x = torch.randn(NV, 3, 120, 160, device=dev)
with torch.no_grad():
    for _ in range(3):
        y = net.net_2d({'image': x})['feature']
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            y = net.net_2d({'image': x})['feature']
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
print("encoder, %d views: %.3f ms per call; checksum %.6f" % (NV, e0.elapsed_time(e1) / 20, float(y.double().abs().mean())))
