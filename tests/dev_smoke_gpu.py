"""Development-only GPU sanity script (not collected by pytest): first contact of the KPConv
gather + MFMA GEMM kernels with the golden fixtures before the rest of the library exists."""
import ctypes, importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("enhancing-3d-point-cloud-segmentation-using-multi-modal-fusion-with-2d-images_amd")
L = pkg._lib
raw = ctypes.CDLL(L.LIB_PATH)
L._SIGNATURES = {k: v for k, v in L._SIGNATURES.items() if hasattr(raw, k)}
ops = importlib.import_module(pkg.__name__ + ".ops")
from util import rel_err
dev = torch.device("cuda:0")
def T(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

# GEMM checks (asymmetric operands)
torch.manual_seed(0)
for (M, N, K) in [(64, 64, 16), (100, 45, 990), (4096, 64, 960), (33, 70, 7)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev)
    ref = (A.double() @ B.double())
    print("gemm NN", M, N, K, rel_err(ops.gemm(A, B).cpu().numpy(), ref.cpu().numpy()))
    print("gemm NT", rel_err(ops.gemm(A, B.t().contiguous(), transB=True).cpu().numpy(), ref.cpu().numpy()))
    print("gemm TN split", rel_err(ops.gemm(A.t().contiguous(), B, transA=True, split_k=3).cpu().numpy(), ref.cpu().numpy()))

for name, infl, agg in [("g4_kpconv_config1", "linear", "sum"), ("g4_kpconv_cin66", "linear", "sum"),
                        ("g4_kpconv_cin2", "linear", "sum"), ("g4_kpconv_gaussian", "gaussian", "sum"),
                        ("g4_kpconv_constant", "constant", "sum"), ("g4_kpconv_closest", "linear", "closest"),
                        ("g4_kpconv_strided", "linear", "sum")]:
    g = dict(np.load(os.path.join(ROOT, "tests/golden", name + ".npz")))
    for idt in (torch.int32, torch.int64):
        x = T(g["x"]).requires_grad_(True); W = T(g["weights"]).requires_grad_(True)
        y, _ = ops.kpconv(T(g["q"]), T(g["s"]), T(g["idx"]).to(idt), x, T(g["kernel_points"]), W, float(g["extent"]), infl, agg)
        (y * T(g["g"])).sum().backward()
        print(name, idt, "y", rel_err(y.detach().cpu().numpy(), g["y"]), "dx", rel_err(x.grad.cpu().numpy(), g["x_grad"]),
              "dW", rel_err(W.grad.cpu().numpy(), g["weights_grad"]))
# timing config-1-like at 20k points
g = dict(np.load(os.path.join(ROOT, "tests/golden/g4_kpconv_config1.npz")))
q, s, idx = T(g["q"]), T(g["s"]), T(g["idx"]); x = T(g["x"]); kp = T(g["kernel_points"]); W = T(g["weights"])
for _ in range(3): ops.kpconv_gather(q, s, idx, x, kp, 0.048)
torch.cuda.synchronize(); t = time.time()
for _ in range(20): A, _ = ops.kpconv_gather(q, s, idx, x, kp, 0.048)
torch.cuda.synchronize(); print("gather 4096x24x64: %.1f us" % ((time.time() - t) / 20 * 1e6))
A2 = A.view(4096, -1); W2 = W.view(-1, 64)
torch.cuda.synchronize(); t = time.time()
for _ in range(20): ops.gemm(A2, W2)
torch.cuda.synchronize(); print("gemm 4096x960x64: %.1f us" % ((time.time() - t) / 20 * 1e6))
