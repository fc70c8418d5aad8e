"""Development-only: deformable mvk_kpconv_gather_fwd alone at the shapes of the deformable levels.
MVK_DEFORM_VEC=0 selects the one-point-per-wave kernel; outputs are saved / compared across the two runs."""
import os, sys, ctypes as C, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
lib = mvkpconv.pkg._lib.lib()
def p(t): return C.c_void_p(t.data_ptr()) if t is not None else None
TAG = os.environ.get("MVK_DEFORM_VEC", "1")
OUT = os.path.join(ROOT, "gpurun_out", "deform_fwd")
os.makedirs(OUT, exist_ok=True)
def run(N, H, Cin, keep_frac, reps=20):
    torch.manual_seed(0)
    K = 15
    s = torch.rand(N, 3, device="cuda")
    q = s.clone()
    idx = torch.randint(0, N + N // 8, (N, H), device="cuda", dtype=torch.int32)   # some shadow entries
    r = 0.5 * keep_frac ** (1 / 3)
    kp = (torch.rand(K, 3, device="cuda") - 0.5) * r
    off = (torch.rand(N, K, 3, device="cuda") - 0.5) * 0.2 * r
    x = torch.randn(N, Cin, device="cuda")
    A = torch.empty(N, K, Cin, device="cuda"); md = torch.empty(N, K, device="cuda")
    arg = torch.zeros(N, K, dtype=torch.int32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def call():
        rc = lib.mvk_kpconv_gather_fwd(p(q), N, p(s), N, p(idx), 0, H, p(x), Cin, p(kp), K, C.c_float(r * 0.6), 1, 1,
                                       p(off), p(md), p(arg), p(A), st)
        assert rc == 0
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    name = os.path.join(OUT, "o_%d_%d_%d_%.2f.pt" % (N, H, Cin, keep_frac))
    note = ""
    if TAG == "0":
        torch.save((A.cpu(), md.cpu(), arg.cpu()), name)
    elif os.path.exists(name):
        A0, m0, a0 = torch.load(name)
        note = " | vs one-point kernel: max|dA| %.2e (scale %.2e) min_d2 equal %s arg equal %.4f" % (
            (A.cpu() - A0).abs().max().item(), A0.abs().max().item(), torch.equal(md.cpu(), m0),
            (arg.cpu() == a0).float().mean().item())
    print("N %5d H %4d Cin %3d keep~%.2f : %.1f us%s" % (N, H, Cin, keep_frac, e0.elapsed_time(e1) / reps * 1e3, note), flush=True)
for a in [(960, 1000, 64, 0.3), (960, 1000, 64, 0.05), (960, 64, 64, 0.3), (256, 300, 128, 0.5), (64, 64, 256, 0.9),
          (64, 64, 512, 0.5), (64, 64, 16, 0.9), (4000, 200, 64, 0.3), (4000, 37, 66, 0.3), (777, 45, 13, 0.5)]:
    run(*a)
