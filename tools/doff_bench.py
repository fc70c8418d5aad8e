"""Development-only: mvk_kpconv_deform_doff alone at the shapes of the deformable levels."""
import os, sys, ctypes as C, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops = mvkpconv.sub("ops"); lib = mvkpconv.pkg._lib.lib()
def p(t): return C.c_void_p(t.data_ptr()) if t is not None else None
def run(N, H, Cin, keep_frac, reps=20):
    torch.manual_seed(0)
    K = 15
    s = torch.rand(N, 3, device="cuda")
    q = s.clone()
    idx = torch.randint(0, N, (N, H), device="cuda", dtype=torch.int32)
    r = 0.5 * keep_frac ** (1 / 3)
    kp = (torch.rand(K, 3, device="cuda") - 0.5) * r
    off = torch.zeros(N, K, 3, device="cuda")
    x = torch.randn(N, Cin, device="cuda"); dA = torch.randn(N, K, Cin, device="cuda")
    g = torch.randn(N, K, device="cuda"); arg = torch.zeros(N, K, dtype=torch.int32, device="cuda")
    out = torch.empty(N, K, 3, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def call():
        rc = lib.mvk_kpconv_deform_doff(p(q), N, p(s), N, p(idx), 0, H, p(x), Cin, p(kp), K, C.c_float(r * 0.6), 1, p(off), p(dA), p(g), p(arg), p(out), st)
        assert rc == 0
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    print("N %5d H %4d Cin %3d keep~%.2f : %.1f us" % (N, H, Cin, keep_frac, e0.elapsed_time(e1) / reps * 1e3), flush=True)
for a in [(960, 1000, 64, 0.3), (960, 1000, 64, 0.05), (960, 64, 64, 0.3), (256, 300, 128, 0.5), (64, 64, 256, 0.9), (64, 64, 256, 0.05), (64, 64, 16, 0.9), (4000, 200, 64, 0.3)]:
    run(*a)
