"""Development-only: the fp16 contraction of the rigid layers -- streaming kernel (gemm_f16_stream, tiles per
workgroup swept through MVK_GEMM16_TILES) against the LDS-staged fp16 kernel and the f32 MFMA kernel, device time of
graph-captured launches. usage: python tools/gemm16_bench.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops = mvkpconv.sub("ops")
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


for (M, N, Kd) in [(19464, 64, 990), (19464, 32, 480), (4986, 64, 960), (40000, 64, 75), (40000, 32, 480), (10000, 64, 960),
                   (76700, 32, 480), (171000, 64, 990)]:
    Kp = (Kd + 31) // 32 * 32
    A32 = torch.randn(M, Kd, device=dev)
    W = torch.randn(Kd, N, device=dev) * 0.1
    A16p = torch.zeros(M, Kp, device=dev, dtype=torch.float16)
    A16p[:, :Kd] = A32.half()
    A16 = A32.half().contiguous()
    Wt, Wr = ops.round_weights_f16(W, ops.gemm_f16_stream_plan(M, N, Kp)[3], True)
    W16 = W.half()
    nv = torch.tensor([M], dtype=torch.int32, device=dev)
    fl = 2.0 * M * N * Kd
    out = torch.zeros(M, N, device=dev)
    us32 = timeit(lambda: ops.gemm(A32, W, out=out))
    if ops.gemm_f32_stream_plan(M, N, Kd)[0]:
        Abuf = torch.zeros(M * Kd + 64, device=dev)
        As = Abuf[:M * Kd].view(M, Kd)
        As.copy_(A32)
        us32s = timeit(lambda: ops.gemm_f32_stream(As, W))
        us32ss = timeit(lambda: ops.gemm_f32_stream(As, W, nv))
        print("%-22s f32 tiled %6.1f us | f32 stream %6.1f / with statistics %6.1f us (%.1f TF = %.2f of the f32 MFMA peak)"
              % ((M, N, Kd), us32, us32s, us32ss, fl / us32s / 1e6, fl / us32s / 1e6 / 157.3), flush=True)
    us16 = timeit(lambda: ops.gemm_f16(A16, W16))
    line = "%-22s f32 %6.1f us | f16 staged %6.1f us |" % ((M, N, Kd), us32, us16)
    for tiles in ("", "3", "4", "5", "6", "8", "12"):
        if tiles:
            os.environ["MVK_GEMM16_TILES"] = tiles
        else:
            os.environ.pop("MVK_GEMM16_TILES", None)
        t = ops.gemm_f16_stream_plan(M, N, Kp)[1]
        us = timeit(lambda: ops.gemm_f16_stream(A16p, Wt))
        uss = timeit(lambda: ops.gemm_f16_stream(A16p, Wt, nv))
        line += " T%s=%d: %5.1f/%5.1f us (%.0f TF, %.2f TB/s)" % ("*" if not tiles else "", t, us, uss, fl / us / 1e6, (M * Kp * 2 + M * N * 4) / us / 1e6)
    os.environ.pop("MVK_GEMM16_TILES", None)
    usw = timeit(lambda: ops.round_weights_f16(W, Wt.shape[1], True))
    print(line + " | round_weights %.1f us" % usw, flush=True)
