"""Development-only: do two hipGraph replays issued from two host threads overlap (submission and execution)?"""
import threading, time, torch
dev = torch.device("cuda:0")
def make(n, size):
    x = torch.randn(size, device=dev)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(3):
            y = x
            for _ in range(n): y = y * 1.0001 + 0.1
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            y = x
            for _ in range(n): y = y * 1.0001 + 0.1
    return g, s
for size in (1 << 12, 1 << 20):
    ga, sa = make(320, size)
    gb, sb = make(220, size)
    def run(g, s, reps):
        with torch.cuda.stream(s):
            for _ in range(reps): g.replay()
        s.synchronize()
    for g, s, name in ((ga, sa, "A (320 nodes)"), (gb, sb, "B (220 nodes)")):
        run(g, s, 3); t = time.perf_counter(); run(g, s, 20); print(size, name, "alone: %.3f ms / replay" % ((time.perf_counter() - t) / 20 * 1e3))
    t = time.perf_counter()
    for _ in range(20):
        with torch.cuda.stream(sa): ga.replay()
        with torch.cuda.stream(sb): gb.replay()
    torch.cuda.synchronize(); print(size, "A then B from one thread: %.3f ms / pair" % ((time.perf_counter() - t) / 20 * 1e3))
    t = time.perf_counter()
    tb = threading.Thread(target=run, args=(gb, sb, 20)); tb.start(); run(ga, sa, 20); tb.join()
    print(size, "A and B from two threads: %.3f ms / pair" % ((time.perf_counter() - t) / 20 * 1e3), flush=True)
