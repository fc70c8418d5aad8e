"""Development-only: do kernels of two HIP streams (one of them a graph replay) overlap on this GPU?"""
import time, torch
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
a = torch.randn(256, 256, device=dev); b = torch.randn(256, 256, device=dev)
c = torch.randn(256, 256, device=dev); d = torch.randn(256, 256, device=dev)
s2 = torch.cuda.Stream()
N = 400


def chain(x, y):
    for _ in range(N):
        x = torch.mm(x, y)
        x = x * 0.01
    return x


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


chain(a, b); torch.cuda.synchronize()
print("eager chain alone              %.2f ms" % timed(lambda: chain(a, b)))


def both_eager():
    with torch.cuda.stream(s2):
        chain(c, d)
    chain(a, b)


s2.wait_stream(torch.cuda.current_stream())
print("two eager chains, two streams  %.2f ms" % timed(both_eager))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = chain(a, b)
print("graph replay alone             %.2f ms" % timed(g.replay))
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=s2):
    out2 = chain(c, d)
print("graph replay (2nd) alone       %.2f ms" % timed(lambda: (s2.wait_stream(torch.cuda.current_stream()), torch.cuda.stream(s2).__enter__(), g2.replay(), torch.cuda.set_stream(torch.cuda.current_stream()))[0]))


def graph_plus_eager():
    g.replay()
    with torch.cuda.stream(s2):
        chain(c, d)


print("graph + eager chain            %.2f ms" % timed(graph_plus_eager))


def two_graphs():
    g.replay()
    with torch.cuda.stream(s2):
        g2.replay()


print("two graphs, two streams        %.2f ms" % timed(two_graphs))

# one graph with two parallel branches (fork / join inside the capture)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        o2 = chain(c, d)
    o1 = chain(a, b)
    cur.wait_stream(s2)
print("one graph, two branches        %.2f ms" % timed(g3.replay))

# interleaved single chain (what a scheduler-free merge would give)
g4 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g4):
    x, y = a, c
    for _ in range(N):
        x = torch.mm(x, b); y = torch.mm(y, d); x = x * 0.01; y = y * 0.01
print("one graph, interleaved chain   %.2f ms" % timed(g4.replay))
