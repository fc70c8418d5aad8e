"""TEST / BASELINE INFRASTRUCTURE ONLY -- the pyramid of one sphere built by P worker PROCESSES at once,
mirroring the reference's DataLoader workers (`input_threads`, KPConv-PyTorch/train_ScanNet_sphere.py:58,
num_workers at :365-377): each worker builds `n` five-level pyramids (datasets/common.py:779-900: conv / pool /
upsample neighbours + grid subsampling per level) with the single-threaded compiled reference core (oracle/_ref)
or the C oracle. Run as a child process by bench.py's cpu_baseline leg (never from the product path):

    python -m oracle.pyramid_workers <cloud.npy> <P> <n> <dl> <conv_radius>

prints one JSON line {"workers", "pyramids_per_worker", "wall_s", "spheres_per_s", "impl"}."""
import json
import multiprocessing as mp
import sys
import time

import numpy as np


def pyramid(cport, p, dl, conv_radius, impl):
    """The 13 neighbour searches + 4 subsamplings of a five-level pyramid (no rotation: cost-neutral)."""
    l = np.array([p.shape[0]], np.int32)
    r = dl * conv_radius
    for lvl in range(5):
        cport.radius_neighbors_batch(p, p, l, l, r, impl=impl)
        if lvl < 4:
            q, ql = cport.subsample_batch(p, l, dl=2 * r / conv_radius, impl=impl)
            cport.radius_neighbors_batch(q, p, ql, l, r, impl=impl)
            cport.radius_neighbors_batch(p, q, l, ql, 2 * r, impl=impl)
            p, l = q, ql
        r *= 2


def _worker(path, n, dl, conv_radius, barrier, out, i):
    from oracle import cport
    impl = "ref" if cport.ref() else "oracle"
    cloud = np.load(path)
    pyramid(cport, cloud, dl, conv_radius, impl)        # warm-up (page in the library, allocator)
    barrier.wait()
    t0 = time.time()
    for _ in range(n):
        pyramid(cport, cloud, dl, conv_radius, impl)
    out[2 * i], out[2 * i + 1] = t0, time.time()


def run(path, P, n, dl, conv_radius):
    ctx = mp.get_context("fork")           # this process never touches the GPU
    barrier = ctx.Barrier(P)
    out = ctx.Array("d", 2 * P)
    procs = [ctx.Process(target=_worker, args=(path, n, dl, conv_radius, barrier, out, i)) for i in range(P)]
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    if any(p.exitcode != 0 for p in procs):
        raise SystemExit("a pyramid worker failed")
    wall = max(out[1::2]) - min(out[0::2])
    from oracle import cport
    return {"workers": P, "pyramids_per_worker": n, "wall_s": wall, "spheres_per_s": P * n / wall,
            "impl": "ref" if cport.ref() else "oracle"}


if __name__ == "__main__":
    a = sys.argv[1:]
    print(json.dumps(run(a[0], int(a[1]), int(a[2]), float(a[3]), float(a[4]))))
