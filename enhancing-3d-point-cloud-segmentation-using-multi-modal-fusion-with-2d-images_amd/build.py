"""Builds libmvkpconv.so (the C-ABI HIP library, include/mvkpconv.h) in-tree for gfx950.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU box.
Usage:  python <pkg>/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmvkpconv.so")
ARCH = "gfx950"

# (source, extra flags). Geometry kernels must reproduce the reference's x86-64 (no FMA) float32
# arithmetic bit for bit -> contraction off there; the KPConv / GEMM kernels want FMAs.
SOURCES = [
    ("error.cpp", []),
    ("kpconv.hip", []),
    ("gemm.hip", []),
    ("gemm32s.hip", []),
    ("pool.hip", []),
    ("bn.hip", []),
    ("optim.hip", []),
    ("deform.hip", []),
    ("loss.hip", []),
    ("batchpad.hip", []),
    ("revlist.hip", []),
    ("subsample.hip", ["-ffp-contract=off"]),
    ("neighbors.hip", ["-ffp-contract=off"]),
    ("fusion.hip", ["-ffp-contract=off"]),
]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False):
    hipcc = _hipcc()
    common = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "blockscan.h"), os.path.join(HERE, "..", "include", "mvkpconv.h"),
              os.path.join(HERE, "..", "include", "mvk_prime_list.h"), os.path.abspath(__file__)]
    objs, jobs = [], []
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            raise FileNotFoundError(sp)
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        if force or _stale(obj, [sp] + common):
            cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-c", sp, "-o", obj] + extra
            if src.endswith(".cpp"):
                cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-c", sp, "-o", obj]
            jobs.append(cmd)
        objs.append(obj)
    if jobs:
        # independent translation units: a few compilers side by side (MVK_BUILD_JOBS, default 6 or the CPU count)
        from concurrent.futures import ThreadPoolExecutor
        n = max(1, min(int(os.environ.get("MVK_BUILD_JOBS", "6")), os.cpu_count() or 1, len(jobs)))

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(n) as pool:
            list(pool.map(run, jobs))
    if force or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
