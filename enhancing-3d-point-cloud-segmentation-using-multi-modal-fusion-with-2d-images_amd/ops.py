"""torch.autograd wrappers over the C ABI (include/mvkpconv.h).

PyTorch only owns the HBM buffers and the HIP stream; all compute below is in libmvkpconv.so.
Every function requires CUDA(HIP) tensors and raises otherwise -- there is no CPU path here.
"""
import collections
import os
import ctypes as C

import weakref

import torch

from ._lib import lib, check

INFLUENCE = {"constant": 0, "linear": 1, "gaussian": 2}
AGGREGATION = {"sum": 0, "closest": 1}


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("MV-KPConv ops need tensors resident in HBM (got a CPU tensor); "
                               "there is no CPU fallback")


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _idx(t):
    if t.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("neighbour indices must be int32 or int64")
    return t.contiguous(), int(t.dtype == torch.int64)


# --------------------------------------------------------------------------------------------
# zero arena (opt-in): one fill per step instead of one per zero-initialised output
# --------------------------------------------------------------------------------------------
# Split-K GEMM outputs and scatter targets must start at zero; eagerly that is ~100 tiny fill kernels per
# step. With the arena enabled, step_begin() zeroes one big buffer with a single launch and _zeros()
# hands out 256-byte aligned slices of it. Slices are only valid until the next step_begin(): this is for
# the captured-graph step of bench.py (fixed addresses), not a general allocator.
_ARENA = {"on": False, "buf": None, "off": 0, "high": 0}


def zero_arena_enable(nbytes, device):
    _ARENA.update(on=True, buf=torch.zeros(int(nbytes), dtype=torch.uint8, device=device), off=0)


def zero_arena_disable():
    _ARENA.update(on=False, buf=None, off=0)


def zero_arena_high_water(reset=False):
    """Largest number of bytes _zeros() handed out (or would have) between two step_begin() calls;
    reset=True starts a new measurement (call it right before the steps that size the arena)."""
    if reset:
        _ARENA["off"] = 0
        _ARENA["high"] = 0
    return _ARENA["high"]


def step_begin():
    _ARENA["high"] = max(_ARENA["high"], _ARENA["off"])
    _ARENA["off"] = 0
    if _ARENA["on"]:
        _ARENA["buf"].zero_()


def _zeros(shape, device, dtype=torch.float32):
    n = 1
    for s in shape:
        n *= int(s)
    nbytes = (n * torch.empty((), dtype=dtype).element_size() + 255) // 256 * 256
    off = _ARENA["off"]
    _ARENA["off"] = off + nbytes
    if _ARENA.get("log") is not None:
        import traceback
        _ARENA["log"].append((nbytes, tuple(shape), traceback.extract_stack(limit=3)[0].name))
    buf = _ARENA["buf"]
    if _ARENA["on"] and n > 0 and off + nbytes <= buf.numel() and buf.device == device:
        return buf[off:off + nbytes].view(dtype)[:n].view(shape)
    return torch.zeros(shape, device=device, dtype=dtype)


# --------------------------------------------------------------------------------------------
# raw kernels
# --------------------------------------------------------------------------------------------

# --------------------------------------------------------------------------------------------
# deterministic mode: ordered reductions (run-to-run bit-identical products, no zero-initialised outputs)
# --------------------------------------------------------------------------------------------
# By default a product whose reduction is split over workgroups adds its partial sums with f32 atomics: the fastest way
# (fire and forget), but the sum depends on the order the workgroups ran in -- rounding only, yet a LeakyReLU input within
# an ulp of zero then takes the other slope and two runs of one network differ visibly. set_deterministic(True) (or
# MVK_DETERMINISTIC=1) hands the library an arena (mvk_gemm_split_arena): every split product then parks its partial
# tiles in HBM and the last-arriving workgroup of a tile adds them in split order (csrc/gemm.hip), the bias gradient of
# the BatchNorm-less layers is summed in workgroup order, outputs need no zero fill and split plans keep their
# BatchNorm-statistics epilogue. The price is three memory-side round trips at the tail of every split launch (~4 us
# each, +0.2 ms on the 4.1 ms early-fusion step): opt-in, like torch.use_deterministic_algorithms.
_DET = {"on": os.environ.get("MVK_DETERMINISTIC", os.environ.get("MVK_GEMM_ORDERED", "0")) == "1"}
_SPLIT_ARENA = {}
_SPLIT_ARENA_BYTES = int(os.environ.get("MVK_GEMM_ARENA_MB", "1024")) << 20
_SPLIT_COUNTERS = 1 << 20


def set_deterministic(flag, device=None):
    """Turns the ordered reductions on or off (see above). Turning them on creates the arena on `device` (default: the
    current one) at once: do it outside any graph capture."""
    _DET["on"] = bool(flag)
    # (torch.backends.cudnn.deterministic is NOT touched: on this image it drops MIOpen to naive convolution kernels, 4 ms
    # -> 226 ms per step; the frozen 2D encoder is a library network outside this mode)
    if flag:
        split_arena_prepare(torch.device("cuda", torch.cuda.current_device()) if device is None else device)
    else:
        split_arena_release()


def is_deterministic():
    """True when the ordered reductions are on. With MVK_DETERMINISTIC=1 the arena used to appear only with the first
    product, so a pyramid built before it saw False (rows in arrival order, no transposed upsampling lists: the first
    step of an env-enabled run was not bit-reproducible): the first question creates the arena."""
    if _DET["on"] and not _SPLIT_ARENA and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        split_arena_prepare(torch.device("cuda", torch.cuda.current_device()))
    return bool(_DET["on"] and _SPLIT_ARENA)


def split_arena_prepare(device):
    """Creates the arena of the ordered reductions on `device` in deterministic mode (no-op otherwise, or when it exists).
    Called by the first product; call it (or set_deterministic) yourself before capturing a graph that was never run
    eagerly."""
    if not _DET["on"]:
        return False
    device = torch.device(device)
    if device.index in _SPLIT_ARENA:
        return True
    if _SPLIT_ARENA:
        raise RuntimeError("the ordered reductions use ONE arena per process (one GPU per process); it lives on "
                           "cuda:%d" % next(iter(_SPLIT_ARENA)))
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("ops: the arena of the ordered reductions does not exist yet and cannot be created inside a "
                           "graph capture; run one eager step first or call ops.set_deterministic(True) before")
    ws = torch.empty(_SPLIT_ARENA_BYTES, dtype=torch.uint8, device=device)
    cnt = torch.zeros(_SPLIT_COUNTERS, dtype=torch.int32, device=device)
    torch.cuda.current_stream(device).synchronize()          # the counters are zero before any stream uses them
    check(lib().mvk_gemm_split_arena(_p(ws), ws.numel(), _p(cnt), cnt.numel()))
    _SPLIT_ARENA[device.index] = (ws, cnt)
    return True


def split_arena_release():
    """Back to the atomic split reductions; frees the arena."""
    if _SPLIT_ARENA:
        torch.cuda.synchronize()
        check(lib().mvk_gemm_split_arena(None, 0, None, 0))
        _SPLIT_ARENA.clear()


def _split_out(shape, device, split, keep=False):
    """Output buffer of a product whose reduction is split `split` ways: the ordered reduction writes every element
    (empty); the atomic one accumulates onto zeros (arena slice, or a tensor of its own when it outlives the step)."""
    if split <= 1 or device.index in _SPLIT_ARENA:
        return torch.empty(shape, device=device, dtype=torch.float32)
    return torch.zeros(shape, device=device, dtype=torch.float32) if keep else _zeros(shape, device)


def gemm_plan(M, N, Kd, split_k=None, want_stats=False):
    """(split, stat_rows) mvk_gemm_f32_ex will use for this shape: the split of the reduction (the output must
    be zero-initialised when > 1) and the row-block size of the BatchNorm partials (0 = none produced)."""
    sp, rows = C.c_int(0), C.c_int(0)
    check(lib().mvk_gemm_f32_plan(int(M), int(N), int(Kd), int(split_k or 0), int(bool(want_stats)), C.byref(sp),
                                  C.byref(rows)))
    return sp.value, rows.value


def kpconv_gather_plan(Nq, Ns, H, Cin, elem_bytes=4, deformable=False):
    """Launch geometry of the gather kernel for a (linear, sum) layer (mvk_kpconv_gather_plan): dict with the
    lanes per point, points per wave, rows per batch, first sharing workgroup, waves per workgroup, workgroups and
    grid threads; 'workgroups' 0 means the layer runs on the one-point-per-wave kernel."""
    out = (C.c_int64 * 8)()
    check(lib().mvk_kpconv_gather_plan(int(Nq), int(Ns), int(H), int(Cin), int(elem_bytes), int(bool(deformable)), out))
    keys = ("lanes_per_point", "points_per_wave", "rows_per_batch", "first_sharing_workgroup", "waves_per_workgroup",
            "workgroups", "grid_threads", "mfma")
    return dict(zip(keys, [int(v) for v in out]))


# rows up to which a BatchNorm takes its statistics from the producing GEMM's epilogue: every workgroup of the
# normalising launch merges ALL the epilogue's (sum, M2) partials of its channels, one per 64 rows -- beyond a few
# hundred of them (FeatureAggregation's 58 392 rows: 913) the separate statistics launch is the cheaper way
_STATS_EPILOGUE_ROWS = int(os.environ.get("MVK_GEMM_STATS_MAX_ROWS", "32768"))

BN_SMALL_ROWS = 128      # csrc/bn.hip: up to this many rows one launch does statistics and normalisation (any D)


def bn_single_launch_rows(D):
    """Rows up to which the BatchNorm of a D-channel input is one launch (csrc/bn.hip: 128, or 1024 when D % 4 == 0):
    the producing GEMM then skips its statistics epilogue."""
    return lib().mvk_bn_single_launch_rows(int(D))


# --------------------------------------------------------------------------------------------
# BatchNorm folded into the products around it (round 5; csrc/gemm.hip GemmArgs (a), (b))
# --------------------------------------------------------------------------------------------
# The producer's half: a product that writes the statistics partials of its output can also FINISH them (mean, invstd,
# running statistics, batch counter) -- the workgroup that arrives last at a column tile's counter merges the partials.
# The op wrappers (linear, linear_pair, kpconv, upsample_cat_linear) take `bn=` (the nn.BatchNorm1d that follows) and
# park it here around their autograd node; gemm() picks it up and reports the result, which travels with the output as
# the third element of `_mvk_bn_stats`. The consumer's half: bn_lrelu_linear().
# MEASURED SLOWER on the one-sphere step and therefore OFF by default (MVK_BN_FOLD=1 / ops.BN_FOLD = True turn it on):
# network chain 2.85 -> 2.94 ms with finished statistics alone, 2.96 ms with the operand fold on top (round 5,
# profiles/r05_bn_fold_chain.txt). The last-arriver hand-off at the tail of a producing product costs
# +5..13 us per launch (store acknowledge, counter round trip, 3-5 dependent rounds of memory-side loads over up to 305
# partials) against the 2-4 us the normalising launch saves by not reducing them, and in a bottleneck block with a
# shortcut layer the convolution's BatchNorm already shares its launch with the shortcut's.
BN_FOLD = os.environ.get("MVK_BN_FOLD", "0") == "1"
_FIN = {"req": None, "done": None}
_AX = {"req": None}


def _fin_counters(bn, device):
    """Persistent zero int32 words of one BatchNorm module (the kernel returns them to zero): one per column tile."""
    c = getattr(bn, "_mvk_fin_counters", None)
    if c is None or c.device != device:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ops: the finish counters of a BatchNorm do not exist yet and cannot be created inside a "
                               "graph capture; run one eager step first")
        c = torch.zeros(128, dtype=torch.int32, device=device)
        torch.cuda.current_stream(device).synchronize()
        bn._mvk_fin_counters = c
    return c


def _fin_request(bn):
    """Parks the BatchNorm that follows the product about to be issued (None: no request). Only training-mode modules
    with per-rank statistics qualify; anything else keeps the separate statistics pass."""
    ok = (BN_FOLD and bn is not None and bn.training and _SYNC_BN["group"] is None and bn.num_features % 4 == 0)
    _FIN["req"] = bn if ok else None
    _FIN["done"] = None


def _fin_take():
    done = _FIN["done"]
    _FIN["req"] = None
    _FIN["done"] = None
    return done


def _fin_struct(bn, N, device):
    from ._lib import BnFinish
    if bn.num_features != N:
        raise RuntimeError("ops: the BatchNorm handed to a product has %d channels, the product %d columns" % (bn.num_features, N))
    mean = torch.empty(N, device=device, dtype=torch.float32)
    invstd = torch.empty(N, device=device, dtype=torch.float32)
    track = bn.track_running_stats and bn.running_mean is not None
    fin = BnFinish(_p(_fin_counters(bn, device)), float(bn.eps), float(bn.momentum if bn.momentum is not None else 0.0),
                   _p(mean), _p(invstd), _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None,
                   _p(bn.num_batches_tracked) if track else None)
    return fin, (mean, invstd)


def gemm(A, B, transA=False, transB=False, out=None, accumulate=False, split_k=None, keep=False, stats_n_valid=None):
    """C = op(A) @ op(B) on v_mfma_f32_16x16x4_f32 (mvk_gemm_f32_ex). split_k=None lets the library pick the
    row tile and the split of the reduction so that small-M / deep-K products (the coarse KPConv layers:
    85 x 7680 x 512) still fill the 256 CUs. keep=True: the result outlives the step (a weight gradient that
    becomes .grad), so it must not be a slice of the per-step zero arena.
    stats_n_valid (DEVICE int32 [1]): also produce the column statistics of C over its first n_valid rows for
    the BatchNorm that follows; returns (C, (partials, rows per block)) -- or (C, None) when the chosen plan
    splits the reduction. A BatchNorm parked by _fin_request() is finished inside the launch when the product carries the
    statistics epilogue (result in _FIN["done"]); an operand transform parked by bn_lrelu_linear() is applied to A."""
    _dev(A, B)
    A, B = _f32c(A), _f32c(B)
    M, Kd = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    N = B.shape[0] if transB else B.shape[1]
    assert (B.shape[1] if transB else B.shape[0]) == Kd, "gemm: inner dimensions differ"
    want = stats_n_valid is not None
    ax = _AX["req"]
    if ax is not None:
        _AX["req"] = None
        if ax["lazy"].data_ptr() != A.data_ptr() or transA or not transB or out is not None or accumulate or M == 0:
            raise RuntimeError("ops.gemm: a parked operand transform did not meet its product")
    if M == 0 or N == 0 or Kd == 0:
        res = out.zero_() if out is not None else torch.zeros((M, N), device=A.device, dtype=torch.float32)
        return (res, None) if want else res
    split_arena_prepare(A.device)
    req = _FIN["req"]
    # a finished BatchNorm needs the epilogue statistics whatever the row count (the one-launch BatchNorm kernels of few
    # rows would otherwise do statistics + normalisation themselves: with a fold there is no such launch)
    stats_ok = want and out is None and not accumulate and M <= _STATS_EPILOGUE_ROWS
    if req is None and ax is None:
        stats_ok = stats_ok and M > bn_single_launch_rows(N)
    split, rows = gemm_plan(M, N, Kd, split_k, stats_ok)
    if out is None:
        out = _split_out((M, N), A.device, split, keep)
    part = torch.empty(((M + rows - 1) // rows, 2, N), device=A.device, dtype=torch.float32) if rows > 0 else None
    fin = None
    if req is not None and part is not None and not transA and split_k is None:
        fin, _FIN["done"] = _fin_struct(req, N, A.device)
        _FIN["req"] = None
    if fin is not None or ax is not None:
        xf = None
        if ax is not None:
            from ._lib import ATransform
            xf = ATransform(_p(ax["mean"]), _p(ax["invstd"]), _p(ax["gamma"]), _p(ax["beta"]), float(ax["slope"]),
                            _p(ax["n_valid"]), _p(ax["lazy"]))
            A = ax["raw"]
        check(lib().mvk_gemm_f32_bn(_p(A), _p(B), _p(out), M, N, Kd, int(transB), _p(part),
                                    _p(stats_n_valid) if part is not None else None,
                                    C.byref(fin) if fin is not None else None, C.byref(xf) if xf is not None else None,
                                    _stream()))
    else:
        check(lib().mvk_gemm_f32_ex(_p(A), _p(B), _p(out), M, N, Kd, int(transA), int(transB), int(accumulate),
                                    int(split), _p(part), _p(stats_n_valid) if part is not None else None, _stream()))
    if want:
        return out, ((part, rows) if part is not None else None)
    return out


def gemm_f32_stream_plan(M, N, Kd):
    """(runs on the streaming kernel, 16-row tiles per workgroup, workgroups, floats needed behind A) for
    C [M,N] = A [M,Kd] . B [Kd,N] in f32 (mvk_gemm_f32_stream_plan)."""
    out = (C.c_int64 * 4)()
    check(lib().mvk_gemm_f32_stream_plan(int(M), int(N), int(Kd), out))
    return bool(out[0]), int(out[1]), int(out[2]), int(out[3])


def _slack_floats(t):
    """Readable float32 elements behind the last element of the (contiguous) tensor t inside its own storage."""
    return t.untyped_storage().nbytes() // 4 - t.storage_offset() - t.numel()


def gemm_f32_stream(A, B, stats_n_valid=None):
    """C [M,N] f32 = A [M,Kd] . B [Kd,N] on the streaming f32 MFMA kernel (weights stationary in registers, rows of A as
    fragment loads). Returns (C, (partials, rows per block) or None). The caller checks gemm_f32_stream_plan and that A
    has the slack the plan asks for (kpconv_gather allocates it)."""
    _dev(A, B, stats_n_valid)
    A, B = _f32c(A), _f32c(B)
    M, Kd = A.shape
    N = B.shape[1]
    ok, tiles, wgs, need = gemm_f32_stream_plan(M, N, Kd)
    if not ok or B.shape[0] != Kd:
        raise RuntimeError("gemm_f32_stream: unsupported shape %d x %d x %d" % (M, N, Kd))
    y = torch.empty((M, N), device=A.device, dtype=torch.float32)
    part = torch.empty((wgs, 2, N), device=A.device, dtype=torch.float32) if stats_n_valid is not None else None
    check(lib().mvk_gemm_f32_stream(_p(A), _slack_floats(A), _p(B), _p(y), M, N, Kd, _p(stats_n_valid), _p(part), _stream()))
    return y, ((part, 16 * tiles) if part is not None else None)


# ---- optional per-launch timing of the gather kernel (bench.py roofline): HIP events on the
# current stream around the C-ABI call; nothing is recorded unless profile_reset(enabled=True).
_PROF = {"on": False, "rec": [], "gemm": []}


def profile_reset(enabled):
    _PROF["on"] = bool(enabled)
    _PROF["rec"] = []
    _PROF["gemm"] = []
    _PROF["h_eff"] = {}


def _timing_events():
    """Start / stop events for one launch. Eager steps are host-bound, so the queue is usually empty when
    the start event is enqueued and the event would also time the host's launch latency of the kernel
    behind it: a short device-side spin first keeps the GPU busy until start event, kernel and stop event
    are all queued."""
    torch.cuda._sleep(200000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0, e1


def profile_collect_contraction():
    """{(M, Kd, N): {launches, total_ms, flops_per_launch}} of the forward K x Cin x Cout contraction
    (gemm_f32_mfma NN); flops = 2*M*Kd*N (SURVEY.md 8d F_mfma)."""
    torch.cuda.synchronize()
    out = {}
    for key, e0, e1 in _PROF["gemm"]:
        r = out.setdefault(key, {"launches": 0, "total_ms": 0.0, "flops_per_launch": 2.0 * key[0] * key[1] * key[2]})
        r["launches"] += 1
        r["total_ms"] += e0.elapsed_time(e1)
    return out


def _gather_kernel_name(Cin, deform):
    if deform:
        if 13 <= Cin <= 512:
            lpp = min((Cin + 3) // 4, 64)
            return "kpconv_gather_vec<NCH=%d,deform>(LPP=%d,PPW=%d)" % (1 if Cin <= 256 else 2, lpp, 64 // lpp)
        return "kpconv_lane_channel<fwd,deform>"
    if Cin <= 4:
        return "kpconv_gather_small(4 lanes per point)"
    if Cin <= 512:
        lpp = min((Cin + 3) // 4, 64)
        return "kpconv_gather_vec<NCH=%d>(LPP=%d,PPW=%d)" % (1 if Cin <= 256 else 2, lpp, 64 // lpp)
    return "kpconv_lane_channel<fwd>"


def _gather_kernel_label(Nq, Ns, H, Cin, deform, elem_bytes=4):
    """Kernel name with the launch geometry the library actually uses (mvk_kpconv_gather_plan)."""
    p = kpconv_gather_plan(Nq, Ns, H, Cin, elem_bytes, deform)
    if p["workgroups"] == 0:
        return _gather_kernel_name(Cin, deform)
    if p.get("mfma"):
        return "kpconv_gather_mfma<T=%d>(1 point per wave)" % p["rows_per_batch"]
    tail = Cin - 4 * p["lanes_per_point"] if 4 * p["lanes_per_point"] < Cin else 0
    return "kpconv_gather_vec<NCH=%d%s>(LPP=%d,PPW=%d%s)" % (1 if Cin <= 256 else 2, ",deform" if deform else "",
                                                             p["lanes_per_point"], p["points_per_wave"],
                                                             ",+%d trailing channels" % tail if tail else "")


def profile_collect(h_eff=None):
    """{(kernel, Nq, Ns, H, Cin, K): {launches, total_ms, bytes_per_launch, ...}}; h_eff maps
    (Nq, Ns, H) -> mean number of real (non-shadow) neighbours per row."""
    torch.cuda.synchronize()
    out = {}
    for key, e0, e1 in _PROF["rec"]:
        r = out.setdefault(key, {"launches": 0, "total_ms": 0.0, "each_ms": []})
        r["launches"] += 1
        t = e0.elapsed_time(e1)
        r["total_ms"] += t
        r["each_ms"].append(t)
    for key, r in out.items():
        name, Nq, Ns, H, Cin, K = key
        he = _PROF.get("h_eff", {}).get((Nq, Ns, H)) if name.endswith("[dx]") else None
        if he is None:
            he = (h_eff or {}).get((Nq, Ns, H), H)
        # algorithmic bytes: feature row + xyz + int32 index per real neighbour,
        # query xyz, and the [Nq,K,Cin] aggregate written by the gather kernel
        sx = 4
        r["bytes_per_launch"] = Nq * he * (Cin * sx + 12 + 4) + Nq * 12 + Nq * K * Cin * sx
        r["kernel"] = name
        r["shape"] = {"Nq": Nq, "Ns": Ns, "H": H, "H_eff": he, "Cin": Cin, "K": K}
    return out


def kpconv_gather(q, s, idx, x, kp, extent, influence="linear", aggregation="sum", offsets=None,
                  want_min_d2=False, order=None, tag=None):
    """A[n,k,c] = sum_h w[n,h,k] x+[idx[n,h],c]; returns (A, min_d2 or None). order [Nq] int32 (a permutation of the
    query rows, e.g. sorted by grid cell): the order the points are WORKED on; the result does not depend on it."""
    _dev(q, s, idx, x, kp, offsets, order)
    if order is not None and (order.dtype != torch.int32 or order.shape != (q.shape[0],) or not order.is_contiguous()):
        raise RuntimeError("kpconv_gather: order must be a contiguous int32 tensor of Nq entries")
    q, s, x, kp = _f32c(q), _f32c(s), _f32c(x), _f32c(kp)
    idx, i64 = _idx(idx)
    Nq, Ns, H, Cin, K = q.shape[0], s.shape[0], idx.shape[1] if idx.dim() == 2 else 0, x.shape[1], kp.shape[0]
    if x.shape[0] != Ns:
        raise RuntimeError("kpconv: features and support points differ in length")
    # 64 floats of slack behind the aggregate: the streaming contraction reads up to 31 floats past the last row
    A = torch.empty(Nq * K * Cin + 64, device=q.device, dtype=torch.float32)[:Nq * K * Cin].view(Nq, K, Cin)
    min_d2 = min_arg = None
    if offsets is not None:
        offsets = _f32c(offsets)
        if want_min_d2:
            min_d2 = torch.empty((Nq, K), device=q.device, dtype=torch.float32)
            min_arg = torch.empty((Nq, K), device=q.device, dtype=torch.int32)
    if _PROF["on"]:
        e0, e1 = _timing_events()
    check(lib().mvk_kpconv_gather_fwd_ordered(_p(q), Nq, _p(s), Ns, _p(idx), i64, H, _p(x), Cin, _p(kp), K,
                                              float(extent), INFLUENCE[influence], AGGREGATION[aggregation],
                                              _p(offsets), _p(min_d2), _p(min_arg), _p(A), _p(order), _stream()))
    if _PROF["on"]:
        e1.record()
        name = _gather_kernel_label(Nq, Ns, H, Cin, offsets is not None)
        if tag:         # the gather-form feature gradient: its own class, with the mean length of ITS (reverse) rows
            name += tag
            _PROF.setdefault("h_eff", {})[(Nq, Ns, H)] = float((idx < Ns).sum().item()) / max(Nq, 1)
        _PROF["rec"].append(((name, Nq, Ns, H, Cin, K), e0, e1))
    if min_d2 is not None:
        min_d2._mvk_min_arg = min_arg          # neighbour column of each minimum: the backward's min_d2 path starts there
    return A, min_d2


def kpconv_scatter(q, s, idx, dA, kp, extent, influence="linear", aggregation="sum", x=None,
                   offsets=None, g_min_d2=None, min_arg=None):
    """dx[idx[n,h],c] += sum_k w[n,h,k] dA[n,k,c]; returns (dx, d_offsets or None)."""
    _dev(q, s, idx, dA, kp)
    q, s, dA, kp = _f32c(q), _f32c(s), _f32c(dA), _f32c(kp)
    idx, i64 = _idx(idx)
    Nq, Ns, H, K, Cin = q.shape[0], s.shape[0], idx.shape[1], kp.shape[0], dA.shape[2]
    dx = _zeros((Ns, Cin), q.device)
    d_off = None
    if offsets is not None:
        offsets, x = _f32c(offsets), _f32c(x)
        # written whole by mvk_kpconv_deform_doff -- which mvk_kpconv_scatter_bwd does not reach for an empty
        # neighbour matrix (its early return): zeros then
        d_off = (torch.zeros if (Nq == 0 or H == 0) else torch.empty)((Nq, K, 3), device=q.device, dtype=torch.float32)
        if g_min_d2 is not None:
            g_min_d2 = _f32c(g_min_d2)
            if min_arg is None:
                raise RuntimeError("kpconv_scatter: the min_d2 gradient needs the forward's arg-min columns")
    check(lib().mvk_kpconv_scatter_bwd(_p(q), Nq, _p(s), Ns, _p(idx), i64, H, Cin, _p(kp), K, float(extent),
                                       INFLUENCE[influence], AGGREGATION[aggregation], _p(dA), _p(dx),
                                       _p(x), _p(offsets), _p(g_min_d2), _p(min_arg), _p(d_off), _stream()))
    return dx, d_off


def kpconv_deform_doff(q, s, idx, x, kp, extent, offsets, dA, g_min_d2=None, min_arg=None):
    """d_offsets [Nq,K,3] of a deformable KPConv alone (mvk_kpconv_deform_doff; kpconv_scatter runs it together with the
    atomic dx scatter): what is left of the scatter entry point once dx is a gather (kpconv_gather_rev_deform)."""
    _dev(q, s, idx, x, kp, offsets, dA, g_min_d2, min_arg)
    q, s, x, kp, offsets, dA = _f32c(q), _f32c(s), _f32c(x), _f32c(kp), _f32c(offsets), _f32c(dA)
    idx, i64 = _idx(idx)
    Nq, Ns, H, K, Cin = q.shape[0], s.shape[0], idx.shape[1], kp.shape[0], dA.shape[2]
    if Nq == 0 or H == 0:
        return torch.zeros((Nq, K, 3), device=q.device, dtype=torch.float32)
    d_off = torch.empty((Nq, K, 3), device=q.device, dtype=torch.float32)
    if g_min_d2 is not None:
        g_min_d2 = _f32c(g_min_d2)
        if min_arg is None:
            raise RuntimeError("kpconv_deform_doff: the min_d2 gradient needs the forward's arg-min columns")
    check(lib().mvk_kpconv_deform_doff(_p(q), Nq, _p(s), Ns, _p(idx), i64, H, _p(x), Cin, _p(kp), K, float(extent),
                                       INFLUENCE["linear"], _p(offsets), _p(dA), _p(g_min_d2), _p(min_arg), _p(d_off),
                                       _stream()))
    return d_off


def kpconv_gather_rev_deform(s, q, rev, g, kp, extent, offsets, modulations=None, order=None):
    """A2 [Ns, K, C]: the forward aggregation of g [Nq, C] over the transposed relation rev [Ns, Hr] with the kernel points
    of the neighbour (query) rows, kp + offsets [Nq,K,3], times modulations [Nq,K] (mvk_kpconv_gather_rev_deform): the
    gather form of a deformable layer's feature gradient, dx = kp_transposed_contraction(A2, W)."""
    _dev(s, q, rev, g, kp, offsets, modulations, order)
    s, q, g, kp, offsets = _f32c(s), _f32c(q), _f32c(g), _f32c(kp), _f32c(offsets)
    if modulations is not None:
        modulations = _f32c(modulations)
    rev, r64 = _idx(rev)
    Ns, Nq, Hr, Cc, K = s.shape[0], q.shape[0], rev.shape[1], g.shape[1], kp.shape[0]
    if rev.shape[0] != Ns or g.shape[0] != Nq or offsets.shape != (Nq, K, 3):
        raise RuntimeError("kpconv_gather_rev_deform: rev [Ns,Hr], g [Nq,C] and offsets [Nq,K,3] expected")
    A2 = torch.empty(Ns * K * Cc + 64, device=s.device, dtype=torch.float32)[:Ns * K * Cc].view(Ns, K, Cc)
    check(lib().mvk_kpconv_gather_rev_deform(_p(s), Ns, _p(q), Nq, _p(rev), r64, Hr, _p(g), Cc, _p(kp), K, float(extent),
                                             _p(offsets), _p(modulations), _p(A2), _p(order), _stream()))
    return A2


# --------------------------------------------------------------------------------------------
# weight-gradient products off the critical chain
# --------------------------------------------------------------------------------------------
# The backward of a layer launches two independent products, dW = A^T g and dA = g W^T; only dA feeds the rest
# of the backward chain, dW is not read before the optimiser. Every product here is a small, latency-bound
# launch (~10 us whatever its size), so inside `with overlap_weight_grads():` the dW products are enqueued on a
# side stream forked from the current one (a parallel branch of the captured hipGraph) and joined when the scope
# ends -- about a hundred launches leave the serial chain of a step. Opt-in: gradients must not be read (or
# accumulated into an existing .grad) before the scope has been left.
_OVERLAP = {"on": False, "streams": None, "i": 0, "used": []}


class overlap_weight_grads:
    def __init__(self, n_streams=2):
        self.n = int(n_streams)

    def __enter__(self):
        if _OVERLAP["streams"] is None or len(_OVERLAP["streams"]) != self.n:
            _OVERLAP["streams"] = [torch.cuda.Stream() for _ in range(self.n)]
        _OVERLAP.update(on=True, i=0, used=[])
        return self

    def __exit__(self, *exc):
        cur = torch.cuda.current_stream()
        for st in _OVERLAP["used"]:
            cur.wait_stream(st)
        _OVERLAP.update(on=False, used=[])
        return False


def _gemm_off_chain(A, B, **kw):
    """gemm() whose result nothing on the current stream reads before the overlap scope is left (a weight gradient)."""
    if not _OVERLAP["on"]:
        return gemm(A, B, **kw)
    cur = torch.cuda.current_stream()
    side = _OVERLAP["streams"][_OVERLAP["i"] % len(_OVERLAP["streams"])]
    _OVERLAP["i"] += 1
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = gemm(A, B, **kw)
    for t in (A, B):
        t.record_stream(side)          # their memory must not be recycled by the current stream while the side product runs
    out.record_stream(cur)
    if side not in _OVERLAP["used"]:
        _OVERLAP["used"].append(side)
    return out


# ---- all weight gradients of a backward pass in (at most) two launches ------------------------------------
# `with defer_weight_grads(): loss.backward()`: every dW product of the pass is only RECORDED (its zero-initialised
# result tensor is handed to autograd at once) and the whole list runs as one grouped launch per tile shape when the
# scope ends (mvk_gemm_f32_tn_grouped): ~100 latency-bound ~10 us launches leave the serial chain of a step. Same
# contract as overlap_weight_grads: nothing may read a weight gradient before the scope has been left.
_DEFER = {"on": False, "items": [], "leaves": set(), "slots": None, "eager": 0}
_DW_MAX = 512


def _defer_slots(device):
    """Pinned host + device table buffers, allocated outside any capture (pinning is not capturable): two rotate for
    eager passes, every captured pass takes one of its own (its memcpy node re-reads the pinned bytes at each replay)."""
    if _DEFER["slots"] is None:
        nb = _DW_MAX * int(lib().mvk_gemm_group_entry_bytes())
        _DEFER["slots"] = [{"host": torch.empty(nb, dtype=torch.uint8).pin_memory(),
                            "dev": torch.empty(nb, dtype=torch.uint8, device=device), "event": None} for _ in range(34)]
    return _DEFER["slots"]


def capture_table_slots_left(device):
    """Captured passes of defer_weight_grads this process can still record (each keeps one pre-pinned table for good; two
    stay with the eager passes). Ask BEFORE starting a capture: running out inside one raises in the middle of the capture."""
    return max(0, len(_defer_slots(device)) - 2)


class defer_weight_grads:
    """`with defer_weight_grads(): loss.backward()`. flush=False leaves the recorded products to the caller:
    `items = scope.take()` after the block, then `flush_deferred(items, stream)` -- e.g. on a side stream, beside the
    next stage of the backward (bench.py: the weight gradients and the optimiser step of everything above the backward
    cut run while the backward of the point-heavy levels is still computing)."""

    def __init__(self, flush=True):
        self.flush = bool(flush)
        self.items = []

    def __enter__(self):
        _DEFER.update(on=True, items=[], leaves=set())
        return self

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                if self.flush:
                    _flush_deferred()
                else:
                    self.items = _DEFER["items"]
        finally:
            _DEFER.update(on=False, items=[], leaves=set())
        return False

    def take(self):
        items, self.items = self.items, []
        return items


def flush_deferred(items, stream=None):
    """The grouped launch for `items` (from defer_weight_grads(flush=False).take()) on `stream` (default: the current
    one). On another stream the operands are marked as used there (record_stream), so their memory is not handed out
    again to work of the producing stream while the grouped product still reads it; the caller orders the streams
    (stream.wait_stream(producer) before, consumer.wait_stream(stream) after)."""
    if not items:
        return
    if stream is None or stream == torch.cuda.current_stream():
        return _flush_deferred(items)
    for (A, B, _, _, _) in items:
        A.record_stream(stream)
        B.record_stream(stream)
    with torch.cuda.stream(stream):
        _flush_deferred(items)


_HAS_USE_COUNT = hasattr(torch._C, "_storage_Use_Count")


def _storage_use_count(storage):
    """Owners of an untyped storage (private torch API, present in the torch 2.x builds this was written against,
    INTEGRATION.md). Without it the deferred mode is not offered at all (_dw_gemm runs every product in line)."""
    return torch._C._storage_Use_Count(storage._cdata)


def _flush_deferred(items=None):
    if items is None:
        items = _DEFER["items"]
    if not items:
        return
    import numpy as np
    dev = items[0][0].device
    n = len(items)
    prob = np.zeros(n, dtype=np.dtype([("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("M", "<i8"), ("N", "<i8"), ("Kd", "<i8")]))
    for i, (A, B, out_ptr, _, storage) in enumerate(items):
        if _storage_use_count(storage) < 2:
            # nothing but this list still owns the result: autograd did not adopt it as a .grad (it added the not yet
            # computed tensor to an existing gradient, or dropped it) -- the product would be lost silently
            raise RuntimeError("defer_weight_grads: a deferred weight-gradient result (%d x %d) was consumed before the "
                               "grouped launch computed it; was the backward run on parameters that already hold a "
                               ".grad, or with hooks that copy gradients? Run it without the scope (MVK_DEFER_DW=0)"
                               % (A.shape[1], B.shape[1]))
        prob[i] = (A.data_ptr(), B.data_ptr(), out_ptr, A.shape[1], B.shape[1], A.shape[0])
    slots = _defer_slots(dev)
    if torch.cuda.is_current_stream_capturing():
        if len(slots) <= 2:
            raise RuntimeError("defer_weight_grads: out of pre-pinned table slots for graph captures")
        slot = slots.pop()              # owned by this capture from now on
        _DEFER.setdefault("captured", []).append(slot)
    else:
        slot = slots[_DEFER["eager"] % 2]
        _DEFER["eager"] += 1
        if slot["event"] is not None:
            slot["event"].synchronize()
    nn, wn, ww = C.c_int(0), C.c_int64(0), C.c_int64(0)
    splits = np.zeros(n, np.int32)
    check(lib().mvk_gemm_f32_tn_grouped_plan(prob.ctypes.data_as(C.c_void_p), n, C.c_void_p(slot["host"].data_ptr()),
                                             C.byref(nn), C.byref(wn), C.byref(ww), splits.ctypes.data_as(C.c_void_p),
                                             _stream()))
    for i, item in enumerate(items):
        if splits[i] > 1 and not item[3] and not lib().mvk_gemm_split_ordered():
            raise RuntimeError("defer_weight_grads: the grouped plan splits a product whose output was not zero-initialised")
    nb = n * int(lib().mvk_gemm_group_entry_bytes())
    slot["dev"][:nb].copy_(slot["host"][:nb], non_blocking=True)
    check(lib().mvk_gemm_f32_tn_grouped(C.c_void_p(slot["dev"].data_ptr()), n, nn.value, wn.value, ww.value, _stream()))
    if not torch.cuda.is_current_stream_capturing():
        ev = torch.cuda.Event()
        ev.record()
        slot["event"] = ev


def _accumulate_in_place_ok(t):
    """Whether a backward node may ADD onto its incoming gradient `t` in place (the fan-out sums of _LinearFn /
    _MaxPoolFn). Autograd hands the SAME tensor to every consumer of an addition's gradient (AddBackward, and
    _AddLReLUFn before it returned copies), so `t` may also be an operand of a weight-gradient product that has only
    been RECORDED (defer_weight_grads) or enqueued on a side stream (overlap_weight_grads): mutating it first would
    corrupt that product silently. Not provably exclusive -> the caller computes `product + t` instead."""
    if _OVERLAP["on"]:
        return False
    if _DEFER["on"] and _DEFER["items"]:
        st = t.untyped_storage().data_ptr()
        for it in _DEFER["items"]:
            if it[0].untyped_storage().data_ptr() == st or it[1].untyped_storage().data_ptr() == st:
                return False
    return True


def _dw_gemm(A, B, transB=False, target=None):
    """A^T @ B for a weight gradient: deferred into the grouped launch inside defer_weight_grads(), on a side stream
    inside overlap_weight_grads(), a plain product otherwise. `target`: the tensor the gradient is FOR (the Function's
    weight input). A deferred product hands autograd a result that is only computed when the scope ends, which is
    sound only if AccumulateGrad ADOPTS that tensor as the parameter's .grad: a parameter that already holds a gradient
    (zero_grad(set_to_none=False), micro-batch accumulation, a second backward stage through the same weight) would
    get `grad += <not yet computed>` and the result tensor would be freed before the grouped launch writes it -- such
    products run in line. A weight used twice inside ONE scope cannot be served either way and is refused."""
    leaf = target if (target is not None and target.is_leaf) else None
    if _DEFER["on"] and leaf is not None:
        if id(leaf) in _DEFER["leaves"]:
            raise RuntimeError("defer_weight_grads: the same parameter receives two weight gradients inside one scope "
                               "(shared weights): run this backward without the scope (MVK_DEFER_DW=0)")
        if leaf.grad is not None:
            return _gemm_off_chain(A, B, transA=True, transB=transB)
    if _DEFER["on"] and _HAS_USE_COUNT and not transB and A.is_cuda and B.shape[1] > 16 and len(_DEFER["items"]) < _DW_MAX \
            and A.shape[0] > 0 and A.shape[1] > 0:
        A, B = _f32c(A), _f32c(B)
        # a split reduction accumulates into a zero-initialised output (arena slice); an unsplit one (the big
        # coarse-level weights: few rows to reduce over, 2/3 of all weight-gradient bytes) writes every element
        split_arena_prepare(A.device)
        zeroed = lib().mvk_gemm_f32_tn_grouped_split(A.shape[1], B.shape[1], A.shape[0]) > 1 \
            and not lib().mvk_gemm_split_ordered()
        out = (_zeros((A.shape[1], B.shape[1]), A.device) if zeroed
               else torch.empty((A.shape[1], B.shape[1]), device=A.device, dtype=torch.float32))
        # only the ADDRESS is recorded: a second reference to the tensor would make autograd's AccumulateGrad clone the
        # gradient instead of adopting it (one copy launch per parameter); the tensor itself lives on as the .grad
        # the STORAGE is held too (not a tensor: AccumulateGrad counts tensor references only): its memory cannot be
        # handed out again before the grouped launch has written it, and its owner count tells at flush time whether
        # autograd adopted the result (>= 2) or consumed and dropped it (1)
        _DEFER["items"].append((A, B, out.data_ptr(), zeroed, out.untyped_storage()))
        if leaf is not None:
            _DEFER["leaves"].add(id(leaf))
        return out
    return _gemm_off_chain(A, B, transA=True, transB=transB)


# --------------------------------------------------------------------------------------------
# KPConv (rigid and deformable) as one autograd node
# --------------------------------------------------------------------------------------------

class _KPConvFn(torch.autograd.Function):
    """y = KPConv(q, s, idx, x; kp, W [, offsets, modulations]); also returns min_d2 when deformable.

    Mirrors the tensor algebra of the reference's KPConv.forward
    (KPConv-PyTorch/models/blocks.py:277-374) and its autograd backward."""

    @staticmethod
    def forward(ctx, q, s, idx, x, kp, W, offsets, modulations, extent, influence, aggregation, stats_n_valid=None,
                order=None, rev=None, rev_order=None):
        K, Cin, Cout = W.shape
        deform = offsets is not None
        A, min_d2 = kpconv_gather(q, s, idx, x, kp, extent, influence, aggregation, offsets, want_min_d2=deform,
                                  order=order)
        ctx.rev = (rev, rev_order)
        Am = A * modulations.unsqueeze(2) if modulations is not None else A     # blocks.py:366-367
        if _PROF["on"]:
            e0, e1 = _timing_events()
        st = None
        if _contraction_streams(Am, q.shape[0], K * Cin, Cout):                 # the big rigid layers (levels 0-1)
            y, st = gemm_f32_stream(Am.view(-1, K * Cin), W.reshape(K * Cin, Cout), stats_n_valid)
        elif stats_n_valid is not None:                                         # blocks.py:370-374
            y, st = gemm(Am.view(-1, K * Cin), W.reshape(K * Cin, Cout), stats_n_valid=stats_n_valid)
        else:
            y = gemm(Am.view(-1, K * Cin), W.reshape(K * Cin, Cout))
        if _PROF["on"]:
            e1.record()
            _PROF["gemm"].append(((q.shape[0], K * Cin, Cout), e0, e1))
        ctx.save_for_backward(q, s, idx, x, kp, W, A, offsets, modulations)
        ctx.min_arg = getattr(min_d2, "_mvk_min_arg", None)
        ctx.cfg = (extent, influence, aggregation)
        ctx.stat_rows = st[1] if st is not None else 0
        _LAST_STATS_ROWS[0] = ctx.stat_rows
        part = st[0] if st is not None else None
        if part is not None:
            ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)    # else autograd hands backward a ZERO tensor (one fill launch) per unused output
        return y, min_d2, part

    @staticmethod
    def backward(ctx, gy, g_min_d2, g_part=None):
        q, s, idx, x, kp, W, A, offsets, modulations = ctx.saved_tensors
        extent, influence, aggregation = ctx.cfg
        K, Cin, Cout = W.shape
        Nq = q.shape[0]
        if gy is None:                      # only min_d2 was used (the regulariser): no feature gradient
            gy = torch.zeros((Nq, Cout), device=q.device, dtype=torch.float32)
        gy = _f32c(gy)
        Am = A * modulations.unsqueeze(2) if modulations is not None else A
        dW = dx = d_off = d_mod = None
        if ctx.needs_input_grad[5]:
            dW = _dw_gemm(Am.view(Nq, K * Cin), gy, target=W).view(K, Cin, Cout)
        need_dA = ctx.needs_input_grad[3] or (offsets is not None)
        rev, rev_order = ctx.rev
        gather_form = (need_dA and rev is not None and REVERSE_DX and rev.shape[0] >= s.shape[0] and Cout >= 5
                       and influence == "linear" and aggregation == "sum")
        if gather_form and offsets is None and modulations is None and (Cout >= 32 or REVERSE_DX_ANY_COUT):
            # gather form (csrc/revlist.hip): the forward kernel over the transposed neighbourhood relation with the
            # kernel points negated, then the per-kernel-point transposed contraction -- no atomics, fixed summation order
            if rev.shape[0] != s.shape[0]:
                rev = rev[:s.shape[0]]
            A2, _ = kpconv_gather(s, q, rev, gy, _neg_kernel_points(kp), extent, influence, aggregation, order=rev_order,
                                  tag="[dx]")
            dx = kp_transposed_contraction(A2, W)
        elif gather_form and offsets is not None and REVERSE_DX_DEFORM:
            # deformable layer (round 5): the same gather with the kernel points -- and modulations -- of the NEIGHBOUR rows
            # (mvk_kpconv_gather_rev_deform); dA is still needed, by the offset gradient and the modulation gradient
            if rev.shape[0] != s.shape[0]:
                rev = rev[:s.shape[0]]
            dAm = gemm(gy, W.reshape(K * Cin, Cout), transB=True).view(Nq, K, Cin)
            if modulations is not None:
                if ctx.needs_input_grad[7]:
                    d_mod = (dAm * A).sum(dim=2)
                dA = dAm * modulations.unsqueeze(2)
            else:
                dA = dAm
            d_off = kpconv_deform_doff(q, s, idx, x, kp, extent, offsets, dA, g_min_d2, ctx.min_arg)
            if ctx.needs_input_grad[3]:
                A2 = kpconv_gather_rev_deform(s, q, rev, gy, kp, extent, offsets, modulations, order=rev_order)
                dx = kp_transposed_contraction(A2, W)
        elif need_dA:
            dAm = gemm(gy, W.reshape(K * Cin, Cout), transB=True).view(Nq, K, Cin)
            if modulations is not None:
                if ctx.needs_input_grad[7]:
                    d_mod = (dAm * A).sum(dim=2)
                dA = dAm * modulations.unsqueeze(2)
            else:
                dA = dAm
            dx, d_off = kpconv_scatter(q, s, idx, dA, kp, extent, influence, aggregation, x=x,
                                       offsets=offsets,
                                       g_min_d2=g_min_d2 if offsets is not None else None, min_arg=ctx.min_arg)
        return None, None, None, dx, None, dW, d_off, d_mod, None, None, None, None, None, None, None


_LAST_STATS_ROWS = [0]       # rows per block of the statistics partials the last KPConv forward produced


def _contraction_streams(A, M, Kd, N):
    """Whether the forward contraction of a layer runs on gemm_f32_stream: a supported shape and an aggregate with the
    slack the kernel reads behind its last row (kpconv_gather's own allocation has it; a modulated copy does not)."""
    if M <= 0:
        return False
    ok, _, _, need = gemm_f32_stream_plan(M, N, Kd)
    return ok and A.is_contiguous() and _slack_floats(A) >= need


def bn_finished(t):
    """True when t carries BatchNorm statistics FINISHED by its producer (mean, invstd; see _fin_request)."""
    e = getattr(t, "_mvk_bn_stats", None)
    return e is not None and len(e) > 2 and e[2] is not None


def bn_stats_of(t):
    """(partials, rows per block) a producing GEMM attached to its output for the BatchNorm that follows, or None."""
    return getattr(t, "_mvk_bn_stats", None)


def kpconv(q, s, idx, x, kp, W, extent, influence="linear", aggregation="sum", offsets=None, modulations=None,
           stats_n_valid=None, order=None, rev=None, rev_order=None, bn=None):
    """Returns (y [Nq,Cout], min_d2 [Nq,K] or None). stats_n_valid (DEVICE int32 [1]): the contraction also produces the column statistics
    of y over its first n_valid rows for the BatchNorm that follows (picked up by bn_lrelu via bn_stats_of).
    order (int32 [Nq], a permutation, e.g. neighbors_cell_order of the query level): the order the f32 gather works
    through the query points in (kpconv_gather); the layer's result does not depend on it.
    rev (int32 [Ns, Hr], reverse_neighbors(idx, Ns)) and rev_order (a work list of the SUPPORT level): rigid f32 layers
    with a power-of-two Cout >= 32 then compute the feature gradient as a gather over the transposed relation (fixed
    summation order, no atomics) instead of the atomic scatter.
    bn: the nn.BatchNorm1d that normalises y; its statistics are then FINISHED by the contraction (see _fin_request)."""
    if rev is not None and (rev.dtype != torch.int32 or rev.dim() != 2 or not rev.is_contiguous()):
        raise RuntimeError("kpconv: rev must be a contiguous int32 [Ns, Hr] matrix (ops.reverse_neighbors)")
    if influence not in INFLUENCE:
        raise ValueError("Unknown influence function type (config.KP_influence)")
    if aggregation not in AGGREGATION:
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    _fin_request(bn if stats_n_valid is not None else None)
    try:
        y, min_d2, part = _KPConvFn.apply(q, s, idx, x, kp, W, offsets, modulations, float(extent), influence, aggregation,
                                   stats_n_valid, order, rev, rev_order)
    finally:
        fin = _fin_take()
    if part is not None:        # rows per statistics block of the launch that just ran (set by the node's forward)
        rows = _LAST_STATS_ROWS[0]
        y._mvk_bn_stats = (part, rows, fin)
    return y, min_d2


# --------------------------------------------------------------------------------------------
# offset regulariser of the deformable layers
# --------------------------------------------------------------------------------------------

class _DeformRegFn(torch.autograd.Function):
    """p2p_fitting_regularizer of one deformable KPConv (models/architectures.py:20-58) as one launch that also
    produces both gradients (mvk_deform_regularizer); the backward only scales them by the upstream gradient."""

    @staticmethod
    def forward(ctx, min_d2, deformed_kp, n_valid, extent, repulse_extent, power):
        _dev(min_d2, deformed_kp, n_valid)
        min_d2, dkp = _f32c(min_d2), _f32c(deformed_kp)
        N, K = min_d2.shape
        loss = _zeros((1,), min_d2.device)
        d_min = torch.empty_like(min_d2)
        d_dkp = torch.empty_like(dkp)
        check(lib().mvk_deform_regularizer(_p(min_d2), _p(dkp), _p(n_valid), N, K, float(extent), float(repulse_extent),
                                           float(power), _p(loss), _p(d_min), _p(d_dkp), _stream()))
        ctx.save_for_backward(d_min, d_dkp)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d_min, d_dkp = ctx.saved_tensors
        return d_min * g, d_dkp * g, None, None, None, None


class _DeformOperandsFn(torch.autograd.Function):
    """offset_features, offsets, deformed_KP (and modulations) of a deformable KPConv from its inner convolution's
    output (models/blocks.py:243-266, :287), one launch forward and one backward (mvk_deform_operands_fwd/bwd)."""

    @staticmethod
    def forward(ctx, raw, bias, kernel_points, extent, modulated):
        _dev(raw, bias, kernel_points)
        raw, bias, kp = _f32c(raw), _f32c(bias), _f32c(kernel_points)
        N, D = raw.shape
        K = kp.shape[0]
        if D != (4 if modulated else 3) * K or bias.numel() != D:
            raise RuntimeError("deform operands: the inner convolution must deliver %d columns" % ((4 if modulated else 3) * K))
        feat = torch.empty_like(raw)
        offsets = torch.empty((N, K, 3), device=raw.device, dtype=torch.float32)
        dkp = torch.empty_like(offsets)
        mod = torch.empty((N, K), device=raw.device, dtype=torch.float32) if modulated else None
        check(lib().mvk_deform_operands_fwd(_p(raw), _p(bias), _p(kp), N, K, int(bool(modulated)), float(extent), _p(feat),
                                            _p(offsets), _p(dkp), _p(mod), _stream()))
        ctx.save_for_backward(mod)
        ctx.cfg = (N, K, D, float(extent), bool(modulated))
        ctx.set_materialize_grads(False)
        return feat, offsets, dkp, mod

    @staticmethod
    def backward(ctx, g_feat, g_off, g_dkp, g_mod):
        (mod,) = ctx.saved_tensors
        N, K, D, extent, modulated = ctx.cfg
        dev = (mod if mod is not None else next(g for g in (g_feat, g_off, g_dkp, g_mod) if g is not None)).device
        gs = [None if g is None else _f32c(g) for g in (g_off, g_dkp, g_mod, g_feat)]
        d_raw = torch.empty((N, D), device=dev, dtype=torch.float32)
        d_bias = _zeros((D,), dev)
        check(lib().mvk_deform_operands_bwd(_p(gs[0]), _p(gs[1]), _p(gs[2]), _p(mod), _p(gs[3]), N, K, int(modulated), extent,
                                            _p(d_raw), _p(d_bias), _stream()))
        return d_raw, d_bias, None, None, None


def deform_operands(raw, bias, kernel_points, extent, modulated=False):
    """(offset_features [N,D], offsets [N,K,3], deformed_KP [N,K,3], modulations [N,K] or None) from the inner
    convolution's output raw [N,D] (D = 3K, 4K when modulated)."""
    return _DeformOperandsFn.apply(raw, bias, kernel_points, extent, modulated)


def _reg_many(cfgs, ts, grads, loss, gscale):
    """mvk_deform_regularizer_many over the layers (min_d2, deformed_kp) = ts[2i], ts[2i+1], REG_MANY per launch (layers
    of another K get their own launch)."""
    from ._lib import RegLayer
    by_k = {}
    for i, (extent, repulse, power, nv) in enumerate(cfgs):
        m, d = ts[2 * i], ts[2 * i + 1]
        L = RegLayer(_p(m), _p(d), _p(nv), _p(grads[2 * i]) if grads else None, _p(grads[2 * i + 1]) if grads else None,
                     m.shape[0], float(extent), float(repulse), float(power))
        by_k.setdefault(int(m.shape[1]), []).append(L)
    for K, layers in by_k.items():
        for b in range(0, len(layers), REG_MANY):
            part = layers[b:b + REG_MANY]
            arr = (RegLayer * len(part))(*part)
            check(lib().mvk_deform_regularizer_many(arr, len(part), K, _p(loss), _p(gscale), _stream()))


class _DeformRegAllFn(torch.autograd.Function):
    """The regulariser terms of ALL deformable layers of a network as one autograd node: every layer's forward launch
    accumulates into one scalar (no additions of per-layer losses), the backward launches the same kernel with the
    upstream gradient as a device scalar and gets both gradients already scaled (no multiplications)."""

    @staticmethod
    def forward(ctx, cfgs, *tensors):
        ts = [_f32c(t) for t in tensors]
        _dev(*ts)
        loss = _zeros((1,), ts[0].device)
        _reg_many(cfgs, ts, None, loss, None)
        ctx.save_for_backward(*ts)
        ctx.cfgs = cfgs
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        ts = ctx.saved_tensors
        g = _f32c(g).reshape(1)
        grads = [torch.empty_like(t) for t in ts]
        _reg_many(ctx.cfgs, ts, grads, None, g)
        return (None, *grads)


def deform_regularizer_all(layers):
    """Sum of the regulariser terms of several deformable layers; layers: list of (min_d2 [N,K], deformed_kp [N,K,3],
    extent, repulse_extent, power, n_valid or None). Layers without rows contribute nothing."""
    layers = [l for l in layers if l[0].shape[0] > 0]
    if not layers:
        return None
    cfgs = [(l[2], l[3], l[4], l[5]) for l in layers]
    flat = []
    for l in layers:
        flat += [l[0], l[1]]
    return _DeformRegAllFn.apply(cfgs, *flat)


def deform_regularizer(min_d2, deformed_kp, extent, repulse_extent, power=1.0, n_valid=None):
    """power * (2 * fitting + repulsive) of one deformable layer; min_d2 [N,K], deformed_kp [N,K,3]; n_valid: DEVICE
    int32 [1] row count of a capacity-padded level (means over the valid rows only) or None."""
    return _DeformRegFn.apply(min_d2, deformed_kp, n_valid, extent, repulse_extent, power)


# --------------------------------------------------------------------------------------------
# masked BatchNorm + LeakyReLU (capacity-padded levels, hipGraph replay)
# --------------------------------------------------------------------------------------------

# capacity-padded mode: {row capacity of a level: DEVICE int32[1] tensor holding the valid row count}.
# Set by the network wrapper around a forward pass (see synthetic.CapacityBatch); empty = plain mode.
_ROW_COUNTS = {}


def set_row_counts(mapping):
    _ROW_COUNTS.clear()
    if mapping:
        _ROW_COUNTS.update(mapping)


def row_count_for(rows):
    return _ROW_COUNTS.get(int(rows))


_COUNT_TABLE = {}                    # device index -> int32 [0, 1, 2, ...]: word r of it holds the value r, for good
_COUNT_TABLE_ROWS = 1 << 22
_BIG_COUNTS = {}                     # (rows, device index) beyond the table: one word each, never freed


def full_count(rows, device):
    """Device int32 [1] holding `rows`: the n_valid of a tensor without padded rows. A view of word `rows` of a constant
    table [0, 1, 2, ...] (16 MB per device, written ONCE by an eager launch): no fill launch per call, no
    host-to-device copy, nothing to evict, and the same word is valid in eager code and in every captured graph -- the
    row count of every pyramid level changes with each batch in ordinary training. The table cannot be created while a
    stream is capturing (its fill would become a node of that graph and the words would be undefined for everyone
    else until that graph has replayed): any eager step before the capture creates it, or full_count_prepare()."""
    rows = int(rows)
    t = _COUNT_TABLE.get(device.index)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ops.full_count: the constant row-count table does not exist yet and cannot be created "
                               "inside a graph capture; run one eager step first or call ops.full_count_prepare(device)")
        t = torch.arange(_COUNT_TABLE_ROWS, dtype=torch.int32, device=device)
        torch.cuda.current_stream(device).synchronize()      # other streams read views of it without any event
        _COUNT_TABLE[device.index] = t
    if 0 <= rows < _COUNT_TABLE_ROWS:
        return t[rows:rows + 1]
    key = (rows, device.index)
    w = _BIG_COUNTS.get(key)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ops.full_count: %d rows lie beyond the constant table; request this count once outside "
                               "the capture" % rows)
        w = torch.full((1,), rows, dtype=torch.int32, device=device)
        torch.cuda.current_stream(device).synchronize()
        _BIG_COUNTS[key] = w
    return w


def full_count_prepare(device):
    """Creates the constant table of full_count() (call before capturing a graph that was never run eagerly)."""
    full_count(0, torch.device(device))


def _ext3(ext):
    """(partials, rows per block, finished (mean, invstd) or None) of a `_mvk_bn_stats` record (None: nothing attached)."""
    if ext is None:
        return None, 0, None
    return (ext[0], ext[1], ext[2] if len(ext) > 2 else None)


class _BNLReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_valid, gamma, beta, running_mean, running_var, eps, momentum, slope, training, nbt=None,
                addend=None, ext=None):
        _dev(x, n_valid, gamma, beta, addend)
        x = _f32c(x)
        if addend is not None:
            addend = _f32c(addend)
            if addend.shape != x.shape:
                raise RuntimeError("bn_lrelu: the residual addend must have the shape of the input")
        R, D = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(D, device=x.device, dtype=torch.float32)
        invstd = torch.empty(D, device=x.device, dtype=torch.float32)
        if not training:
            raise RuntimeError("masked BatchNorm is a training-mode op; use nn.BatchNorm1d in eval mode")
        ext_part, ext_rows, fin = _ext3(ext)
        if ext_part is not None and (ext_part.shape[2] != D or ext_part.shape[0] != (R + ext_rows - 1) // ext_rows):
            raise RuntimeError("bn_lrelu: the statistics partials do not belong to this tensor")
        if fin is not None:
            # statistics finished by the producing product (mean, invstd, running statistics, batch counter): apply only
            mean, invstd = fin
            check(lib().mvk_bn_lrelu_fwd(_p(x), _p(n_valid), R, D, _p(gamma), _p(beta), float(eps), float(momentum),
                                         float(slope), None, None, _p(mean), _p(invstd), None, _p(y), None, _p(addend),
                                         None, -1, _stream()))
        else:
            scratch = None if ext_part is not None else torch.empty(((R + 63) // 64) * 2 * D, device=x.device,
                                                                    dtype=torch.float32)
            check(lib().mvk_bn_lrelu_fwd(_p(x), _p(n_valid), R, D, _p(gamma), _p(beta), float(eps), float(momentum),
                                         float(slope), _p(running_mean), _p(running_var), _p(mean), _p(invstd),
                                         _p(scratch), _p(y), _p(nbt), _p(addend), _p(ext_part), int(ext_rows), _stream()))
        ctx.save_for_backward(x, n_valid, gamma, beta, mean, invstd, y if addend is not None else None)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, g):
        x, n_valid, gamma, beta, mean, invstd, yout = ctx.saved_tensors
        g = _f32c(g)
        R, D = x.shape
        dgb = torch.empty(2 * D, device=x.device, dtype=torch.float32)
        scratch = torch.empty(((R + 63) // 64) * 2 * D, device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        d_add = torch.empty_like(x) if yout is not None else None
        check(lib().mvk_bn_lrelu_bwd(_p(x), _p(g), _p(n_valid), R, D, _p(gamma), _p(beta), _p(mean), _p(invstd),
                                     ctx.slope, _p(scratch), _p(dgb), _p(dx), _p(yout), _p(d_add), _stream()))
        return dx, None, dgb[D:], dgb[:D], None, None, None, None, None, None, None, d_add, None


# ---- synchronised BatchNorm (opt-in; SURVEY.md 8e, models/blocks.py:453-460) --------------------------------------
# The reference normalises over the stacked point axis of the WHOLE batch. Sharding the spheres over ranks leaves every
# rank the statistics of its own spheres (the default, a documented deviation); with set_sync_batchnorm(group) the
# statistics of every BatchNorm -- blocks and FeatureAggregation alike -- are those of all ranks' rows: N ranks x S
# spheres then compute what one rank x N*S spheres computes (tests/test_model_gpu.py, two ranks). Written with
# differentiable tensor ops and torch.distributed.nn's differentiable all-reduce (two all-reduces of C floats forward,
# their mirror images backward): an option for runs that must match a single-GPU run, not a tuned path.
_SYNC_BN = {"group": None}


def set_sync_batchnorm(group):
    """group: a torch.distributed process group, True (the default group) or None (per-rank statistics: the default)."""
    if group is True:
        import torch.distributed as dist
        group = dist.group.WORLD
    _SYNC_BN["group"] = group


def _sync_bn_lrelu(x, n_valid, bn, slope, addend):
    import torch.distributed as dist
    from torch.distributed.nn.functional import all_reduce
    group = _SYNC_BN["group"]
    if not bn.training:
        raise RuntimeError("synchronised BatchNorm is a training-mode op; use nn.BatchNorm1d in eval mode")
    x = _f32c(x)
    R, D = x.shape
    mask = (torch.arange(R, device=x.device) < n_valid.to(torch.int64)).to(x.dtype).unsqueeze(1)     # valid rows
    n = all_reduce(n_valid.to(x.dtype).reshape(1), op=dist.ReduceOp.SUM, group=group)
    mean = all_reduce((x * mask).sum(0), op=dist.ReduceOp.SUM, group=group) / n
    d = (x - mean) * mask
    var = all_reduce((d * d).sum(0), op=dist.ReduceOp.SUM, group=group) / n                          # biased, about the GLOBAL mean
    y = d * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias
    if addend is not None:
        y = y + addend
    y = torch.nn.functional.leaky_relu(y, slope) if slope != 1.0 else y
    y = y * mask
    if bn.track_running_stats:
        with torch.no_grad():
            m = bn.momentum if bn.momentum is not None else 0.0
            bn.running_mean.mul_(1 - m).add_(m * mean.detach())
            bn.running_var.mul_(1 - m).add_(m * var.detach() * (n / (n - 1).clamp(min=1)))
            bn.num_batches_tracked += 1
    return y


def bn_lrelu(x, n_valid, bn, slope=1.0, addend=None):
    """y = LeakyReLU_slope(BatchNorm1d(x[:n_valid]) [+ addend]) with rows >= n_valid zeroed; `bn` is an
    nn.BatchNorm1d whose parameters / running statistics are used and updated; n_valid is a DEVICE
    int32 tensor of one element. addend [R,D]: the shortcut of a residual block, joined before the
    activation (blocks.py:649) inside the same launch; it receives its own gradient."""
    if _SYNC_BN["group"] is not None:
        return _sync_bn_lrelu(x, n_valid, bn, slope, addend)
    nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None   # += 1 inside the kernel
    return _BNLReLUFn.apply(x, n_valid, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                            bn.momentum if bn.momentum is not None else 0.0, slope, bn.training, nbt, addend,
                            bn_stats_of(x))


class _BNLazyFn(torch.autograd.Function):
    """y = LeakyReLU_slope(BatchNorm(x)) with FINISHED statistics whose forward launches nothing: the product that consumes
    y applies the transform while it stages its A operand and writes y on the way (mvk_a_transform, bn_lrelu_linear).
    The backward is the masked BatchNorm's own (_BNLReLUFn.backward without an addend)."""

    @staticmethod
    def forward(ctx, x, n_valid, gamma, beta, mean, invstd, slope):
        y = torch.empty_like(x)
        ctx.save_for_backward(x, n_valid, gamma, beta, mean, invstd)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, g):
        x, n_valid, gamma, beta, mean, invstd = ctx.saved_tensors
        g = _f32c(g)
        R, D = x.shape
        dgb = torch.empty(2 * D, device=x.device, dtype=torch.float32)
        scratch = torch.empty(((R + 63) // 64) * 2 * D, device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        check(lib().mvk_bn_lrelu_bwd(_p(x), _p(g), _p(n_valid), R, D, _p(gamma), _p(beta), _p(mean), _p(invstd),
                                     ctx.slope, _p(scratch), _p(dgb), _p(dx), None, None, _stream()))
        return dx, None, dgb[D:], dgb[:D], None, None, None


XF_KMAX = 512        # csrc/gemm.hip: longest reduction a product with an operand transform takes


def bn_lrelu_linear(x, n_valid, bn, slope, W, stats_n_valid=None, bn_out=None):
    """linear(bn_lrelu(x, n_valid, bn, slope), W, stats_n_valid=..., bn=bn_out) with the BatchNorm's apply pass folded
    into the product's operand load (blocks.py:639-644: batch_norm_conv + LeakyReLU + unary2): x must carry statistics
    FINISHED by its producer (kpconv(..., bn=bn) / linear(..., bn=bn)). Returns (y, activations) -- the activations are
    the tensor bn_lrelu would have returned, written by the product -- or None when the fold does not apply (the caller
    then runs the two steps)."""
    ext = bn_stats_of(x)
    fin = ext[2] if (ext is not None and len(ext) > 2) else None
    R, Kd = x.shape
    if (fin is None or not BN_FOLD or _SYNC_BN["group"] is not None or not x.is_cuda or x.dtype != torch.float32
            or not x.is_contiguous() or W.dtype != torch.float32 or not W.is_contiguous() or W.shape[1] != Kd
            or W.shape[0] <= 32 or Kd > XF_KMAX or Kd % 4 != 0 or R == 0 or not slope > 0):
        return None
    mean, invstd = fin
    lazy = _BNLazyFn.apply(x, n_valid, bn.weight, bn.bias, mean, invstd, float(slope))
    _AX["req"] = {"raw": x.detach(), "mean": mean, "invstd": invstd, "gamma": bn.weight.detach(), "beta": bn.bias.detach(),
                  "slope": float(slope), "n_valid": n_valid, "lazy": lazy}
    try:
        y = linear(lazy, W, stats_n_valid=stats_n_valid, bn=bn_out)
    finally:
        pending, _AX["req"] = _AX["req"], None
    if pending is not None:
        raise RuntimeError("bn_lrelu_linear: the product did not take the operand transform")
    return y, lazy


class _BNLReLUPairFn(torch.autograd.Function):
    """Two masked BatchNorm (+ LeakyReLU) problems of the same row count, one launch each way (mvk_bn_lrelu_fwd_pair /
    _bwd_pair): the convolution output and the shortcut of a bottleneck block. No residual addend in either."""

    @staticmethod
    def forward(ctx, xa, xb, n_valid, ga, ba, rma, rva, gb, bb, rmb, rvb, cfg_a, cfg_b, nbt_a, nbt_b, ext_a, ext_b):
        from ._lib import BnFwdProblem
        _dev(xa, xb, n_valid, ga, ba, gb, bb)
        xa, xb = _f32c(xa), _f32c(xb)
        if xa.shape[0] != xb.shape[0]:
            raise RuntimeError("bn_lrelu_pair: the two inputs differ in rows")
        R = xa.shape[0]
        probs, keep = [], []
        for x, g, b, rm, rv, (eps, mom, slope), nbt, ext in ((xa, ga, ba, rma, rva, cfg_a, nbt_a, ext_a),
                                                              (xb, gb, bb, rmb, rvb, cfg_b, nbt_b, ext_b)):
            D = x.shape[1]
            y = torch.empty_like(x)
            mean = torch.empty(D, device=x.device, dtype=torch.float32)
            invstd = torch.empty(D, device=x.device, dtype=torch.float32)
            ext_part, ext_rows, fin = _ext3(ext)
            if ext_part is not None and (ext_part.shape[2] != D or ext_part.shape[0] != (R + ext_rows - 1) // ext_rows):
                raise RuntimeError("bn_lrelu_pair: the statistics partials do not belong to this tensor")
            if fin is not None:       # finished by the producing product: apply only (see _BNLReLUFn)
                mean, invstd = fin
                scratch = None
                probs.append(BnFwdProblem(_p(x), _p(n_valid), R, D, _p(g), _p(b), float(eps), float(mom), float(slope), None,
                                          None, _p(mean), _p(invstd), None, _p(y), None, None, None, -1))
                keep.append((y, mean, invstd, scratch))
                continue
            scratch = None if ext_part is not None else torch.empty(((R + 63) // 64) * 2 * D, device=x.device,
                                                                    dtype=torch.float32)
            probs.append(BnFwdProblem(_p(x), _p(n_valid), R, D, _p(g), _p(b), float(eps), float(mom), float(slope), _p(rm),
                                      _p(rv), _p(mean), _p(invstd), _p(scratch), _p(y), _p(nbt), None, _p(ext_part),
                                      int(ext_rows)))
            keep.append((y, mean, invstd, scratch))
        check(lib().mvk_bn_lrelu_fwd_pair(C.byref(probs[0]), C.byref(probs[1]), _stream()))
        ctx.save_for_backward(xa, xb, n_valid, ga, ba, gb, bb, keep[0][1], keep[0][2], keep[1][1], keep[1][2])
        ctx.slopes = (float(cfg_a[2]), float(cfg_b[2]))
        ctx.set_materialize_grads(False)
        return keep[0][0], keep[1][0]

    @staticmethod
    def backward(ctx, g_a, g_b):
        from ._lib import BnBwdProblem
        xa, xb, n_valid, ga, ba, gb, bb, mean_a, is_a, mean_b, is_b = ctx.saved_tensors
        R = xa.shape[0]
        sets = []
        for x, g, gam, bet, mean, istd, slope in ((xa, g_a, ga, ba, mean_a, is_a, ctx.slopes[0]),
                                                  (xb, g_b, gb, bb, mean_b, is_b, ctx.slopes[1])):
            if g is None:
                g = torch.zeros_like(x)
            g = _f32c(g)
            D = x.shape[1]
            dgb = torch.empty(2 * D, device=x.device, dtype=torch.float32)
            scratch = torch.empty(((R + 63) // 64) * 2 * D, device=x.device, dtype=torch.float32)
            dx = torch.empty_like(x)
            sets.append((BnBwdProblem(_p(x), _p(g), _p(n_valid), R, D, _p(gam), _p(bet), _p(mean), _p(istd), float(slope),
                                      _p(scratch), _p(dgb), _p(dx), None, None), dgb, dx, g, scratch))
        check(lib().mvk_bn_lrelu_bwd_pair(C.byref(sets[0][0]), C.byref(sets[1][0]), _stream()))
        (_, dgb_a, dx_a, _, _), (_, dgb_b, dx_b, _, _) = sets
        Da, Db = xa.shape[1], xb.shape[1]
        return (dx_a, dx_b, None, dgb_a[Da:], dgb_a[:Da], None, None, dgb_b[Db:], dgb_b[:Db], None, None, None, None, None,
                None, None, None)


def bn_lrelu_pair(xa, bn_a, slope_a, xb, bn_b, slope_b, n_valid):
    """(bn_lrelu(xa, n_valid, bn_a, slope_a), bn_lrelu(xb, n_valid, bn_b, slope_b)) as ONE launch each way: two inputs of
    the same row count normalised independently (the convolution and the shortcut of a bottleneck block)."""
    def unpack(bn):
        nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None
        return nbt, (bn.eps, bn.momentum if bn.momentum is not None else 0.0)
    if not (bn_a.training and bn_b.training):
        raise RuntimeError("masked BatchNorm is a training-mode op; use nn.BatchNorm1d in eval mode")
    if _SYNC_BN["group"] is not None:
        return _sync_bn_lrelu(xa, n_valid, bn_a, slope_a, None), _sync_bn_lrelu(xb, n_valid, bn_b, slope_b, None)
    nbt_a, (eps_a, mom_a) = unpack(bn_a)
    nbt_b, (eps_b, mom_b) = unpack(bn_b)
    return _BNLReLUPairFn.apply(xa, xb, n_valid, bn_a.weight, bn_a.bias, bn_a.running_mean, bn_a.running_var,
                                bn_b.weight, bn_b.bias, bn_b.running_mean, bn_b.running_var,
                                (eps_a, mom_a, slope_a), (eps_b, mom_b, slope_b), nbt_a, nbt_b,
                                bn_stats_of(xa), bn_stats_of(xb))


class _AddLReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, slope):
        _dev(a, b)
        a, b = _f32c(a), _f32c(b)
        y = torch.empty_like(a)
        check(lib().mvk_add_lrelu_fwd(_p(a), _p(b), a.numel(), float(slope), _p(y), _stream()))
        ctx.save_for_backward(y)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _f32c(g)
        d = torch.empty_like(y)
        check(lib().mvk_add_lrelu_bwd(_p(y), _p(g), y.numel(), ctx.slope, _p(d), _stream()))
        # two DISTINCT tensors: downstream nodes accumulate onto their incoming gradient in place (_LinearFn, _MaxPoolFn)
        # or record it as the operand of a deferred product -- one buffer for both branches would alias them
        return d, d.clone(), None


def _bias_grad(y):
    """Buffer for the bias gradient of mvk_bias_lrelu_bwd: it becomes the .grad of a leaf parameter, so it is never a
    slice of the per-step zero arena; written whole by the ordered reduction, zero-initialised otherwise."""
    split_arena_prepare(y.device)
    if lib().mvk_gemm_split_ordered():
        return torch.empty((y.shape[1],), device=y.device, dtype=torch.float32)
    return torch.zeros((y.shape[1],), device=y.device, dtype=torch.float32)


class _BiasLReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, slope):
        _dev(x, bias)
        x, bias = _f32c(x), _f32c(bias)
        y = torch.empty_like(x)
        check(lib().mvk_bias_lrelu_fwd(_p(x), _p(bias), x.shape[0], x.shape[1], float(slope), _p(y), _stream()))
        ctx.save_for_backward(y)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _f32c(g)
        dx = torch.empty_like(y)
        db = _bias_grad(y)
        check(lib().mvk_bias_lrelu_bwd(_p(y), _p(g), y.shape[0], y.shape[1], ctx.slope, _p(dx), _p(db), _stream()))
        return dx, db, None


def bias_lrelu(x, bias, slope=0.1):
    """LeakyReLU_slope(x + bias) for [R, C <= 256] rows in one launch each way (blocks.py:462-463 + the block's
    LeakyReLU; slope = 1: the bias alone). slope must be positive (the backward reads the sign from the output)."""
    if x.dim() != 2 or bias.dim() != 1 or bias.shape[0] != x.shape[1]:
        raise RuntimeError("bias_lrelu: x [R,C] and bias [C] expected")
    if not slope > 0:
        raise ValueError("bias_lrelu: slope must be positive")
    return _BiasLReLUFn.apply(x, bias, slope)


class _LinearBiasActFn(torch.autograd.Function):
    """LeakyReLU_slope(x W^T + bias) with the bias and the activation in the GEMM's store (mvk_gemm_f32_bias_act); the
    backward is bias_lrelu's (mask from the saved output, bias gradient in the same launch) followed by the layer's two
    products."""

    @staticmethod
    def forward(ctx, x, W, bias, slope):
        _dev(x, W, bias)
        x, W, bias = _f32c(x), _f32c(W), _f32c(bias)
        y = torch.empty((x.shape[0], W.shape[0]), device=x.device, dtype=torch.float32)
        if x.shape[0] > 0:
            check(lib().mvk_gemm_f32_bias_act(_p(x), _p(W), _p(y), x.shape[0], W.shape[0], x.shape[1], 1, _p(bias),
                                              float(slope), _stream()))
        ctx.save_for_backward(x, W, y)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, y = ctx.saved_tensors
        g = _f32c(g)
        d = torch.empty_like(y)
        db = _bias_grad(y)
        check(lib().mvk_bias_lrelu_bwd(_p(y), _p(g), y.shape[0], y.shape[1], ctx.slope, _p(d), _p(db), _stream()))
        dx = gemm(d, W) if ctx.needs_input_grad[0] else None
        dW = _dw_gemm(d, x, target=W) if ctx.needs_input_grad[1] else None
        return dx, dW, db, None


def linear_bias_lrelu(x, W, bias, slope=0.1):
    """bias_lrelu(linear(x, W), bias, slope) in one launch forward (a BatchNorm-less UnaryBlock: the head layers)."""
    if x.dim() != 2 or bias.dim() != 1 or bias.shape[0] != W.shape[0] or W.shape[1] != x.shape[1]:
        raise RuntimeError("linear_bias_lrelu: x [R,K], W [C,K] and bias [C] expected")
    if not slope > 0:
        raise ValueError("linear_bias_lrelu: slope must be positive")
    return _LinearBiasActFn.apply(x, W, bias, slope)


def add_lrelu(a, b, slope=0.1):
    """LeakyReLU(a + b) in one launch (residual join of ResnetBottleneckBlock, blocks.py:649)."""
    if a.shape != b.shape:
        raise RuntimeError("add_lrelu: shapes differ")
    return _AddLReLUFn.apply(a, b, slope)


# --------------------------------------------------------------------------------------------
# pooling helpers of blocks.py
# --------------------------------------------------------------------------------------------

class _MaxPoolFn(torch.autograd.Function):
    """passthrough (see _LinearFn): a second, aliasing output of x for another consumer of x; the backward scatters onto
    that consumer's gradient (the kernel accumulates with atomics anyway) instead of onto zeros + one more add."""

    @staticmethod
    def forward(ctx, x, inds, passthrough=False):
        _dev(x, inds)
        x = _f32c(x)
        ctx.rev = reverse_for(inds) if is_deterministic() else None       # (by the batch's own tensor object)
        inds, i64 = _idx(inds)
        Nq, H = inds.shape
        out = torch.empty((Nq, x.shape[1]), device=x.device, dtype=torch.float32)
        arg = torch.empty((Nq, x.shape[1]), device=x.device, dtype=torch.int32)
        check(lib().mvk_max_pool_fwd(_p(x), x.shape[0], x.shape[1], _p(inds), i64, Nq, H, _p(out), _p(arg), _stream()))
        ctx.save_for_backward(inds, arg)
        ctx.ns, ctx.c = x.shape[0], x.shape[1]
        ctx.set_materialize_grads(False)
        return out, (x.view_as(x) if passthrough else None)

    @staticmethod
    def backward(ctx, g, g_alias=None):
        inds, arg = ctx.saved_tensors
        if g is None:
            return g_alias, None, None
        g = _f32c(g)
        if is_deterministic():
            rev = ctx.rev
            if rev is not None and rev.shape[0] >= ctx.ns:
                base = g_alias if (g_alias is not None and g_alias.shape == (ctx.ns, ctx.c)) else None
                dx = torch.empty((ctx.ns, ctx.c), device=g.device, dtype=torch.float32)
                check(lib().mvk_max_pool_bwd_gather(_p(g), _p(arg), _p(inds), int(inds.dtype == torch.int64), inds.shape[1],
                                                    inds.shape[0], _p(rev), rev.shape[1], ctx.ns, ctx.c,
                                                    _p(_f32c(base)) if base is not None else None, _p(dx), _stream()))
                if g_alias is not None and base is None:
                    dx = dx + g_alias
                return dx, None, None
            _no_reverse_list("a max_pool matrix")
        if (g_alias is not None and g_alias.dtype == torch.float32 and g_alias.is_contiguous()
                and g_alias.shape == (ctx.ns, ctx.c) and _accumulate_in_place_ok(g_alias)):
            dx = g_alias            # this node's own grad input, not an operand of a pending product
        else:
            dx = _zeros((ctx.ns, ctx.c), g.device)
        check(lib().mvk_max_pool_bwd(_p(g), _p(arg), _p(inds), int(inds.dtype == torch.int64), inds.shape[0],
                                     inds.shape[1], ctx.ns, g.shape[1], _p(dx), _stream()))
        if g_alias is not None and dx is not g_alias:
            dx = dx + g_alias
        return dx, None, None


class _GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inds2d):
        _dev(x, inds2d)
        x = _f32c(x)
        ctx.rev = reverse_for(inds2d, first_column=True) if is_deterministic() else None
        inds2d, i64 = _idx(inds2d)
        Nq = inds2d.shape[0]
        stride = inds2d.shape[1] if inds2d.dim() == 2 else 1
        out = torch.empty((Nq, x.shape[1]), device=x.device, dtype=torch.float32)
        check(lib().mvk_gather_rows_fwd(_p(x), x.shape[0], x.shape[1], _p(inds2d), i64, Nq, stride, _p(out), _stream()))
        ctx.save_for_backward(inds2d)
        ctx.ns, ctx.stride = x.shape[0], stride
        return out

    @staticmethod
    def backward(ctx, g):
        (inds2d,) = ctx.saved_tensors
        # a column slice of a wider gradient (one half of the decoder's concatenation) is read in place
        if not (g.dtype == torch.float32 and g.dim() == 2 and g.stride(1) == 1 and g.stride(0) >= g.shape[1]):
            g = _f32c(g)
        if is_deterministic():
            rev = ctx.rev
            if rev is not None and rev.shape[0] >= ctx.ns:
                return gather_sum_rows(g, rev[:ctx.ns]), None
            _no_reverse_list("a closest_pool / upsampling matrix")
        dx = _zeros((ctx.ns, g.shape[1]), g.device)
        check(lib().mvk_gather_rows_bwd_ld(_p(g), g.stride(0), _p(inds2d), int(inds2d.dtype == torch.int64),
                                           inds2d.shape[0], ctx.stride, ctx.ns, g.shape[1], _p(dx), _stream()))
        return dx, None


class _UpsampleCatFn(torch.autograd.Function):
    """[closest_pool(x, inds) | skip] in one launch (mvk_gather_rows_cat_fwd); the backward reads the upsampled half of
    the gradient in place (as _GatherRowsFn does behind torch.cat) and hands the skip half on as a view."""

    @staticmethod
    def forward(ctx, x, inds2d, skip):
        _dev(x, inds2d, skip)
        x, skip = _f32c(x), _f32c(skip)
        ctx.rev = reverse_for(inds2d, first_column=True) if is_deterministic() else None
        inds2d, i64 = _idx(inds2d)
        Nq = inds2d.shape[0]
        if skip.shape[0] != Nq:
            raise RuntimeError("upsample_cat: the skip features and the upsampling indices differ in length")
        stride = inds2d.shape[1] if inds2d.dim() == 2 else 1
        C1, C2 = x.shape[1], skip.shape[1]
        out = torch.empty((Nq, C1 + C2), device=x.device, dtype=torch.float32)
        check(lib().mvk_gather_rows_cat_fwd(_p(x), x.shape[0], C1, _p(inds2d), i64, Nq, stride, _p(skip), C2, _p(out),
                                            _stream()))
        ctx.save_for_backward(inds2d)
        ctx.ns, ctx.stride, ctx.c1 = x.shape[0], stride, C1
        return out

    @staticmethod
    def backward(ctx, g):
        (inds2d,) = ctx.saved_tensors
        g = _f32c(g)
        c2 = g.shape[1] - ctx.c1
        if is_deterministic():
            rev = ctx.rev
            if rev is not None and rev.shape[0] >= ctx.ns:
                dx = gather_sum_rows(g[:, :ctx.c1], rev[:ctx.ns]) if ctx.needs_input_grad[0] else None
                return dx, None, (g[:, ctx.c1:].contiguous() if ctx.needs_input_grad[2] else None)
            _no_reverse_list("a closest_pool / upsampling matrix")
        dx = _zeros((ctx.ns, ctx.c1), g.device) if ctx.needs_input_grad[0] else None
        # the skip half leaves as a dense tensor of its own (same launch): the encoder block that produced the skip
        # features accumulates its own gradient onto it (max_pool / linear passthrough) instead of a separate add
        d_skip = torch.empty((g.shape[0], c2), device=g.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
        check(lib().mvk_gather_rows_cat_bwd(_p(g), _p(inds2d), int(inds2d.dtype == torch.int64), inds2d.shape[0],
                                            ctx.stride, ctx.ns, ctx.c1, c2, _p(dx), _p(d_skip), _stream()))
        return dx, None, d_skip


class _UpsampleCatLinearFn(torch.autograd.Function):
    """linear(cat([closest_pool(x, inds), skip], 1), W): the decoder's upsampling, concatenation and unary layer. Forward:
    the fused gather + concatenation, then the GEMM (statistics epilogue as in linear()). Backward: ONE product for both
    inputs -- g W is never stored, the GEMM's epilogue scatters its upsampled columns onto dx and writes the skip columns
    (mvk_gemm_f32_scatter_cat) -- and the weight gradient from the saved concatenation."""

    @staticmethod
    def forward(ctx, x, inds2d, skip, W, stats_n_valid):
        _dev(x, inds2d, skip, W)
        x, skip = _f32c(x), _f32c(skip)
        ctx.rev = reverse_for(inds2d, first_column=True) if is_deterministic() else None
        inds2d, i64 = _idx(inds2d)
        Nq = inds2d.shape[0]
        if skip.shape[0] != Nq:
            raise RuntimeError("upsample_cat_linear: the skip features and the upsampling indices differ in length")
        stride = inds2d.shape[1] if inds2d.dim() == 2 else 1
        C1, C2 = x.shape[1], skip.shape[1]
        cat = torch.empty((Nq, C1 + C2), device=x.device, dtype=torch.float32)
        check(lib().mvk_gather_rows_cat_fwd(_p(x), x.shape[0], C1, _p(inds2d), i64, Nq, stride, _p(skip), C2, _p(cat),
                                            _stream()))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(cat, W, inds2d)
        ctx.dims = (x.shape[0], C1, C2, stride, i64)
        if stats_n_valid is None:
            return gemm(cat, W, transB=True), None
        y, st = gemm(cat, W, transB=True, stats_n_valid=stats_n_valid)
        part = st[0] if st is not None else None
        if part is not None:
            ctx.mark_non_differentiable(part)
        return y, part

    @staticmethod
    def backward(ctx, g, g_part=None):
        cat, W, inds2d = ctx.saved_tensors
        ns, C1, C2, stride, i64 = ctx.dims
        g = _f32c(g)
        M, N, Kd = cat.shape[0], C1 + C2, W.shape[0]
        split_arena_prepare(g.device)
        if is_deterministic() and M > 0:
            rev = ctx.rev
            if rev is not None and rev.shape[0] >= ns:
                # (sum of the gradient rows of every coarse point's fine points) . W[:, :C1]: the scatter as a gather, then
                # the two halves of g W as two products -- fixed summation order throughout
                dx = gemm_ldb(gather_sum_rows(g, rev[:ns]), W, 0, C1, N) if ctx.needs_input_grad[0] else None
                d_skip = gemm_ldb(g, W, C1, C2, N) if ctx.needs_input_grad[2] else None
                dW = _dw_gemm(g, cat, target=W) if ctx.needs_input_grad[3] else None
                return dx, None, d_skip, dW, None
            _no_reverse_list("a closest_pool / upsampling matrix")
        dx = _zeros((ns, C1), g.device)
        split = gemm_plan(M, N, Kd, None, False)[0] if M > 0 else 1
        d_skip = _split_out((M, C2), g.device, split)
        if M > 0:
            check(lib().mvk_gemm_f32_scatter_cat(_p(g), _p(W), M, N, Kd, _p(inds2d), i64, stride, ns, C1, _p(dx),
                                                 _p(d_skip), _stream()))
        dW = _dw_gemm(g, cat, target=W) if ctx.needs_input_grad[3] else None
        return (dx if ctx.needs_input_grad[0] else None), None, (d_skip if ctx.needs_input_grad[2] else None), dW, None


def upsample_cat_linear(x, inds, skip, W, stats_n_valid=None, bn=None):
    """linear(upsample_cat(x, inds, skip), W, stats_n_valid=...) with a one-launch backward for x and skip."""
    _fin_request(bn if stats_n_valid is not None else None)
    try:
        y, part = _UpsampleCatLinearFn.apply(x, inds, skip, W, stats_n_valid)
    finally:
        fin = _fin_take()
    if part is not None:
        y._mvk_bn_stats = (part, gemm_plan(y.shape[0], y.shape[1], W.shape[1], None, True)[1], fin)
    return y


def upsample_cat(x, inds, skip):
    """torch.cat([closest_pool(x, inds), skip], dim=1) of the KPFCNN decoder (blocks.py:79-91, architectures.py:334)."""
    return _UpsampleCatFn.apply(x, inds, skip)


def max_pool(x, inds, passthrough=False):
    """blocks.py:94-110 (zero shadow row takes part in the max). passthrough=True returns (pooled, x') with x' an alias
    of x for another consumer of x, whose gradient the backward accumulates onto (see linear())."""
    out, alias = _MaxPoolFn.apply(x, inds, passthrough)
    return (out, alias) if passthrough else out


def closest_pool(x, inds):
    """blocks.py:79-91 (first column = closest neighbour because rows are sorted)."""
    return _GatherRowsFn.apply(x, inds)


# --------------------------------------------------------------------------------------------
# input pyramid: grid subsampling + radius neighbours (device resident)
# --------------------------------------------------------------------------------------------

import numpy as _np

_WS = {}


def _workspace(tag, nbytes, device):
    """Grow-only HBM scratch per (tag, device); owned by the Python host layer, handed to the C ABI."""
    key = (tag, device.index)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _lens_host(lens):
    if isinstance(lens, torch.Tensor):
        lens = lens.detach().cpu().numpy()
    return _np.ascontiguousarray(lens, dtype=_np.int32)


def grid_subsample_batch(points, lens, features=None, labels=None, dl=0.1, max_p=0, rotations=None):
    """Device version of cpp_subsampling.subsample_batch (wrapper.cpp:62-333).

    points [N,3] f32 (HBM), lens host int32 [B], optional features [N,fdim] f32 and labels [N,ldim]
    int32 (HBM). Returns (s_points [M,3] HBM, s_lens np.int32 [B][, s_features][, s_labels]).
    Bit-identical to the reference incl. output order. Synchronises once (the counts come back to the host).
    rotations (host float32 [B,3,3]): the random grid orientation of batch_grid_subsampling
    (datasets/common.py:89-134) applied inside the same call (rotate, subsample, rotate back)."""
    _dev(points, features, labels)
    points = _f32c(points)
    lens_h = _lens_host(lens)
    N, B = points.shape[0], int(lens_h.shape[0])
    fdim = ldim = 0
    if features is not None:
        features = _f32c(features)
        fdim = features.shape[1]
    if labels is not None:
        labels = labels.to(torch.int32).contiguous()
        if labels.dim() == 1:
            labels = labels.unsqueeze(1)
        ldim = labels.shape[1]
    ws = _workspace("sub", lib().mvk_grid_subsample_workspace(N, B, fdim, ldim), points.device)
    out_pts = torch.empty((max(N, 1), 3), device=points.device, dtype=torch.float32)
    out_f = torch.empty((max(N, 1), fdim), device=points.device, dtype=torch.float32) if fdim else None
    out_l = torch.empty((max(N, 1), ldim), device=points.device, dtype=torch.int32) if ldim else None
    out_lens = torch.empty((B,), device=points.device, dtype=torch.int32)
    out_lens_h = _np.empty((B,), _np.int32)
    tail = (_p(features), fdim, _p(labels), ldim, float(dl), int(max_p), _p(out_pts), _p(out_f), _p(out_l),
            _p(out_lens), out_lens_h.ctypes.data_as(C.c_void_p), _p(ws), ws.numel(), _stream())
    if rotations is None:
        check(lib().mvk_grid_subsample_batch(_p(points), N, lens_h.ctypes.data_as(C.c_void_p), B, *tail))
    else:
        rot = _np.ascontiguousarray(rotations, dtype=_np.float32)
        if rot.shape != (B, 3, 3):
            raise RuntimeError("grid_subsample_batch: rotations must be [B,3,3]")
        check(lib().mvk_grid_subsample_batch_oriented(_p(points), N, lens_h.ctypes.data_as(C.c_void_p), B,
                                                      rot.ctypes.data_as(C.c_void_p), *tail))
    M = int(out_lens_h.sum())
    res = [out_pts[:M], out_lens_h]
    if fdim:
        res.append(out_f[:M])
    if ldim:
        res.append(out_l[:M])
    return tuple(res)


_NB_GRID = {}      # device index -> workspace address that holds the last built grid


def radius_neighbors_batch(queries, supports, q_lens, s_lens, radius, limit=None, status=None, reuse_grid=False):
    """Device version of cpp_neighbors.batch_query (wrapper.cpp:58-238): int32 [Nq, W] in HBM.

    limit=None: W = the data-dependent max count (one extra counting pass + host sync, like the
    reference's two-pass fill); limit=k: keep the k nearest per row (= big_neighborhood_filter,
    datasets/common.py:411-421) without the counting pass, W = min(k, max count) like the reference.

    status (int32 [2] HBM tensor, needs limit): enqueue-only mode for a sync-free pyramid -- W = limit
    always, nothing is read back; status accumulates [max row count, overflow flag] for ONE later check
    (check_neighbor_status). reuse_grid=True (enqueue-only mode): the caller guarantees that the previous
    search on this device used the same supports, s_lens and radius, so its cell grid is reused."""
    _dev(queries, supports)
    q, s = _f32c(queries), _f32c(supports)
    ql, sl = _lens_host(q_lens), _lens_host(s_lens)
    if ql.shape[0] != sl.shape[0]:
        raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
    Nq, Ns, B = q.shape[0], s.shape[0], int(ql.shape[0])
    ws = _workspace("nb", lib().mvk_radius_neighbors_workspace(Nq, Ns, B), q.device)
    args = (_p(q), Nq, _p(s), Ns, ql.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), B, float(radius))
    if status is not None:
        if limit is None:
            raise RuntimeError("radius_neighbors_batch: the enqueue-only mode needs a column limit")
        _dev(status)
        width = int(limit)
        out = torch.empty((Nq, width), device=q.device, dtype=torch.int32)
        reuse = bool(reuse_grid) and _NB_GRID.get(q.device.index) == ws.data_ptr()   # a regrown workspace lost the grid
        if width > 0 and Nq > 0:
            check(lib().mvk_radius_neighbors_enqueue(*args, _p(out), width, _p(status), int(reuse), _p(ws),
                                                     ws.numel(), _stream()))
            _NB_GRID[q.device.index] = ws.data_ptr()
        return out
    _NB_GRID.pop(q.device.index, None)
    if limit is None:
        w = C.c_int(0)
        check(lib().mvk_radius_neighbors_batch(*args, None, 0, C.byref(w), _p(ws), ws.numel(), _stream()))
        width = int(w.value)
    else:
        width = int(limit)
    out = torch.empty((Nq, width), device=q.device, dtype=torch.int32)
    if width > 0 and Nq > 0:
        w = C.c_int(0)
        check(lib().mvk_radius_neighbors_batch(*args, _p(out), width, C.byref(w), _p(ws), ws.numel(), _stream()))
        if int(w.value) < width:            # reference shape: min(limit, data-dependent max count)
            out = out[:, :int(w.value)].contiguous()
    return out


def grid_subsample_dev(points, lens_dev, dl, out_points, out_lens_dev, status, rotations_dev=None,
                       total_out=None, pad_value=1e6):
    """Capturable subsampling (mvk_grid_subsample_batch_dev): points [cap_in,3] with DEVICE lens [B] ->
    out_points [out_cap,3] (rows past the total = pad_value), out_lens_dev [B], total_out [1]. No host sync."""
    _dev(points, lens_dev, out_points, out_lens_dev, status, rotations_dev, total_out)
    points = _f32c(points)
    cap_in, B = points.shape[0], int(lens_dev.shape[0])
    if not out_points.is_contiguous() or out_points.dtype != torch.float32:
        raise RuntimeError("grid_subsample_dev: out_points must be a contiguous float32 [cap,3] tensor")
    ws = _workspace("sub", lib().mvk_grid_subsample_workspace(cap_in, B, 0, 0), points.device)
    check(lib().mvk_grid_subsample_batch_dev(_p(points), cap_in, _p(lens_dev), B, _p(rotations_dev), float(dl),
                                             _p(out_points), out_points.shape[0], float(pad_value),
                                             _p(out_lens_dev), _p(total_out), _p(status), _p(ws), ws.numel(), _stream()))


def radius_neighbors_dev(queries, supports, q_lens_dev, s_lens_dev, radius, out, shadow, status, reuse_grid=False,
                         rev=None, rev_counts=None, rev_status=None):
    """Capturable neighbour search (mvk_radius_neighbors_dev) into the fixed matrix out [Nq_cap, width] int32.
    rev [Ns_cap, Hr] int32 + rev_counts [Ns_cap] int32 (zero) + rev_status [2]: the search also fills the transposed
    relation while it writes the rows (mvk_radius_neighbors_dev_rev, width <= 64; rows in order of arrival); finish the
    lists of the whole pyramid with reverse_finish_many()."""
    _dev(queries, supports, q_lens_dev, s_lens_dev, out, status, rev, rev_counts, rev_status)
    q, s = _f32c(queries), _f32c(supports)
    if out.dtype != torch.int32 or not out.is_contiguous() or out.shape[0] != q.shape[0]:
        raise RuntimeError("radius_neighbors_dev: out must be a contiguous int32 [Nq_cap, width] tensor")
    B = int(q_lens_dev.shape[0])
    ws = _workspace("nb", lib().mvk_radius_neighbors_workspace(q.shape[0], s.shape[0], B), q.device)
    if rev is not None:
        if (rev.dtype != torch.int32 or not rev.is_contiguous() or rev.dim() != 2 or rev.shape[0] < s.shape[0]
                or rev_counts is None or rev_counts.dtype != torch.int32 or rev_counts.numel() < s.shape[0] or rev_status is None):
            raise RuntimeError("radius_neighbors_dev: rev must be a contiguous int32 [>= Ns_cap, Hr] matrix with int32 "
                               "counters [>= Ns_cap] and a status word")
        check(lib().mvk_radius_neighbors_dev_rev(_p(q), q.shape[0], _p(s), s.shape[0], _p(q_lens_dev), _p(s_lens_dev), B,
                                                 float(radius), _p(out), out.shape[1], int(shadow), _p(status),
                                                 int(bool(reuse_grid)), _p(ws), ws.numel(), _p(rev), rev.shape[1],
                                                 _p(rev_counts), _p(rev_status), _stream()))
    else:
        check(lib().mvk_radius_neighbors_dev(_p(q), q.shape[0], _p(s), s.shape[0], _p(q_lens_dev), _p(s_lens_dev), B,
                                             float(radius), _p(out), out.shape[1], int(shadow), _p(status),
                                             int(bool(reuse_grid)), _p(ws), ws.numel(), _stream()))
    _NB_GRID.pop(q.device.index, None)


REG_MANY = 16           # mvkpconv.h MVK_REG_MANY
REV_MANY = 12      # include/mvkpconv.h: MVK_REV_MANY


def reverse_finish_many(lists, status=None):
    """lists: [(rev [rows, Hr] int32, counts [>= rows] int32, rows, shadow)] filled by radius_neighbors_dev(rev=...): pads the
    tails, returns the counters to zero and accumulates the longest row into status[0] -- one launch for up to 12 lists
    (mvk_reverse_finish_many)."""
    from ._lib import RevList
    for i0 in range(0, len(lists), REV_MANY):
        chunk = lists[i0:i0 + REV_MANY]
        arr = (RevList * len(chunk))()
        for k, (rev, counts, rows, shadow) in enumerate(chunk):
            _dev(rev, counts, status)
            arr[k] = RevList(_p(rev), _p(counts), _p(status), int(rows), int(rev.shape[1]), int(shadow))
        check(lib().mvk_reverse_finish_many(arr, len(chunk), _stream()))


def neighbors_cell_order(Nq, Ns, B, out=None, s_lens_dev=None, device=None):
    """Work list for kpconv(order=...) out of the cell grid the LAST neighbour search on this device built
    (mvk_neighbors_cell_order; the search must have had these Nq / Ns / B -- the capacities for radius_neighbors_dev,
    whose device lengths go in s_lens_dev): the support rows sorted by cloud, grid cell and row, int32 [Ns] (or written
    into `out` [cap >= Ns], identity beyond the clouds' rows). One launch, no synchronisation."""
    dev = out.device if out is not None else (s_lens_dev.device if s_lens_dev is not None else device)
    if out is None:
        out = torch.empty((Ns,), device=dev, dtype=torch.int32)
    _dev(out, s_lens_dev)
    if out.dtype != torch.int32 or not out.is_contiguous() or out.dim() != 1 or out.shape[0] < Ns:
        raise RuntimeError("neighbors_cell_order: out must be a contiguous int32 vector of at least Ns entries")
    ws = _workspace("nb", lib().mvk_radius_neighbors_workspace(Nq, Ns, B), dev)
    check(lib().mvk_neighbors_cell_order(Ns, B, _p(s_lens_dev), _p(out), out.shape[0], _p(ws), ws.numel(), _stream()))
    return out


_REV_COUNTS = {}      # device index -> persistent zero int32 buffer of mvk_reverse_neighbors (self-cleaning)
_REV_COUNTS_RETIRED = []      # outgrown buffers: a graph captured earlier still adds into and zeroes its old address
REVERSE_DX = os.environ.get("MVK_REVERSE_DX", "1") == "1"      # development switch: 0 = the scatter backward everywhere
REVERSE_DX_DEFORM = os.environ.get("MVK_REVERSE_DX_DEFORM", "1") == "1"     # 0: deformable layers keep the atomic scatter
REVERSE_DX_ANY_COUT = os.environ.get("MVK_REVERSE_DX_ANY_COUT", "1") == "1"  # 0: gather form only for Cout a power of two >= 32


def _rev_counts(n, device):
    # one buffer per device: reverse lists are built by ONE stream at a time (the input side of a step); two builds
    # running concurrently on different streams would share the counters
    key = device.index
    buf = _REV_COUNTS.get(key)
    if buf is None or buf.numel() < n:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ops.reverse_neighbors: the counter buffer of this device does not exist yet (or is too "
                               "small) and cannot be created inside a graph capture; build one batch eagerly first")
        if buf is not None:
            # never handed back to the allocator: replays of earlier captures keep using it
            _REV_COUNTS_RETIRED.append(buf)
        buf = torch.zeros(max(int(n), 1 << 16), dtype=torch.int32, device=device)
        torch.cuda.current_stream(device).synchronize()      # zero before any other stream's build uses it
        _REV_COUNTS[key] = buf
    return buf


REV_MAX_WIDTH = 8192      # include/mvkpconv.h: MVK_REV_MAX_WIDTH


def reverse_width_cap(sort=None):
    """Longest row a reverse list may have (8192, sorted or not): the wide relations of the deformable layers -- a support
    near the middle of a coarse level is a neighbour of almost every query at the deform radius."""
    return REV_MAX_WIDTH


def reverse_neighbors(idx, Ns, width=None, out=None, status=None, shadow=None, sort=None, first_column=False):
    """The transposed neighbourhood relation of idx [Nq, H] (int32 / int64, entries outside [0, Ns) = shadow): rev
    [Ns, width] int32, row j = the rows n of idx that contain j, padded with `shadow` (default Nq)
    (mvk_reverse_neighbors). It turns the feature gradient of a rigid KPConv into a gather (kpconv(..., rev=...)).
    sort: rows ascending (what run-to-run identical sums need; default: in deterministic mode) or in order of arrival
    (saves the ranking pass). first_column: the relation of idx[:, 0] alone (nearest upsampling).
    width None: the exact longest row (one read-back: eager use); with a width, `status` (int32 [2] HBM) accumulates
    [longest row, overflow] without any synchronisation -- check it with check_reverse_status. out: write into this
    int32 [>= Ns, width] matrix instead of allocating (capacity-padded batches)."""
    _dev(idx, out, status)
    idx, i64 = _idx(idx)
    Nq, H = idx.shape[0], idx.shape[1] if idx.dim() == 2 else 0
    stride = H
    if first_column:
        H = min(H, 1)
    Ns = int(Ns)
    shadow = Nq if shadow is None else int(shadow)
    dev = idx.device
    counts = _rev_counts(Ns, dev)
    do_sort = int(is_deterministic() if sort is None else bool(sort))

    def run(dst, w, st):
        check(lib().mvk_reverse_neighbors(_p(idx), i64, Nq, H, max(stride, H), Ns, _p(dst), int(w), shadow, do_sort,
                                          _p(counts), _p(st), _stream()))

    if out is not None:
        if out.dtype != torch.int32 or not out.is_contiguous() or out.dim() != 2 or out.shape[0] < Ns:
            raise RuntimeError("reverse_neighbors: out must be a contiguous int32 [>= Ns, width] matrix")
        run(out, out.shape[1], status)
        return out
    if width is not None:
        rev = torch.empty((Ns, max(int(width), 1)), device=dev, dtype=torch.int32)
        run(rev, rev.shape[1], status)
        return rev
    st = torch.zeros(2, dtype=torch.int32, device=dev)
    cap = reverse_width_cap(do_sort)
    w = max(8, min(cap, 2 * H + 8))
    rev = torch.empty((Ns, w), device=dev, dtype=torch.int32)
    run(rev, w, st)
    longest, ovf = (int(v) for v in st.cpu())
    if ovf:
        if longest > cap:
            raise RuntimeError("reverse_neighbors: a support has %d reverse neighbours (> %d)" % (longest, cap))
        rev = torch.empty((Ns, longest), device=dev, dtype=torch.int32)
        run(rev, longest, None)
        return rev
    return rev[:, :max(longest, 1)].contiguous()


# reverse lists of pooling / upsampling matrices, found again by the matrix itself (the blocks hand max_pool and
# closest_pool nothing but `batch.pools[l]` / `batch.upsamples[l]`): deterministic mode's gather-form backwards
_REVERSES = collections.OrderedDict()
_REVERSES_KEPT = 256


def _rev_key(inds, first):
    return (inds.device.index, inds.data_ptr(), tuple(inds.shape), inds.dtype, bool(first))


def remember_reverse(inds, rev, first_column=False):
    """Registers rev = reverse_neighbors(inds, ...) (first_column: of inds[:, 0]) under the index matrix itself. The
    entry holds a weak reference to THAT tensor object: another matrix of the same shape at a recycled address never
    matches (a wrong reverse list would be a silently wrong gradient, unlike a stale work list)."""
    key = _rev_key(inds, first_column)
    if len(_REVERSES) >= 16:
        # entries whose index matrix is gone hold tens of MB each at level 0: drop them now, not 256 batches later
        for k in [k for k, e in _REVERSES.items() if e[0]() is None]:
            del _REVERSES[k]
    _REVERSES[key] = (weakref.ref(inds), rev)
    _REVERSES.move_to_end(key)
    while len(_REVERSES) > _REVERSES_KEPT:
        _REVERSES.popitem(last=False)


def reverse_for(inds, first_column=False):
    """The reverse list registered for this very tensor object (call it where the matrix is still the object the batch
    holds: in the forward), or None."""
    if not (torch.is_tensor(inds) and inds.is_cuda):
        return None
    e = _REVERSES.get(_rev_key(inds, first_column))
    return e[1] if (e is not None and e[0]() is inds) else None


_DET_WARNED = set()


def _no_reverse_list(what):
    """Deterministic mode met a scatter without a reverse list (a deformable block's pooling, a batch that was moved
    or rebuilt outside datasets.common): the atomic path runs, the result is correct but not bit-reproducible."""
    if what not in _DET_WARNED:
        _DET_WARNED.add(what)
        import warnings
        warnings.warn("deterministic mode: no reverse list for %s; its backward uses float atomics (not bit-reproducible)" % what)


def gather_sum_rows(g, rev, base=None):
    """out[j] = (base[j] or 0) + sum of g[n] over the entries n of rev[j] (mvk_gather_sum_rows); g [Nq, C] may be a column
    block of a wider tensor (row stride >= C)."""
    _dev(g, rev, base)
    if not (g.dtype == torch.float32 and g.dim() == 2 and g.stride(1) == 1 and g.stride(0) >= g.shape[1]):
        g = _f32c(g)
    Ns, Hr = rev.shape
    out = torch.empty((Ns, g.shape[1]), device=g.device, dtype=torch.float32)
    if base is not None:
        base = _f32c(base)
    check(lib().mvk_gather_sum_rows(_p(g), g.stride(0), g.shape[0], _p(rev), Hr, Ns, g.shape[1], _p(base), _p(out), _stream()))
    return out


def gemm_ldb(A, B_base, col0, N, ldb):
    """A [M,Kd] @ B[:, col0:col0+N] for a row-major B [Kd, ldb] read in place (mvk_gemm_f32_ldb)."""
    _dev(A, B_base)
    A, B_base = _f32c(A), _f32c(B_base)
    M, Kd = A.shape
    split_arena_prepare(A.device)
    out = _split_out((M, N), A.device, gemm_plan(M, N, Kd)[0] if M > 0 and N > 0 else 1)
    if M > 0 and N > 0:
        check(lib().mvk_gemm_f32_ldb(_p(A), C.c_void_p(B_base.data_ptr() + 4 * int(col0)), _p(out), M, int(N), Kd, int(ldb),
                                     _stream()))
    return out


def check_reverse_status(status):
    """Reads a reverse-list status word back (synchronises): raises when a row did not fit its width."""
    longest, ovf = (int(v) for v in status.cpu())
    if ovf:
        raise RuntimeError("reverse neighbours: a support row has %d entries, more than the width it was built with"
                           % longest)
    return longest


def _neg_kernel_points(kp):
    """-kp, cached on the tensor (kernel points are frozen parameters; the cache follows in-place updates)."""
    c = getattr(kp, "_mvk_neg", None)
    if c is None or c[0] != kp._version:
        c = (kp._version, (-kp.detach()).contiguous())
        kp._mvk_neg = c
    return c[1]


def kp_transposed_contraction(A2, W):
    """dx [M, Cin] = sum_k A2[:, k, :] . W[k]^T for A2 [M, K, Cout], W [K, Cin, Cout] (mvk_gemm_f32_kp_transposed; a Cout
    that is not a power of two >= 32 -- the 45 / 60 offset channels of a deformable layer's inner convolution -- takes a
    transposed copy of the weights and the plain product)."""
    _dev(A2, W)
    A2, W = _f32c(A2), _f32c(W)
    M, K, Cout = A2.shape
    Cin = W.shape[1]
    if W.shape[0] != K or W.shape[2] != Cout:
        raise RuntimeError("kp_transposed_contraction: A2 [M,K,Cout] and W [K,Cin,Cout] expected")
    if Cout < 32 or (Cout & (Cout - 1)) != 0:
        return gemm(A2.view(M, K * Cout), W.permute(0, 2, 1).reshape(K * Cout, Cin).contiguous())
    split_arena_prepare(A2.device)
    split = gemm_plan(M, Cin, K * Cout)[0] if M > 0 else 1
    dx = _split_out((M, Cin), A2.device, split)
    if M > 0:
        check(lib().mvk_gemm_f32_kp_transposed(_p(A2), _p(W), _p(dx), M, K, Cin, Cout, _stream()))
    return dx


_WORK_ORDERS = collections.OrderedDict()      # (device index, address, rows) of a points tensor -> its work list
_WORK_ORDERS_KEPT = 64


def remember_work_order(points, order):
    """Side channel for batch containers that cannot carry the lists (the reference's flat input_list): the work list
    of a level, found again by the address and row count of its points tensor. A stale hit (another cloud of the same
    size at a recycled address) is still a permutation of its rows, which is all the gather needs."""
    key = (points.device.index, points.data_ptr(), points.shape[0])
    _WORK_ORDERS[key] = order
    _WORK_ORDERS.move_to_end(key)
    while len(_WORK_ORDERS) > _WORK_ORDERS_KEPT:
        _WORK_ORDERS.popitem(last=False)


def work_order_for(points):
    return _WORK_ORDERS.get((points.device.index, points.data_ptr(), points.shape[0])) if points.is_cuda else None


def check_neighbor_status(status):
    """Reads an enqueue-only status word back (synchronises with the work that produced it)."""
    maxc, ovf = (int(v) for v in status.cpu())
    if ovf:
        raise RuntimeError("neighbors: a query has more in-range supports than the in-kernel list holds")
    return maxc


def pad_points(src, dst, fill, count_out=None):
    """dst [cap,w] <- src [n,w] padded with `fill`; count_out (int32 [1] HBM) <- n. One launch."""
    _dev(src, dst, count_out)
    src = _f32c(src)
    if not dst.is_contiguous() or dst.dtype != torch.float32 or dst.shape[1] != src.shape[1]:
        raise RuntimeError("pad_points: destination must be a contiguous float32 [cap, %d] tensor" % src.shape[1])
    check(lib().mvk_pad_points(_p(src), src.shape[0], _p(dst), dst.shape[0], src.shape[1], float(fill),
                               _p(count_out), _stream()))


def pad_index_rows(src, shadow_src, dst, shadow_dst):
    """dst [cap,w_dst] <- src [n,w_src] with the shadow index rewritten, padded with shadow_dst. One launch."""
    _dev(src, dst)
    src, i64 = _idx(src)
    if dst.dtype != src.dtype or not dst.is_contiguous():
        raise RuntimeError("pad_index_rows: destination must be contiguous and of the source's index type")
    w_src = src.shape[1] if src.dim() == 2 else 0
    check(lib().mvk_pad_index_rows(_p(src), i64, src.shape[0], w_src, int(shadow_src), _p(dst), dst.shape[0],
                                   dst.shape[1], int(shadow_dst), _stream()))


# --------------------------------------------------------------------------------------------
# 2D -> 3D fusion
# --------------------------------------------------------------------------------------------

def unproject_depth(depth_mm, cam_matrix, poses):
    """depth (nv,h,w) uint16/int16-compatible HBM tensor [mm], cam_matrix (>=3x3) float32 host array
    (already rescaled to the image size), poses (nv,4,4) f32 HBM -> (xyz (nv,h,w,3) f64, valid (nv,h,w) bool).
    ScanNet_sphere_color.py:66-72,409-417 (float64 because the pixel grid is int64 there)."""
    _dev(depth_mm, poses)
    nv, h, w = depth_mm.shape
    d16 = depth_mm.to(torch.int16).contiguous() if depth_mm.dtype != torch.int16 else depth_mm.contiguous()
    kinv = _np.linalg.inv(_np.asarray(cam_matrix, dtype=_np.float32)[:3, :3]).astype(_np.float64)  # :71 (f32 inverse)
    kinv = _np.ascontiguousarray(kinv)
    poses = _f32c(poses)
    xyz = torch.empty((nv, h, w, 3), device=depth_mm.device, dtype=torch.float64)
    valid = torch.empty((nv, h, w), device=depth_mm.device, dtype=torch.uint8)
    check(lib().mvk_unproject_depth(_p(d16), nv, h, w, kinv.ctypes.data_as(C.c_void_p), _p(poses), _p(xyz),
                                    _p(valid), _stream()))
    return xyz, valid.bool()


def knn_pixels(sphere_points, image_xyz, image_mask, k=3):
    """Exact k-NN (float64) of sphere points among valid unprojected pixels -> flat pixel indices
    view*h*w + row*w + col, int64 [ns,k] (ScanNet_sphere_color.py:436-451)."""
    _dev(sphere_points, image_xyz, image_mask)
    q = _f32c(sphere_points)
    keys = image_xyz.reshape(-1, 3).to(torch.float64).contiguous()
    valid = image_mask.reshape(-1).to(torch.uint8).contiguous()
    nq, nk = q.shape[0], keys.shape[0]
    ws = _workspace("knn", lib().mvk_knn_workspace(nq, nk, k), q.device)
    out = torch.empty((nq, k), device=q.device, dtype=torch.int64)
    check(lib().mvk_knn_f64(_p(q), nq, _p(keys), _p(valid), nk, int(k), _p(out), _p(ws), ws.numel(), _stream()))
    return out


class _GroupPointsFn(torch.autograd.Function):
    """mvpnet/ops/group_points.py:5-17; float32 and float64 like the reference's extension."""

    @staticmethod
    def forward(ctx, points, index):
        _dev(points, index)
        if index.dtype != torch.int64:
            raise RuntimeError("group_points: index must be int64")
        if points.dim() != 3 or index.dim() != 3 or points.shape[0] != index.shape[0]:
            raise RuntimeError("group_points: expected points (B,C,N1) and index (B,N2,K)")
        f64 = points.dtype == torch.float64
        points, index = (points.contiguous() if f64 else _f32c(points)), index.contiguous()
        B, Cc, N1 = points.shape
        _, N2, K = index.shape
        out = torch.empty((B, Cc, N2, K), device=points.device, dtype=points.dtype)
        fn = lib().mvk_group_points_fwd_f64 if f64 else lib().mvk_group_points_fwd
        check(fn(_p(points), _p(index), B, Cc, N1, N2, K, _p(out), _stream()))
        ctx.save_for_backward(index)
        ctx.n1, ctx.f64 = N1, f64
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (index,) = ctx.saved_tensors
        g = grad_out.double().contiguous() if ctx.f64 else _f32c(grad_out)
        B, Cc, N2, K = g.shape
        gi = torch.zeros((B, Cc, ctx.n1), device=g.device, dtype=g.dtype)
        fn = lib().mvk_group_points_bwd_f64 if ctx.f64 else lib().mvk_group_points_bwd
        check(fn(_p(g), _p(index), B, Cc, ctx.n1, N2, K, _p(gi), _stream()))
        return gi, None


def group_points(points, index):
    """points (B,C,N1), index (B,N2,K) int64 -> (B,C,N2,K) (mvpnet/ops/group_points.py:20-31)."""
    return _GroupPointsFn.apply(points, index)


# --------------------------------------------------------------------------------------------
# fused FeatureAggregation path: gather kernel + MFMA linear layers
# --------------------------------------------------------------------------------------------

def fa_gather(feature_2d, image_xyz, knn, points):
    """X [C+4, np*k] channel-major for ONE sphere: feature_2d (nv,C,h,w) f32, image_xyz (nv,h,w,3) f32,
    knn (np,k) int64 flat pixel indices, points (np,3) f32. Treated as a constant by autograd (the 2D
    encoder is frozen in every MV-KPConv variant, architectures_sphere.py:234-237)."""
    _dev(feature_2d, image_xyz, knn, points)
    f = feature_2d.detach()
    # the frozen encoder's convolutions deliver channels-last maps: read in place (no NCHW copy of the map)
    cl = (f.dim() == 4 and f.dtype == torch.float32 and not f.is_contiguous()
          and f.is_contiguous(memory_format=torch.channels_last))
    if not cl:
        f = _f32c(f)
    xyz, pts = _f32c(image_xyz), _f32c(points)
    knn = knn.contiguous()
    if knn.dtype != torch.int64:
        raise RuntimeError("fa_gather: knn must be int64")
    nv, Cc, h, w = f.shape
    np_, k = knn.shape
    X = torch.empty((Cc + 4, np_ * k), device=f.device, dtype=torch.float32)
    check(lib().mvk_fa_gather_fwd_ex(_p(f), int(cl), _p(xyz), _p(knn), _p(pts), Cc, nv, h * w, np_, k, _p(X), _stream()))
    return X


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T on the f32 MFMA GEMM. x is [M,Kd] row-major, or [Kd,M] when x_is_transposed.
    passthrough: also returns x itself as a second (aliasing) output for a parallel consumer of x -- the shortcut of a
    residual block; the backward then ACCUMULATES g @ W into that consumer's gradient inside the GEMM (its epilogue adds
    onto the buffer) instead of leaving autograd to sum two tensors with one more launch per block."""

    @staticmethod
    def forward(ctx, x, W, x_is_transposed, stats_n_valid=None, passthrough=False):
        ctx.save_for_backward(x, W)
        ctx.xt = bool(x_is_transposed)
        ctx.set_materialize_grads(False)    # no zero tensors for unused gradients (statistics, an unused passthrough)
        alias = x.view_as(x) if passthrough else None
        if stats_n_valid is None:
            return gemm(x, W, transA=ctx.xt, transB=True), None, alias
        y, st = gemm(x, W, transA=ctx.xt, transB=True, stats_n_valid=stats_n_valid)
        part = st[0] if st is not None else None
        if part is not None:
            ctx.mark_non_differentiable(part)
        return y, part, alias

    @staticmethod
    def backward(ctx, g, g_part=None, g_alias=None):
        x, W = ctx.saved_tensors
        if g is None:
            return g_alias, None, None, None, None
        g = _f32c(g)
        dx = dW = None
        if ctx.needs_input_grad[0]:
            if ctx.xt:
                dx = gemm(W, g, transA=True, transB=True)                    # [Kd,M]
                if g_alias is not None:
                    dx = dx + g_alias
            elif (g_alias is not None and g_alias.dtype == torch.float32 and g_alias.is_contiguous()
                  and g_alias.shape == (g.shape[0], W.shape[1]) and _accumulate_in_place_ok(g_alias)):
                # the parallel consumer's gradient is this node's own grad input and no pending product reads it: add onto it
                dx = gemm(g, W, out=g_alias, accumulate=True)
            else:
                dx = gemm(g, W)                                              # [M,Kd]
                if g_alias is not None:
                    dx = dx + g_alias
        if ctx.needs_input_grad[1]:
            # dW [N,Kd] = g^T [N,M] @ x [M,Kd]
            dW = _dw_gemm(g, x, transB=ctx.xt, target=W)
        return dx, dW, None, None, None


_GEMM_DUAL = os.environ.get("MVK_GEMM_DUAL", "1") == "1"


def gemm_dual(A, B, A2, B2):
    """A @ B + A2 @ B2 (all row-major, same output shape) in one launch with the two reductions laid end to end
    (mvk_gemm_f32_dual), or None when the shape is not supported (the caller runs two products)."""
    _dev(A, B, A2, B2)
    A, B, A2, B2 = _f32c(A), _f32c(B), _f32c(A2), _f32c(B2)
    M, Kd = A.shape
    N, Kd2 = B.shape[1], A2.shape[1]
    if A2.shape[0] != M or B.shape[0] != Kd or B2.shape != (Kd2, N):
        raise RuntimeError("gemm_dual: shapes do not match")
    split_arena_prepare(A.device)
    split = C.c_int(0)
    check(lib().mvk_gemm_f32_dual_plan(M, N, Kd, Kd2, C.byref(split)))
    if split.value == 0:
        return None
    out = _split_out((M, N), A.device, split.value)
    check(lib().mvk_gemm_f32_dual(_p(A), _p(B), _p(A2), _p(B2), _p(out), M, N, Kd, Kd2, _stream()))
    return out


class _LinearPairFn(torch.autograd.Function):
    """(x W0^T, x W1^T) in one launch (mvk_gemm_f32_pair): unary1 and the shortcut layer of a bottleneck block read the
    same input. The backward is the two layers' own: dx = g0 W0 + g1 W1 (the second product accumulates onto the first),
    the weight gradients go through _dw_gemm like every other layer's."""

    @staticmethod
    def forward(ctx, x, W0, W1, stats_n_valid, plan):
        M, Kd = x.shape
        N0, N1 = W0.shape[0], W1.shape[0]
        _, s0, s1, r0, r1 = plan
        want = r0 > 0 or r1 > 0
        outs, parts = [], []
        for N, sp, rows in ((N0, s0, r0), (N1, s1, r1)):
            outs.append(_split_out((M, N), x.device, sp))
            parts.append(torch.empty(((M + rows - 1) // rows, 2, N), device=x.device, dtype=torch.float32) if rows > 0 else None)
        req = _FIN.get("req_pair")
        fins = [None, None]
        if req is not None and want:
            done = [None, None]
            for i, (bn, part, N) in enumerate(((req[0], parts[0], N0), (req[1], parts[1], N1))):
                if bn is not None and part is not None:
                    fins[i], done[i] = _fin_struct(bn, N, x.device)
            _FIN["done_pair"] = tuple(done)
        if fins[0] is not None or fins[1] is not None:
            check(lib().mvk_gemm_f32_pair_bn(_p(x), _p(W0), _p(W1), _p(outs[0]), _p(outs[1]), M, N0, N1, Kd, 1,
                                             _p(parts[0]), _p(parts[1]), _p(stats_n_valid),
                                             C.byref(fins[0]) if fins[0] is not None else None,
                                             C.byref(fins[1]) if fins[1] is not None else None, _stream()))
        else:
            check(lib().mvk_gemm_f32_pair(_p(x), _p(W0), _p(W1), _p(outs[0]), _p(outs[1]), M, N0, N1, Kd, 1, int(want),
                                          _p(parts[0]), _p(parts[1]), _p(stats_n_valid) if want else None, _stream()))
        ctx.save_for_backward(x, W0, W1)
        ctx.set_materialize_grads(False)
        for part in parts:
            if part is not None:
                ctx.mark_non_differentiable(part)
        return outs[0], parts[0], outs[1], parts[1]

    @staticmethod
    def backward(ctx, g0, gp0, g1, gp1):
        x, W0, W1 = ctx.saved_tensors
        dx = dW0 = dW1 = None
        g0 = _f32c(g0) if g0 is not None else None
        g1 = _f32c(g1) if g1 is not None else None
        if ctx.needs_input_grad[0]:
            if g0 is not None and g1 is not None and _GEMM_DUAL:
                dx = gemm_dual(g0, W0, g1, W1)          # g0 W0 + g1 W1 as one launch, or None
            if dx is None:
                if g0 is not None:
                    dx = gemm(g0, W0)
                if g1 is not None:
                    dx = gemm(g1, W1, out=dx, accumulate=True) if dx is not None else gemm(g1, W1)
        if ctx.needs_input_grad[1] and g0 is not None:
            dW0 = _dw_gemm(g0, x, target=W0)
        if ctx.needs_input_grad[2] and g1 is not None:
            dW1 = _dw_gemm(g1, x, target=W1)
        return dx, dW0, dW1, None, None


def linear_pair(x, W0, W1, stats_n_valid=None, bn0=None, bn1=None):
    """(linear(x, W0), linear(x, W1)) in one launch, or None when the two products cannot share one (different tile
    classes: the caller then runs them one after the other). Outputs carry their BatchNorm statistics like linear()'s."""
    _dev(x, W0, W1, stats_n_valid)
    if x.dtype != torch.float32 or not x.is_contiguous() or W0.dtype != torch.float32 or W1.dtype != torch.float32 \
            or not W0.is_contiguous() or not W1.is_contiguous() or x.dim() != 2 or x.shape[0] == 0:
        return None
    M, Kd = x.shape
    split_arena_prepare(x.device)
    fold = BN_FOLD and (bn0 is not None or bn1 is not None)       # finished statistics: the epilogue whatever the row count
    want = stats_n_valid is not None and M <= _STATS_EPILOGUE_ROWS and \
        (fold or (M > bn_single_launch_rows(W0.shape[0]) and M > bn_single_launch_rows(W1.shape[0])))
    plan = (C.c_int * 5)()
    check(lib().mvk_gemm_f32_pair_plan(M, W0.shape[0], W1.shape[0], Kd, int(want), plan))
    if not plan[0]:
        return None
    plan = tuple(int(v) for v in plan)

    def ok(bn):
        return bn if (BN_FOLD and bn is not None and bn.training and _SYNC_BN["group"] is None
                      and bn.num_features % 4 == 0) else None
    _FIN["req_pair"] = (ok(bn0), ok(bn1)) if want else None
    _FIN["done_pair"] = None
    try:
        y0, p0, y1, p1 = _LinearPairFn.apply(x, W0, W1, stats_n_valid if want else None, plan)
    finally:
        done = _FIN.get("done_pair") or (None, None)
        _FIN["req_pair"] = None
        _FIN["done_pair"] = None
    if p0 is not None:
        y0._mvk_bn_stats = (p0, plan[3], done[0])
    if p1 is not None:
        y1._mvk_bn_stats = (p1, plan[4], done[1])
    return y0, y1


def linear(x, W, x_is_transposed=False, stats_n_valid=None, passthrough=False, bn=None):
    """nn.Linear without bias / a 1x1 convolution over rows, on gemm_f32_mfma. stats_n_valid: see kpconv().
    passthrough=True returns (y, x') with x' an alias of x whose gradient is summed into x's inside the backward GEMM
    (use x' for the other consumer of x). bn: the nn.BatchNorm1d that follows (statistics finished by the product)."""
    _fin_request(bn if stats_n_valid is not None else None)
    try:
        y, part, alias = _LinearFn.apply(x, W, x_is_transposed, stats_n_valid, passthrough)
    finally:
        fin = _fin_take()
    if part is not None:        # the plan is a pure function of the shape: the same rows the product just used
        y._mvk_bn_stats = (part, gemm_plan(y.shape[0], y.shape[1], W.shape[1], None, True)[1], fin)
    return (y, alias) if passthrough else y


class _XentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, lut, weight):
        _dev(logits, labels, lut, weight)
        logits = _f32c(logits)
        if labels.dtype not in (torch.int32, torch.int64):
            labels = labels.long()
        labels = labels.contiguous()
        N, Cc = logits.shape
        part = torch.empty((lib().mvk_xent_workspace_floats(N),), device=logits.device, dtype=torch.float32)
        out2 = torch.empty((2,), device=logits.device, dtype=torch.float32)
        check(lib().mvk_xent_fwd(_p(logits), N, Cc, _p(labels), int(labels.dtype == torch.int64), _p(lut), lut.numel(),
                                 _p(weight), _p(part), _p(out2), _stream()))
        ctx.save_for_backward(logits, labels, lut, weight, out2)
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        logits, labels, lut, weight, out2 = ctx.saved_tensors
        g = _f32c(g)
        d = torch.empty_like(logits)
        check(lib().mvk_xent_bwd(_p(logits), logits.shape[0], logits.shape[1], _p(labels),
                                 int(labels.dtype == torch.int64), _p(lut), lut.numel(), _p(weight), _p(out2), _p(g), _p(d),
                                 _stream()))
        return d, None, None, None


def cross_entropy_lut(logits, labels, lut, class_weight=None):
    """KPFCNN's segmentation loss (architectures.py:345-372) as one autograd node, three launches forward + backward:
    labels [N] (int32 / int64, raw dataset values) are renumbered through `lut` (int32: lut[l + 1] = class of label
    l or -1 = ignored; lut[0] for negative labels, lut[-1] for labels above the table), then the class-weighted
    cross entropy of logits [N, C], mean over the kept points (torch.nn.CrossEntropyLoss(weight, ignore_index=-1))."""
    if lut.dtype != torch.int32 or lut.dim() != 1 or lut.numel() < 3:
        raise RuntimeError("cross_entropy_lut: lut must be a 1-D int32 tensor of at least 3 entries")
    if logits.dim() != 2 or labels.dim() != 1 or labels.shape[0] != logits.shape[0]:
        raise RuntimeError("cross_entropy_lut: logits [N, C] and labels [N] expected")
    if class_weight is not None and (class_weight.dtype != torch.float32 or class_weight.numel() != logits.shape[1]):
        raise RuntimeError("cross_entropy_lut: class_weight must be float32 [C]")
    return _XentFn.apply(logits, labels, lut, None if class_weight is None else class_weight.contiguous())


def bias_act_nhwc(x, bias, res=None, bias2=None, relu=True, out=None):
    """act(x + bias[c] (+ res (+ bias2[c]))) on a channels-last 4-D tensor (in place unless `out`): the pointwise
    tail of a convolution of the frozen 2D encoder after its BatchNorm was folded into the weights."""
    _dev(x, bias, res, bias2)
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last) or x.dtype != torch.float32:
        raise RuntimeError("bias_act_nhwc: x must be a float32 channels-last (N,C,H,W) tensor")
    if res is not None and (res.shape != x.shape or not res.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("bias_act_nhwc: the residual must have the shape and layout of x")
    y = x if out is None else out
    check(lib().mvk_bias_act_nhwc(_p(x), _p(bias), _p(res), _p(bias2), _p(y), x.numel(), x.shape[1], int(bool(relu)),
                                  _stream()))
    return y


# --------------------------------------------------------------------------------------------
# sphere extraction + sampling potentials
# --------------------------------------------------------------------------------------------

def ball_query(points, center, radius, return_d2=False):
    """Indices (ascending, int64, HBM) of the points of `points` [N,3] f32 within `radius` of `center`
    (3 host floats), float64 membership like KDTree.query_radius (ScanNet_sphere_color.py:592-597).
    Synchronises once (the count comes back to size the result)."""
    _dev(points)
    pts = _f32c(points)
    N = pts.shape[0]
    c = _np.ascontiguousarray(_np.asarray(center, dtype=_np.float64).reshape(3))
    idx = torch.empty((max(N, 1),), device=pts.device, dtype=torch.int64)
    d2 = torch.empty((max(N, 1),), device=pts.device, dtype=torch.float64) if return_d2 else None
    cnt = torch.zeros((1,), device=pts.device, dtype=torch.int64)
    ws = _workspace("ball", lib().mvk_ball_query_workspace(N), pts.device)
    check(lib().mvk_ball_query(_p(pts), N, c.ctypes.data_as(C.c_void_p), float(radius), _p(idx), _p(d2), _p(cnt),
                               _p(ws), ws.numel(), _stream()))
    n = int(cnt.item())
    return (idx[:n], d2[:n]) if return_d2 else idx[:n]


def tukey_update(points, center, radius, potentials):
    """potentials (float64, HBM) += Tukey weights of the ball around center (ScanNet_sphere_color.py:576-582)."""
    _dev(points, potentials)
    if potentials.dtype != torch.float64 or not potentials.is_contiguous():
        raise RuntimeError("tukey_update: potentials must be a contiguous float64 tensor")
    pts = _f32c(points)
    c = _np.ascontiguousarray(_np.asarray(center, dtype=_np.float64).reshape(3))
    check(lib().mvk_tukey_update(_p(pts), pts.shape[0], c.ctypes.data_as(C.c_void_p), float(radius), _p(potentials),
                                 _stream()))
    return potentials
