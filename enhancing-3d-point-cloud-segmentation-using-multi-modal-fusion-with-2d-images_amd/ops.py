"""torch.autograd wrappers over the C ABI (include/mvkpconv.h).

PyTorch only owns the HBM buffers and the HIP stream; all compute below is in libmvkpconv.so.
Every function requires CUDA(HIP) tensors and raises otherwise -- there is no CPU path here.
"""
import ctypes as C

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
# raw kernels
# --------------------------------------------------------------------------------------------

def gemm(A, B, transA=False, transB=False, out=None, accumulate=False, split_k=1):
    """C = op(A) @ op(B) on v_mfma_f32_32x32x2_f32 (mvk_gemm_f32)."""
    _dev(A, B)
    A, B = _f32c(A), _f32c(B)
    M, Kd = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    N = B.shape[0] if transB else B.shape[1]
    assert (B.shape[1] if transB else B.shape[0]) == Kd, "gemm: inner dimensions differ"
    if out is None:
        out = torch.zeros if split_k > 1 else torch.empty
        out = out((M, N), device=A.device, dtype=torch.float32)
    check(lib().mvk_gemm_f32(_p(A), _p(B), _p(out), M, N, Kd, int(transA), int(transB), int(accumulate),
                             int(split_k), _stream()))
    return out


def kpconv_gather(q, s, idx, x, kp, extent, influence="linear", aggregation="sum", offsets=None,
                  want_min_d2=False):
    """A[n,k,c] = sum_h w[n,h,k] x+[idx[n,h],c]; returns (A, min_d2 or None)."""
    _dev(q, s, idx, x, kp, offsets)
    q, s, x, kp = _f32c(q), _f32c(s), _f32c(x), _f32c(kp)
    idx, i64 = _idx(idx)
    Nq, Ns, H, Cin, K = q.shape[0], s.shape[0], idx.shape[1] if idx.dim() == 2 else 0, x.shape[1], kp.shape[0]
    if x.shape[0] != Ns:
        raise RuntimeError("kpconv: features and support points differ in length")
    A = torch.empty((Nq, K, Cin), device=q.device, dtype=torch.float32)
    min_d2 = None
    if offsets is not None:
        offsets = _f32c(offsets)
        if want_min_d2:
            min_d2 = torch.empty((Nq, K), device=q.device, dtype=torch.float32)
    check(lib().mvk_kpconv_gather_fwd(_p(q), Nq, _p(s), Ns, _p(idx), i64, H, _p(x), Cin, _p(kp), K,
                                      float(extent), INFLUENCE[influence], AGGREGATION[aggregation],
                                      _p(offsets), _p(min_d2), _p(A), _stream()))
    return A, min_d2


def kpconv_scatter(q, s, idx, dA, kp, extent, influence="linear", aggregation="sum", x=None,
                   offsets=None, g_min_d2=None):
    """dx[idx[n,h],c] += sum_k w[n,h,k] dA[n,k,c]; returns (dx, d_offsets or None)."""
    _dev(q, s, idx, dA, kp)
    q, s, dA, kp = _f32c(q), _f32c(s), _f32c(dA), _f32c(kp)
    idx, i64 = _idx(idx)
    Nq, Ns, H, K, Cin = q.shape[0], s.shape[0], idx.shape[1], kp.shape[0], dA.shape[2]
    dx = torch.zeros((Ns, Cin), device=q.device, dtype=torch.float32)
    d_off = None
    if offsets is not None:
        offsets, x = _f32c(offsets), _f32c(x)
        d_off = torch.zeros((Nq, K, 3), device=q.device, dtype=torch.float32)
        if g_min_d2 is not None:
            g_min_d2 = _f32c(g_min_d2)
    check(lib().mvk_kpconv_scatter_bwd(_p(q), Nq, _p(s), Ns, _p(idx), i64, H, Cin, _p(kp), K, float(extent),
                                       INFLUENCE[influence], AGGREGATION[aggregation], _p(dA), _p(dx),
                                       _p(x), _p(offsets), _p(g_min_d2), _p(d_off), _stream()))
    return dx, d_off


def _split_for(n_red, tiles):
    """split-K factor for reductions over the point axis: fill ~2 waves of workgroups per CU."""
    want = max(1, (512 + tiles - 1) // max(tiles, 1))
    return int(max(1, min(want, (n_red + 255) // 256)))


# --------------------------------------------------------------------------------------------
# KPConv (rigid and deformable) as one autograd node
# --------------------------------------------------------------------------------------------

class _KPConvFn(torch.autograd.Function):
    """y = KPConv(q, s, idx, x; kp, W [, offsets, modulations]); also returns min_d2 when deformable.

    Mirrors the tensor algebra of the reference's KPConv.forward
    (KPConv-PyTorch/models/blocks.py:277-374) and its autograd backward."""

    @staticmethod
    def forward(ctx, q, s, idx, x, kp, W, offsets, modulations, extent, influence, aggregation):
        K, Cin, Cout = W.shape
        deform = offsets is not None
        A, min_d2 = kpconv_gather(q, s, idx, x, kp, extent, influence, aggregation, offsets, want_min_d2=deform)
        Am = A * modulations.unsqueeze(2) if modulations is not None else A     # blocks.py:366-367
        y = gemm(Am.view(-1, K * Cin), W.reshape(K * Cin, Cout))                # blocks.py:370-374
        ctx.save_for_backward(q, s, idx, x, kp, W, A, offsets, modulations)
        ctx.cfg = (extent, influence, aggregation)
        return y, min_d2

    @staticmethod
    def backward(ctx, gy, g_min_d2):
        q, s, idx, x, kp, W, A, offsets, modulations = ctx.saved_tensors
        extent, influence, aggregation = ctx.cfg
        K, Cin, Cout = W.shape
        Nq = q.shape[0]
        gy = _f32c(gy)
        Am = A * modulations.unsqueeze(2) if modulations is not None else A
        dW = dx = d_off = d_mod = None
        if ctx.needs_input_grad[5]:
            tiles = ((K * Cin + 63) // 64) * ((Cout + 63) // 64)
            dW = gemm(Am.view(Nq, K * Cin), gy, transA=True, split_k=_split_for(Nq, tiles)).view(K, Cin, Cout)
        need_dA = ctx.needs_input_grad[3] or (offsets is not None)
        if need_dA:
            dAm = gemm(gy, W.reshape(K * Cin, Cout), transB=True).view(Nq, K, Cin)
            if modulations is not None:
                if ctx.needs_input_grad[7]:
                    d_mod = (dAm * A).sum(dim=2)
                dA = dAm * modulations.unsqueeze(2)
            else:
                dA = dAm
            dx, d_off = kpconv_scatter(q, s, idx, dA, kp, extent, influence, aggregation, x=x,
                                       offsets=offsets,
                                       g_min_d2=g_min_d2 if offsets is not None else None)
        return None, None, None, dx, None, dW, d_off, d_mod, None, None, None


def kpconv(q, s, idx, x, kp, W, extent, influence="linear", aggregation="sum", offsets=None, modulations=None):
    """Returns (y [Nq,Cout], min_d2 [Nq,K] or None)."""
    if influence not in INFLUENCE:
        raise ValueError("Unknown influence function type (config.KP_influence)")
    if aggregation not in AGGREGATION:
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    return _KPConvFn.apply(q, s, idx, x, kp, W, offsets, modulations, float(extent), influence, aggregation)


# --------------------------------------------------------------------------------------------
# pooling helpers of blocks.py
# --------------------------------------------------------------------------------------------

class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inds):
        _dev(x, inds)
        x = _f32c(x)
        inds, i64 = _idx(inds)
        Nq, H = inds.shape
        out = torch.empty((Nq, x.shape[1]), device=x.device, dtype=torch.float32)
        arg = torch.empty((Nq, x.shape[1]), device=x.device, dtype=torch.int32)
        check(lib().mvk_max_pool_fwd(_p(x), x.shape[0], x.shape[1], _p(inds), i64, Nq, H, _p(out), _p(arg), _stream()))
        ctx.save_for_backward(inds, arg)
        ctx.ns = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        inds, arg = ctx.saved_tensors
        g = _f32c(g)
        dx = torch.zeros((ctx.ns, g.shape[1]), device=g.device, dtype=torch.float32)
        check(lib().mvk_max_pool_bwd(_p(g), _p(arg), _p(inds), int(inds.dtype == torch.int64), inds.shape[0],
                                     inds.shape[1], ctx.ns, g.shape[1], _p(dx), _stream()))
        return dx, None


class _GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inds2d):
        _dev(x, inds2d)
        x = _f32c(x)
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
        g = _f32c(g)
        dx = torch.zeros((ctx.ns, g.shape[1]), device=g.device, dtype=torch.float32)
        check(lib().mvk_gather_rows_bwd(_p(g), _p(inds2d), int(inds2d.dtype == torch.int64), inds2d.shape[0],
                                        ctx.stride, ctx.ns, g.shape[1], _p(dx), _stream()))
        return dx, None


def max_pool(x, inds):
    """blocks.py:94-110 (zero shadow row takes part in the max)."""
    return _MaxPoolFn.apply(x, inds)


def closest_pool(x, inds):
    """blocks.py:79-91 (first column = closest neighbour because rows are sorted)."""
    return _GatherRowsFn.apply(x, inds)
