"""GPU tests of the BatchNorm fold (round 5): statistics FINISHED inside the producing product
(mvk_bn_finish) and the apply pass inside the consuming product's operand load (mvk_a_transform), against the separate
BatchNorm launches and against torch. Reference semantics: KPConv-PyTorch/models/blocks.py:430-467 (BatchNormBlock),
:621-649 (ResnetBottleneckBlock: batch_norm_conv + LeakyReLU + unary2 + join). Floating point: 1e-5 on activations (the
fold evaluates (x - mean) * (invstd * gamma) + beta, the separate launch ((x - mean) * invstd) * gamma + beta), 1e-4
(north_star) on gradients."""
import importlib

import numpy as np
import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu

PKG = "enhancing-3d-point-cloud-segmentation-using-multi-modal-fusion-with-2d-images_amd"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    m = importlib.import_module(PKG + ".ops")
    old = m.BN_FOLD
    m.BN_FOLD = True            # the fold is off by default (measured slower at one sphere): tested all the same
    yield m
    m.BN_FOLD = old


def _bn(D, seed=0):
    g = torch.Generator().manual_seed(seed)
    bn = torch.nn.BatchNorm1d(D, momentum=0.02).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(D, generator=g) + 0.5)
        bn.bias.copy_(torch.rand(D, generator=g) - 0.5)
    return bn


def _twin(bn):
    ref = torch.nn.BatchNorm1d(bn.num_features, momentum=bn.momentum).cuda()
    ref.load_state_dict(bn.state_dict())
    return ref


@pytest.mark.parametrize("M,N,K,n", [(19464, 32, 480, 19000), (19464, 64, 990, 19464), (4300, 256, 64, 4211), (1300, 128, 96, 1207),
                                     (300, 64, 128, 300), (85, 512, 64, 70), (64, 32, 32, 1), (5000, 200, 64, 4999)])
def test_producer_finishes_the_statistics(ops, M, N, K, n):
    """linear(..., bn=bn): the product's last-arriving workgroups write mean / invstd, update the running statistics and
    the batch counter exactly like the separate statistics pass (torch's BatchNorm1d over the valid rows is the referee),
    the counters are back at zero (three calls in a row), and bn_lrelu on the result only applies."""
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device="cuda")
    x[:, 0] = 25.0                                   # |mean| >> sigma: the merge must not cancel
    W = torch.randn(N, K, device="cuda") * 0.2
    nv = torch.tensor([n], dtype=torch.int32, device="cuda")
    bn = _bn(N, 1)
    ref = _twin(bn)
    if ops.gemm_plan(M, N, K, None, True)[1] == 0:
        # the plan prefers a split reduction for this shape (atomics; no statistics epilogue): nothing is finished, the
        # BatchNorm keeps its own statistics pass
        y = ops.linear(x, W, stats_n_valid=nv, bn=bn)
        assert not ops.bn_finished(y)
        out = ops.bn_lrelu(y, nv, bn, slope=0.1)
        want = torch.nn.functional.leaky_relu(ref(y[:n]), 0.1)
        assert rel_err(out[:n].detach().cpu().numpy(), want.detach().cpu().numpy()) < 2e-5
        return
    for it in range(3):
        y = ops.linear(x, W, stats_n_valid=nv, bn=bn)
        assert ops.bn_finished(y), "the product did not finish the statistics"
        mean, invstd = ops.bn_stats_of(y)[2]
        out = ops.bn_lrelu(y, nv, bn, slope=0.1)
        want = torch.nn.functional.leaky_relu(ref(y[:n]), 0.1) if n > 1 else None
        yv = y[:n].double()
        assert rel_err(mean.cpu().numpy(), yv.mean(0).cpu().numpy()) < 1e-5
        var = yv.var(0, unbiased=False)
        assert rel_err(invstd.cpu().numpy(), torch.rsqrt(var + bn.eps).cpu().numpy()) < 1e-4
        assert (out[n:] == 0).all()
        if want is not None:
            assert rel_err(out[:n].detach().cpu().numpy(), want.detach().cpu().numpy()) < 2e-5
            assert rel_err(bn.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy()) < 1e-5
            assert rel_err(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy()) < 1e-4
            assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == it + 1
    assert int(bn._mvk_fin_counters.abs().sum()) == 0


@pytest.mark.parametrize("R,D,n,slope", [(19464, 64, 19000, 0.1), (4300, 128, 4300, 1.0), (700, 32, 650, 0.1), (90, 256, 90, 0.1)])
def test_apply_only_batchnorm_equals_the_full_one(ops, R, D, n, slope):
    """bn_lrelu on an input with finished statistics (forward = the apply pass alone) against bn_lrelu on the same
    values without them: outputs 1e-6, all three gradients 1e-5, with and without a residual addend."""
    torch.manual_seed(R)
    x0 = torch.randn(R, 48, device="cuda")
    W = torch.randn(D, 48, device="cuda") * 0.3
    nv = torch.tensor([n], dtype=torch.int32, device="cuda")
    add = torch.randn(R, D, device="cuda")
    go = torch.randn(R, D, device="cuda")
    res = {}
    for mode in ("fold", "plain"):
        bn = _bn(D, 2)
        full = ops.linear(x0, W, stats_n_valid=nv, bn=bn if mode == "fold" else None)
        stats = ops.bn_stats_of(full)
        assert ops.bn_finished(full) == (mode == "fold")
        y = full.detach().requires_grad_(True)
        if stats is not None:
            y._mvk_bn_stats = stats
        for addend in (None, add):
            out = ops.bn_lrelu(y, nv, bn, slope=slope, addend=addend)
            g = torch.autograd.grad(out, [y, bn.weight, bn.bias], go)
            res[(mode, addend is not None)] = (out.detach(), *g)
    for key in (False, True):
        for a, b in zip(res[("fold", key)], res[("plain", key)]):
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("R,Kd,N,n", [(19464, 32, 128, 19000), (4300, 64, 256, 4300), (1300, 128, 512, 1250), (19464, 64, 128, 19464),
                                      (333, 256, 1024, 300), (2000, 36, 64, 1999)])
def test_batchnorm_inside_the_operand_load(ops, R, Kd, N, n):
    """bn_lrelu_linear(x, ...) = linear(bn_lrelu(x, ...), W): the normalised activations never get a launch of their own
    -- the product applies BatchNorm + LeakyReLU while staging A and writes them on the way. Output, the activation
    tensor (rows >= n_valid zero), the statistics it finishes for the NEXT BatchNorm and every gradient against the
    two-step form."""
    torch.manual_seed(R + Kd)
    x0 = torch.randn(R, 40, device="cuda")
    x0[:, 1] = 9.0
    Wp = torch.randn(Kd, 40, device="cuda") * 0.3                  # the producer of x (stands for the KPConv contraction)
    nv = torch.tensor([n], dtype=torch.int32, device="cuda")
    go = torch.randn(R, N, device="cuda")
    go[n:] = 0                                                      # what the BatchNorm behind the product hands back
    out = {}
    for mode in ("fold", "steps"):
        bn_in, bn_out = _bn(Kd, 3), _bn(N, 4)
        W = torch.nn.Parameter(torch.randn(N, Kd, generator=torch.Generator().manual_seed(5)).cuda() * 0.2)
        xp = torch.nn.Parameter(Wp.clone())
        x = ops.linear(x0, xp, stats_n_valid=nv, bn=bn_in if mode == "fold" else None)
        if mode == "fold":
            res = ops.bn_lrelu_linear(x, nv, bn_in, 0.1, W, stats_n_valid=nv, bn_out=bn_out)
            assert res is not None, "the fold did not apply"
            y, act = res
            assert ops.bn_finished(y)
        else:
            act = ops.bn_lrelu(x, nv, bn_in, slope=0.1)
            y = ops.linear(act, W, stats_n_valid=nv, bn=None)
        z = ops.bn_lrelu(y, nv, bn_out, slope=1.0)
        grads = torch.autograd.grad(z, [xp, W, bn_in.weight, bn_in.bias, bn_out.weight, bn_out.bias], go)
        out[mode] = (y.detach(), act.detach(), z.detach(), [g.detach() for g in grads], bn_in, bn_out)
    yf, af, zf, gf, bi_f, bo_f = out["fold"]
    ys, as_, zs, gs, bi_s, bo_s = out["steps"]
    assert (af[n:] == 0).all() and (yf[n:] == 0).all()
    assert rel_err(af.cpu().numpy(), as_.cpu().numpy()) < 1e-5
    assert rel_err(yf[:n].cpu().numpy(), ys[:n].cpu().numpy()) < 1e-5
    assert rel_err(zf.cpu().numpy(), zs.cpu().numpy()) < 2e-5
    for a, b in zip(gf, gs):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-4
    for a, b in ((bi_f, bi_s), (bo_f, bo_s)):
        assert rel_err(a.running_mean.cpu().numpy(), b.running_mean.cpu().numpy()) < 1e-5
        assert rel_err(a.running_var.cpu().numpy(), b.running_var.cpu().numpy()) < 1e-4
        assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 1


def test_fold_conditions_fall_back_quietly(ops):
    """What the fold does not take returns None / leaves the statistics unfinished, and the two-step path still works:
    an input without finished statistics, a narrow product (<= 32 columns), a reduction longer than 512, eval mode."""
    x = torch.randn(3000, 64, device="cuda")
    nv = torch.tensor([3000], dtype=torch.int32, device="cuda")
    bn = _bn(64)
    W = torch.randn(128, 64, device="cuda")
    assert ops.bn_lrelu_linear(x, nv, bn, 0.1, W) is None                                  # no statistics at all
    y = ops.linear(x, torch.randn(64, 64, device="cuda"), stats_n_valid=nv, bn=bn)
    assert ops.bn_finished(y)
    assert ops.bn_lrelu_linear(y, nv, bn, 0.1, torch.randn(32, 64, device="cuda")) is None   # narrow product
    bn.eval()
    y2 = ops.linear(x, torch.randn(64, 64, device="cuda"), stats_n_valid=nv, bn=bn)
    assert not ops.bn_finished(y2)                                                          # eval mode: nothing to finish
    big = ops.linear(torch.randn(500, 32, device="cuda"), torch.randn(1024, 32, device="cuda"),
                     stats_n_valid=torch.tensor([500], dtype=torch.int32, device="cuda"), bn=_bn(1024))
    assert ops.bn_lrelu_linear(big, nv, _bn(1024), 0.1, torch.randn(64, 1024, device="cuda")) is None   # Kd > 512


def test_linear_pair_finishes_both_statistics(ops):
    """unary1 and the shortcut layer of a bottleneck block as one launch (linear_pair) with both BatchNorms finished."""
    torch.manual_seed(7)
    x = torch.randn(19464, 64, device="cuda")
    W0, W1 = torch.randn(32, 64, device="cuda") * 0.2, torch.randn(128, 64, device="cuda") * 0.2
    nv = torch.tensor([19000], dtype=torch.int32, device="cuda")
    bn0, bn1 = _bn(32, 1), _bn(128, 2)
    r0, r1 = _twin(bn0), _twin(bn1)
    pair = ops.linear_pair(x, W0, W1, nv, bn0=bn0, bn1=bn1)
    assert pair is not None and ops.bn_finished(pair[0]) and ops.bn_finished(pair[1])
    a, b = ops.bn_lrelu_pair(pair[0], bn0, 0.1, pair[1], bn1, 1.0, nv)
    wa = torch.nn.functional.leaky_relu(r0(pair[0][:19000]), 0.1)
    wb = r1(pair[1][:19000])
    assert rel_err(a[:19000].detach().cpu().numpy(), wa.detach().cpu().numpy()) < 2e-5
    assert rel_err(b[:19000].detach().cpu().numpy(), wb.detach().cpu().numpy()) < 2e-5
    for m, r in ((bn0, r0), (bn1, r1)):
        assert rel_err(m.running_var.cpu().numpy(), r.running_var.cpu().numpy()) < 1e-4
        assert int(m.num_batches_tracked) == 1


def test_resnet_block_with_and_without_the_fold(ops):
    """A ResnetBottleneckBlock of the reference's first level (blocks.py:596-649) on a random cloud: the folded wiring
    (finished statistics, batch_norm_conv inside unary2's operand load) against the separate launches -- output 2e-5,
    every parameter gradient 1e-4 relative to its own scale, running statistics equal."""
    blocks = importlib.import_module(PKG + ".dropin.models.blocks")
    cfgm = importlib.import_module(PKG + ".dropin.utils.config")
    common = importlib.import_module(PKG + ".dropin.datasets.common")
    torch.manual_seed(11)
    rng = np.random.default_rng(5)
    pts = torch.from_numpy((rng.random((6000, 3)) * [2.0, 2.0, 1.0]).astype(np.float32)).cuda()
    lens = torch.tensor([6000], dtype=torch.int32)
    nb = ops.radius_neighbors_batch(pts, pts, lens, lens, 0.125, limit=30)

    class Batch:
        points = [pts]
        neighbors = [nb]
        pools = []
        lengths = [lens]
    cfg = cfgm.Config()
    cfg.use_batch_norm = True
    cfg.batch_norm_momentum = 0.02
    outs = {}
    feats = torch.randn(6000, 64, device="cuda")
    go = torch.randn(6000, 128, device="cuda")
    state = None
    for fold in (True, False):
        ops.BN_FOLD = fold
        try:
            blk = blocks.ResnetBottleneckBlock("resnetb", 64, 128, 0.125, 0, cfg).cuda()
            if state is None:
                state = {k: v.clone() for k, v in blk.state_dict().items()}
            blk.load_state_dict(state)
            blk.train()
            x = feats.clone().requires_grad_(True)
            y = blk(x, Batch)
            params = [p for p in blk.parameters() if p.requires_grad]
            grads = torch.autograd.grad(y, [x] + params, go)
            outs[fold] = (y.detach(), [g.detach() for g in grads],
                          {k: v.clone() for k, v in blk.state_dict().items() if "running" in k or "tracked" in k})
        finally:
            ops.BN_FOLD = True      # (the module fixture restores the default at the end)
    assert rel_err(outs[True][0].cpu().numpy(), outs[False][0].cpu().numpy()) < 2e-5
    for a, b in zip(outs[True][1], outs[False][1]):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-4
    for k, v in outs[True][2].items():
        assert rel_err(v.double().cpu().numpy(), outs[False][2][k].double().cpu().numpy()) < 1e-4, k


def test_ordered_split_is_the_in_order_sum_of_its_slices(ops):
    """The parking accesses are compiler-counted buffer loads / stores since round 5 (csrc/common.h; was inline
    assembly): an ordered split product must equal, BIT FOR BIT, the slices' own unsplit products added in
    split order (same k order inside a slice, float addition of the slices in order 0, 1, ...)."""
    ops.set_deterministic(True)
    try:
        torch.manual_seed(3)
        for M, N, K, split in ((19464, 64, 990, 4), (1300, 128, 1920, 6), (85, 512, 7680, 20), (4096, 32, 480, 3), (2000, 256, 1024, 8)):
            A = torch.randn(M, K, device="cuda")
            B = torch.randn(K, N, device="cuda")
            y = ops.gemm(A, B, split_k=split)
            ksteps = (K + 31) // 32
            kps = (ksteps + split - 1) // split * 32
            acc = torch.zeros(M, N, device="cuda")
            k0 = 0
            while k0 < K:
                k1 = min(K, k0 + kps)
                acc = acc + ops.gemm(A[:, k0:k1].contiguous(), B[k0:k1].contiguous(), split_k=1)
                k0 = k1
            assert torch.equal(y, acc), (M, N, K, split, float((y - acc).abs().max()))
    finally:
        ops.set_deterministic(False)


def test_a_tiny_arena_fails_loudly_instead_of_sharing_slices(ops):
    """the arena of the ordered reductions never hands one slice to two launches that may overlap. A product
    larger than the arena, and a capture that would need more than the arena holds, raise; eager launches recycle the
    bottom region after a device synchronisation and keep working."""
    old = ops._SPLIT_ARENA_BYTES
    ops.set_deterministic(False)
    ops._SPLIT_ARENA_BYTES = 8 << 20
    try:
        ops.set_deterministic(True)
        A = torch.randn(19464, 990, device="cuda")
        B = torch.randn(990, 64, device="cuda")
        with pytest.raises(RuntimeError, match="arena"):
            ops.gemm(A, B, split_k=4)                         # 19.9 MB of parking space > 8 MB
        a = torch.randn(2048, 512, device="cuda")
        b = torch.randn(512, 64, device="cuda")
        ref = (a.double() @ b.double()).cpu().numpy()
        first = ops.gemm(a, b, split_k=4)                    # 2 MB per call: wraps every four calls
        for _ in range(12):
            assert torch.equal(ops.gemm(a, b, split_k=4), first)
        assert rel_err(first.cpu().numpy(), ref) < 1e-5
        # captured launches take their slices from the top of the arena and keep them: replays and the eager launches
        # that recycle the bottom region meanwhile never meet
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                captured = ops.gemm(a, b, split_k=4)
        for _ in range(3):
            g.replay()
            for _ in range(5):
                assert torch.equal(ops.gemm(a, b, split_k=4), first)
        torch.cuda.synchronize()
        assert torch.equal(captured, first)
    finally:
        ops.set_deterministic(False)
        ops._SPLIT_ARENA_BYTES = old
