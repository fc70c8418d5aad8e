"""GPU: the multi-stage golden fixtures captured from the reference -- full pyramid (G3), the
reference's own KPFCNN with its state dict (G5: checkpoint compatibility + logits/loss/grads), and
the 2D->3D fusion chain (G6: unprojection, scikit-learn 3-NN, group_points, FeatureAggregation)."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from util import referee_check, REFEREE_FACTOR, REFEREE_FLOOR, bits_equal, rel_err, check_err
from test_oracle_vs_golden import check_pyramid, g5_config, g5b_config, g5_batch, _Cfg, g12_inputs, g12_check_gradients

pytestmark = pytest.mark.gpu
G5_GRAD_TOL = 1e-3      # round 3: ~3x the measured worst case (2.9e-4 on G5, 5.6e-5 on G5b; profiles/r03_parity_errors.txt); was 2e-3
PKG = "enhancing-3d-point-cloud-segmentation-using-multi-modal-fusion-with-2d-images_amd"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_pyramid_device_vs_reference_golden():
    common = importlib.import_module(PKG + ".dropin.datasets.common")
    g = load_golden("g3_pyramid")
    for dt in (torch.int64, torch.int32):
        pyr = common.segmentation_inputs_sphere(_Cfg, T(g["points0"]), g["lens0"], list(g["limits"]), dt,
                                                rotations=list(g["rotations"]))
        assert pyr["neighbors"][0].dtype == dt
        check_pyramid([p.cpu().numpy() for p in pyr["points"]], [l.numpy() for l in pyr["lengths"]],
                      [a.cpu().numpy().astype(np.int32) for a in pyr["neighbors"]],
                      [a.cpu().numpy().astype(np.int32) for a in pyr["pools"]],
                      [a.cpu().numpy().astype(np.int32) for a in pyr["upsamples"]], g)


def test_reference_kpfcnn_state_dict_runs_on_the_hip_path():
    """Loads the REFERENCE network's state dict into the drop-in KPFCNN (same parameter names) and
    reproduces the reference's logits / loss / gradients."""
    arch = importlib.import_module(PKG + ".dropin.models.architectures")
    common = importlib.import_module(PKG + ".dropin.datasets.common")
    g = load_golden("g5_kpfcnn")
    cfg = g5_config()
    np.random.seed(0)
    net = arch.KPFCNN(cfg, list(range(20)), []).cuda()
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    missing = net.load_state_dict(sd, strict=True)
    net.train()
    b = g5_batch(g)
    pyr = dict(points=[t.cuda() for t in b["points"]], neighbors=[t.cuda() for t in b["neighbors"]],
               pools=[t.cuda() for t in b["pools"]], upsamples=[t.cuda() for t in b["upsamples"]],
               lengths=[torch.tensor([t.shape[0]], dtype=torch.int32) for t in b["points"]])
    batch = common.SphereBatch(pyr, b["labels"].cuda(), features=b["features"].cuda())
    out = net(batch, cfg)
    loss = net.loss(out, batch.labels)
    loss.backward()
    check_err("G5 KPFCNN logits vs reference", rel_err(out.detach().cpu().numpy(), g["logits"]), 1e-4)
    check_err("G5 KPFCNN loss vs reference (abs)", abs(loss.item() - float(g["loss"])), 1e-5)
    named = dict(net.named_parameters())
    worst = max((rel_err(named[k[5:]].grad.cpu().numpy(), g[k]), k) for k in g if k.startswith("grad/"))
    # parameter gradients through 5 levels of train-mode BatchNorm (the reference's own float32 run is the fixture)
    check_err("G5 KPFCNN worst parameter gradient (%s)" % worst[1][5:], worst[0], G5_GRAD_TOL)
    # the float64 referee: the HIP path is no further from the float64 network than the reference's own float32 run
    r = load_golden("g14_f64_referee")
    bad = []
    referee_check("G5 logits", out.detach().cpu().numpy(), g["logits"], r["g5/logits"], failures=bad)
    for k in sorted(k for k in g if k.startswith("grad/")):
        referee_check("G5 grad %s" % k[5:], named[k[5:]].grad.cpu().numpy(), g[k], r["g5/" + k], failures=bad)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("name", ["g5b_kpfcnn_deform", "g5b_kpfcnn_deform_mod"])
def test_reference_deformable_kpfcnn_state_dict_runs_on_the_hip_path(name):
    """The reference's DEFORMABLE KPFCNN (train_ScanNet_sphere_middle_fusion.py:87-105 architecture,
    modulated or not, non-zero offset_bias) through the drop-in network on the HIP kernels: block wiring,
    deformable operator, p2p_fitting_regularizer (models/architectures.py:20-58), loss and gradients."""
    arch = importlib.import_module(PKG + ".dropin.models.architectures")
    common = importlib.import_module(PKG + ".dropin.datasets.common")
    g = load_golden(name)
    cfg = g5b_config(int(g["modulated"]))
    np.random.seed(0)
    net = arch.KPFCNN(cfg, list(range(20)), []).cuda()
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    net.load_state_dict(sd, strict=True)
    net.train()
    b = g5_batch(g)
    for dt in (torch.int64, torch.int32):
        net.zero_grad(set_to_none=True)
        net.load_state_dict(sd, strict=True)        # running statistics back to the fixture's
        pyr = dict(points=[t.cuda() for t in b["points"]], neighbors=[t.cuda().to(dt) for t in b["neighbors"]],
                   pools=[t.cuda().to(dt) for t in b["pools"]], upsamples=[t.cuda().to(dt) for t in b["upsamples"]],
                   lengths=[torch.tensor([t.shape[0]], dtype=torch.int32) for t in b["points"]])
        batch = common.SphereBatch(pyr, b["labels"].cuda(), features=b["features"].cuda())
        out = net(batch, cfg)
        loss = net.loss(out, batch.labels)
        loss.backward()
        tag = "%s %s" % (name, "i64" if dt == torch.int64 else "i32")
        check_err("G5b %s logits vs reference" % tag, rel_err(out.detach().cpu().numpy(), g["logits"]), 1e-4)
        check_err("G5b %s output loss (abs)" % tag, abs(net.output_loss.item() - float(g["output_loss"])), 1e-5)
        check_err("G5b %s regulariser loss (rel)" % tag, abs(net.reg_loss.item() - float(g["reg_loss"])) / float(g["reg_loss"]), 1e-4)
        check_err("G5b %s total loss (rel)" % tag, abs(loss.item() - float(g["loss"])) / float(g["loss"]), 1e-4)
        named = dict(net.named_parameters())
        worst = max((rel_err(named[k[5:]].grad.cpu().numpy(), g[k]), k) for k in g if k.startswith("grad/"))
        # the fixture's weights were drawn so that no LeakyReLU input lies within 5e-6 of its tensor's scale from the
        # kink (make_golden.py seed loop): on such inputs the two float32 paths take different slopes and the END-TO-END
        # gradient is discontinuous -- what is bounded here is rounding, not that discontinuity
        check_err("G5b %s worst parameter gradient (%s)" % (tag, worst[1][5:]), worst[0], G5_GRAD_TOL)
        r = load_golden("g14_f64_referee")
        rt = "g5b_deform_mod" if int(g["modulated"]) else "g5b_deform"
        bad = []
        referee_check("G5b %s logits" % tag, out.detach().cpu().numpy(), g["logits"], r[rt + "/logits"], failures=bad)
        for k in sorted(k for k in g if k.startswith("grad/")):
            referee_check("G5b %s grad %s" % (tag, k[5:]), named[k[5:]].grad.cpu().numpy(), g[k], r["%s/%s" % (rt, k)], failures=bad)
        assert not bad, "\n".join(bad)


def referee_digest(label, names, hip_flat, idx_of, ref_of, f64_of, loose=1e-2):
    """The float64 referee on a gradient DIGEST (fixed elements + norm per parameter tensor, fixtures G12 / G13): per tensor
    e = ||digest - float64 digest|| / float64 norm + |norm / float64 norm - 1| for the HIP path and for the float32 fixture
    (the reference's classes for G12, the CPU port for G13). ROUNDING is bounded tightly: e_hip <= REFEREE_FACTOR x e_ref +
    REFEREE_FLOOR. What rounding cannot explain is the LeakyReLU kink: an activation input within float32 rounding of
    zero takes the other slope in another summation order, which moves the gradient of its layer by a quantum of ~1 /
    rows (1e-3 at a few hundred rows, 4.5e-2 at the 22 rows of G12's coarsest level: the callers pass `loose` = 1 / rows of
    their coarsest level) and, through the backward, EVERY tensor upstream of it in that branch. Measured
    (tools/debug_g12_middle.py, round 5): three runs of the same command on G12 middle put the whole 2D encoder branch at
    3.4e-3, 4.6e-6 and 1.6e-3 from float64 -- the split products' float atomics add in a run-dependent order --, the
    deterministic mode at 3.9e-6, and the reference's own float32 fixture has the same quanta elsewhere (G12 early:
    7.9e-3 in one tensor). Hence: every tensor within max(`loose`, REFEREE_FACTOR x e_ref); the tensors within the tight
    rounding bound are held to it, the largest offender and the fraction of offenders are logged with both columns. A
    wiring error shows in every run, in all tensors downstream of it, far above the quantum. (Tensors that are analytically ~0 -- a bias in front
    of a BatchNorm -- are bounded absolutely by the callers.)"""
    scale = max(f64_of(n)[1] for n in names)
    rows, over = [], []
    for n in names:
        v64, n64 = f64_of(n)
        if n64 < 1e-3 * scale:
            continue
        got, idx = hip_flat(n), idx_of(n)
        v32, n32 = ref_of(n)
        e_hip = float(np.linalg.norm(got[idx] - v64) / n64) + abs(float(np.linalg.norm(got)) / n64 - 1.0)
        e_ref = float(np.linalg.norm(np.asarray(v32, np.float64) - v64) / n64) + abs(n32 / n64 - 1.0)
        rows.append((e_hip - (REFEREE_FACTOR * e_ref + REFEREE_FLOOR), e_hip, e_ref, n))
        if rows[-1][0] > 0:
            over.append(rows[-1][1:])
    assert rows
    tight = [r for r in rows if r[0] <= 0]
    if tight:
        _, e_hip, e_ref, n = max(tight)
        check_err("%s: tensor closest to its rounding bound (%s): HIP vs float64 (float32 fixture vs float64: %.3e)"
                  % (label, n, e_ref), e_hip, REFEREE_FACTOR * e_ref + REFEREE_FLOOR)
    if over:
        e_hip, e_ref, n = max(over)
        check_err("%s: largest kink quantum among the %d tensors beyond the rounding bound (%s): HIP vs float64 (float32 "
                  "fixture vs float64: %.3e)" % (label, len(over), n, e_ref), e_hip, max(loose, REFEREE_FACTOR * e_ref))
        for e_hip, e_ref, n in over:
            assert e_hip <= max(loose, REFEREE_FACTOR * e_ref), (n, e_hip, e_ref)
    # (logged, not bounded: one flip near the head reaches every tensor upstream of it -- measured 0 % in most runs and
    # 93 % in one run of G12 middle, the same command giving 0 % the next time)
    check_err("%s: fraction of the %d tensors beyond the rounding bound (logged; a kink flip reaches everything upstream)"
              % (label, len(rows)), len(over) / len(rows), 1.01)
    check_err("%s: the float32 FIXTURE's own largest digest distance from float64 (for scale)" % label, max(r[2] for r in rows), loose)


@pytest.mark.parametrize("variant", ["early", "middle", "late"])
def test_fusion_networks_vs_reference_forward_texts(variant):
    """a16 on the HIP path: the drop-in early / middle / late fusion networks against fixture G12 = the reference's own
    `KPFCNN_featureAggre.forward` texts executed over the reference's blocks (see make_golden.g12_fusion_wirings for the
    three substituted names). Two ragged spheres, so the per-sphere group_points loop and the BatchNorm over the stacked
    batch are exercised; the fixed feature map enters as `batch.feature_2d` (the frozen encoder's output)."""
    syn = importlib.import_module(PKG + ".synthetic")
    common = importlib.import_module(PKG + ".dropin.datasets.common")
    g = load_golden("g12_fusion_wirings")
    cfg, sd, b = g12_inputs(g, variant)
    np.random.seed(0)
    net = syn.build_model(cfg, torch.device("cuda:0"))
    res = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and all(k.startswith("net_2d.") for k in res.missing_keys), res
    net.train()
    for m in net.net_2d._modules.values():
        m.train(False)
    for dt in (torch.int64, torch.int32):
        net.zero_grad(set_to_none=True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        pyr = dict(points=[t.cuda() for t in b["points"]], neighbors=[t.cuda().to(dt) for t in b["neighbors"]],
                   pools=[t.cuda().to(dt) for t in b["pools"]], upsamples=[t.cuda().to(dt) for t in b["upsamples"]],
                   lengths=[torch.from_numpy(l) for l in b["lengths"]])
        batch = common.SphereBatch(pyr, b["labels"].cuda(), feature_3d=b["feature_3d"].cuda(),
                                   feat_aggre_points=b["feat_aggre_points"].cuda(), image_xyz=b["image_xyz"].cuda(),
                                   images=b["images"].cuda(), knn_list=[k.numpy() for k in b["knn_list"]])
        batch.feature_2d = b["feature_2d"].cuda()
        out = net(batch, cfg)
        loss = net.loss(out, batch.labels)
        loss.backward()
        tag = "%s %s" % (variant, "i64" if dt == torch.int64 else "i32")
        check_err("G12 %s logits vs the reference's forward" % tag, rel_err(out.detach().cpu().numpy(), g[variant + "/logits"]), 1e-4)
        check_err("G12 %s loss (abs)" % tag, abs(loss.item() - float(g[variant + "/loss"])), 1e-5)
        grads = {n: p.grad.cpu().numpy() for n, p in net.named_parameters() if p.grad is not None}
        assert any(k.startswith("feat_aggreg.") for k in grads) == (variant == "late")
        # what one LeakyReLU / max-pool kink flip can move: the coarsest level of this batch has 22 rows (6 in one sphere), a
        # flip there is 1 / rows of its layer's gradient and reaches every tensor upstream (referee_digest's docstring)
        quantum = 1.0 / min(int(p.shape[0]) for p in b["points"])
        g12_check_gradients(g, variant, grads, tag, max(5e-3, 0.25 * quantum), 2e-3)
        # the float64 referee on the same digest: per parameter the 64 fixed elements and the norm -- the wide bounds above
        # are the REFERENCE's own float32 distance from the float64 network (early: 1.8e-3 in one norm), not the HIP path's
        r = load_golden("g14_f64_referee")
        referee_check("G12 %s logits" % tag, out.detach().cpu().numpy(), g[variant + "/logits"], r["g12/%s/logits" % variant])
        referee_digest("G12 %s" % tag, sorted(k[len(variant) + 7:] for k in g if k.startswith(variant + "/gnorm/")),
                       lambda n: np.asarray(grads[n], np.float64).reshape(-1), lambda n: g["%s/gidx/%s" % (variant, n)],
                       lambda n: (g["%s/gval/%s" % (variant, n)], float(g["%s/gnorm/%s" % (variant, n)])),
                       lambda n: (r["g12/%s/gval/%s" % (variant, n)], float(r["g12/%s/gnorm/%s" % (variant, n)])),
                       loose=max(1e-2, quantum))


@pytest.mark.parametrize("name,variant,deformable,radius", [("g13_early_19k", "early", False, 1.2),
                                                            ("g13_late_deform_mod_55k", "late", True, 1.7)])
def test_full_size_gradients_vs_cpu_port_digest(name, variant, deformable, radius):
    """Gradients at BASELINE's own sizes (configs[2]: 19 464 points, early fusion; configs[4]'s geometry: 55 070 points,
    late fusion, deformable + modulated): the product rebuilds the fixture's batch from the synthetic sphere (HIP
    subsampling, pyramid with the captured rotations and limits, unprojection, 3-NN), runs forward + backward on the
    fixture's weight formula and is held to the CPU port's digest (make_golden.g13_full_size_gradients): loss, 256
    logit rows, and per parameter tensor the gradient norm, the cosine over 256 fixed elements and their error."""
    from util import g13_state, g12_feature_map
    syn = importlib.import_module(PKG + ".synthetic")
    g = load_golden(name)
    dev = torch.device("cuda:0")
    cfg = syn.make_config(variant, deformable=deformable, modulated=deformable)
    sph = syn.raw_sphere(seed=0, radius=radius)
    views = syn.sphere_views(sph, nv=3, h=120, w=160)
    staged = syn.stage_spheres([sph], dev, [views])
    batch, lens = syn.build_batch(cfg, staged, [int(v) for v in g["limits"]], torch.int32, rotations=list(g["rotations"]))
    assert lens[0] == int(g["n_points"])
    batch.feature_2d = torch.from_numpy(g12_feature_map(3, 64, 120, 160, seed=1313)).to(dev)
    np.random.seed(0)
    net = syn.build_model(cfg, dev)
    shapes = {str(n): tuple(int(v) for v in str(s).split(",") if v) for n, s in zip(g["param_names"], g["param_shapes"])}
    kp = {k[3:]: g[k] for k in g if k.startswith("kp/")}
    sd = g13_state(shapes, variant, deformable, kp)
    res = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and all(k.startswith("net_2d.") for k in res.missing_keys), res
    net.train()
    for m in net.net_2d._modules.values():
        m.train(False)
    out = net(batch, cfg)
    loss = net.loss(out, batch.labels)
    loss.backward()
    rows = torch.from_numpy(g["logit_rows"]).to(dev)
    err = np.abs(out.detach()[rows].cpu().numpy().astype(np.float64) - g["logits"]).max() / float(g["logits_absmax"])
    check_err("G13 %s: 256 logit rows vs the CPU port" % name, err, 1e-4)
    check_err("G13 %s: loss (rel)" % name, abs(loss.item() - float(g["loss"])) / max(1.0, abs(float(g["loss"]))), 1e-5)
    names = sorted(k[6:] for k in g if k.startswith("gnorm/"))
    grads = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    assert set(names) == set(grads), set(names) ^ set(grads)
    scale = max(float(g["gnorm/" + n]) for n in names)
    worst = {"norm": 0.0, "cos": 0.0, "elem": 0.0}
    for n in names:
        want_norm = float(g["gnorm/" + n])
        got = grads[n].reshape(-1)
        got_norm = got.double().norm().item()
        if want_norm < 1e-3 * scale:                    # analytically ~0 (a bias in front of a BatchNorm): rounding only
            assert got_norm < 2e-3 * scale, n
            continue
        a = got[torch.from_numpy(g["gidx/" + n]).to(dev)].double().cpu().numpy()
        b = g["gval/" + n].astype(np.float64)
        worst["norm"] = max(worst["norm"], abs(got_norm / want_norm - 1.0))
        worst["cos"] = max(worst["cos"], 1.0 - float(a @ b) / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
        worst["elem"] = max(worst["elem"], np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    check_err("G13 %s: worst per-parameter |gradient norm ratio - 1|" % name, worst["norm"], 1e-2)
    check_err("G13 %s: worst per-parameter 1 - cosine over 256 fixed elements" % name, worst["cos"], 1e-3)
    # Single elements against ANOTHER float32 run are not a meaningful bound (a LeakyReLU input within rounding of zero takes
    # the other slope in another float32 evaluation order: measured 5e-2 between the port and the HIP path). The real bound
    # is the float64 referee (round 5): per tensor the digest's distance from the float64 run of the same network, against
    # the float32 port's own distance from it.
    r = load_golden("g14_f64_referee")
    check_err("G13 %s: loss vs float64 (rel)" % name, abs(loss.item() - float(r[name + "/loss"])) / abs(float(r[name + "/loss"])), 2e-6)
    referee_check("G13 %s 256 logit rows" % name, out.detach()[rows].cpu().numpy(), g["logits"], r[name + "/logits"])
    referee_digest("G13 %s" % name, names, lambda n: grads[n].reshape(-1).double().cpu().numpy(), lambda n: g["gidx/" + n],
                   lambda n: (g["gval/" + n], float(g["gnorm/" + n])),
                   lambda n: (r["%s/gval/%s" % (name, n)], float(r["%s/gnorm/%s" % (name, n)])),
                   loose=max(1e-2, 1.0 / min(int(p.shape[0]) for p in batch.points)))     # (one kink flip at the coarsest level)


def test_fusion_chain_vs_golden():
    ops = importlib.import_module(PKG + ".ops")
    fa_mod = importlib.import_module(PKG + ".dropin.mvpnet.models.mvpnet_3d")
    g = load_golden("g6_fusion")
    nv, h, w = g["depth"].shape
    xyz, valid = ops.unproject_depth(T(g["depth"].astype(np.int16)), g["cam"], T(g["poses"]))
    assert np.array_equal(valid.cpu().numpy(), g["mask"])
    assert rel_err(xyz.cpu().numpy(), g["xyz"]) < 1e-13            # float64; BLAS vs plain evaluation order: last-bit
    # exact 3-NN vs scikit-learn ball_tree, on the reference's own unprojected pixels and on ours
    knn_ref_keys = ops.knn_pixels(T(g["points"]), T(g["xyz"]), T(g["mask"]), k=3)
    assert np.array_equal(knn_ref_keys.cpu().numpy(), g["knn"])
    assert np.array_equal(ops.knn_pixels(T(g["points"]), xyz, valid, k=3).cpu().numpy(), g["knn"])
    index = T(g["knn"]).unsqueeze(0)
    feat2d = T(g["feat2d"]).requires_grad_(True)
    xyz32 = T(np.transpose(g["xyz"].astype(np.float32), (3, 0, 1, 2)).reshape(1, 3, nv * h * w).copy())
    gfeat, gxyz = ops.group_points(feat2d, index), ops.group_points(xyz32, index)
    assert bits_equal(gfeat.detach().cpu().numpy(), g["grouped_feat"]) and bits_equal(gxyz.cpu().numpy(), g["grouped_xyz"])
    fa = fa_mod.FeatureAggregation(64).cuda()
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    fa.load_state_dict(sd, strict=True)
    fa.train()
    tgt = T(g["points"]).t().unsqueeze(0).contiguous()
    gf = gfeat.detach().requires_grad_(True)
    out = fa(gxyz, tgt, gf)
    out.backward(T(g["gout"]))
    assert rel_err(out.detach().cpu().numpy(), g["out_train"]) < 1e-4
    assert rel_err(gf.grad.cpu().numpy(), g["grouped_feat_grad"]) < 1e-4
    assert rel_err(fa.mlp[0].conv.weight.grad.cpu().numpy(), g["w0_grad"]) < 1e-4
    fa2 = fa_mod.FeatureAggregation(64).cuda()
    fa2.load_state_dict(sd, strict=True)
    fa2.eval()
    assert rel_err(fa2(gxyz, tgt, gfeat.detach()).detach().cpu().numpy(), g["out_eval"]) < 1e-4


def test_fused_feature_aggregation_vs_golden():
    """FeatureAggregation.forward_fused (HIP gather kernel + MFMA linear layers) against the reference
    class's outputs and weight gradients (G6)."""
    fa_mod = importlib.import_module(PKG + ".dropin.mvpnet.models.mvpnet_3d")
    g = load_golden("g6_fusion")
    nv, h, w = g["depth"].shape
    feat = T(g["feat2d"].reshape(64, nv, h * w).transpose(1, 0, 2).reshape(nv, 64, h, w).copy())   # (nv,C,h,w)
    xyz32 = T(g["xyz"].astype(np.float32))                                                           # (nv,h,w,3)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    fa = fa_mod.FeatureAggregation(64).cuda()
    fa.load_state_dict(sd, strict=True)
    fa.train()
    out = fa.forward_fused(feat, xyz32, T(g["knn"]), T(g["points"]))           # (np, 64)
    want = np.transpose(g["out_train"][0], (1, 0))                             # (np, 64)
    assert rel_err(out.detach().cpu().numpy(), want) < 1e-4
    out.backward(T(np.transpose(g["gout"][0], (1, 0)).copy()))
    assert rel_err(fa.mlp[0].conv.weight.grad.cpu().numpy(), g["w0_grad"]) < 1e-4
    rm = fa.mlp[0].bn.running_mean.clone()
    fa2 = fa_mod.FeatureAggregation(64).cuda()
    fa2.load_state_dict(sd, strict=True)
    fa2.eval()
    want_eval = np.transpose(g["out_eval"][0], (1, 0))
    assert rel_err(fa2.forward_fused(feat, xyz32, T(g["knn"]), T(g["points"])).detach().cpu().numpy(), want_eval) < 1e-4
    # running statistics moved like BatchNorm2d's would
    fa3 = fa_mod.FeatureAggregation(64).cuda()
    fa3.load_state_dict(sd, strict=True)
    fa3.train()
    idx = T(g["knn"]).unsqueeze(0)
    ops = importlib.import_module(PKG + ".ops")
    f_cm = T(g["feat2d"])
    x_cm = T(np.transpose(g["xyz"].astype(np.float32), (3, 0, 1, 2)).reshape(1, 3, nv * h * w).copy())
    fa3(ops.group_points(x_cm, idx), T(g["points"]).t().unsqueeze(0).contiguous(), ops.group_points(f_cm, idx))
    assert rel_err(rm.cpu().numpy(), fa3.mlp[0].bn.running_mean.cpu().numpy()) < 1e-4


def test_numpy_dropin_extensions_vs_golden():
    """The cpp_wrappers drop-ins (NumPy in / NumPy out, reference signatures) on the reference's goldens."""
    sub = importlib.import_module(PKG + ".dropin.cpp_wrappers.cpp_subsampling.grid_subsampling")
    nbm = importlib.import_module(PKG + ".dropin.cpp_wrappers.cpp_neighbors.radius_neighbors")
    g = load_golden("g1_sub_batch")
    sp, sl = sub.subsample_batch(g["points"], g["lens"], sampleDl=float(g["dl"]), max_p=0, verbose=0)
    assert sp.dtype == np.float32 and sl.dtype == np.int32
    assert bits_equal(sp, g["out_points"]) and np.array_equal(sl, g["out_lens"])
    g = load_golden("g1_sub_feat_lab")
    p, f, l = sub.subsample(g["points"], features=g["features"], classes=g["labels"], sampleDl=float(g["dl"]), verbose=0)
    assert bits_equal(p, g["out_points"]) and bits_equal(f, g["out_features"]) and np.array_equal(l, g["out_labels"])
    assert l.dtype == np.int32
    only_p = sub.subsample(g["points"], sampleDl=float(g["dl"]))
    assert isinstance(only_p, np.ndarray) and bits_equal(only_p, g["out_points"])
    g = load_golden("g2_nb_conv_b1")
    nb = nbm.batch_query(g["queries"], g["supports"], g["q_lens"], g["s_lens"], radius=float(g["radius"]))
    assert nb.dtype == np.int32 and np.array_equal(nb, g["out"])
    with pytest.raises(RuntimeError, match="^Error$"):          # empty result (wrapper.cpp:201-205)
        nbm.batch_query(np.zeros((0, 3), np.float32), g["supports"], [0], g["s_lens"], radius=0.1)


def test_sphere_picking_vs_sklearn_golden():
    """Device-resident potentials sampler against scikit-learn KDTree.query_radius (G7)."""
    sp = importlib.import_module(PKG + ".dropin.datasets.sphere_picking")
    g = load_golden("g7_sphere_picking")
    s = sp.PotentialSphereSampler([g["coarse0"], g["coarse1"]], [g["input0"], g["input1"]], float(g["in_radius"]),
                                  init_potentials=[g["init_pot0"], g["init_pot1"]])
    for it in range(6):
        r = s.pick()
        assert r["cloud_ind"] == int(g["it%d_cloud" % it]) and r["point_ind"] == int(g["it%d_point" % it])
        assert np.array_equal(r["center"], g["it%d_center" % it])
        assert np.array_equal(r["input_inds"].cpu().numpy(), g["it%d_input_inds" % it])
        assert np.array_equal(r["mask_inds"].cpu().numpy(), g["it%d_mask_inds" % it])
        pot = s.potentials[r["cloud_ind"]].cpu().numpy()
        assert np.allclose(pot, g["it%d_pot" % it], rtol=0, atol=1e-15)      # sqrt / divide rounding: last bit at most
    # ball query on an empty set and on everything
    ops = importlib.import_module(PKG + ".ops")
    pts = T(g["input0"])
    assert ops.ball_query(pts, [100, 100, 100], 0.5).numel() == 0
    assert ops.ball_query(pts, [0, 0, 0], 1e3).numel() == pts.shape[0]


def test_voting_metrics_and_frame_selection():
    """SURVEY.md 8f-3/8f-4 helpers: confusion / IoU against the reference's utils.metrics (G8), voting
    update and greedy frame selection against their NumPy restatements, overlap table against brute force."""
    vt = importlib.import_module(PKG + ".dropin.utils.voting")
    syn = importlib.import_module(PKG + ".synthetic")
    g = load_golden("g8_metrics")
    conf = vt.confusion(T(g["true"]), T(g["pred"]), 20)
    assert np.array_equal(conf.cpu().numpy(), g["confusion"])
    assert np.allclose(vt.IoU_from_confusions(conf).cpu().numpy(), g["iou"], rtol=1e-6, atol=1e-9)
    # voting (tester.py:185)
    rng = np.random.default_rng(1)
    tp = rng.random((1000, 20)).astype(np.float32)
    inds = rng.permutation(1000)[:300]
    pr = rng.random((300, 20)).astype(np.float32)
    want = tp.copy()
    want[inds] = 0.95 * want[inds] + (1 - 0.95) * pr
    got = vt.vote_update(T(tp), T(inds), T(pr), 0.95)
    assert np.allclose(got.cpu().numpy(), want, rtol=1e-6)
    # greedy frame selection (ScanNet_sphere_color.py:53-63 restated)
    ov = rng.random((500, 12)) < 0.2
    o = ov.copy()
    ref = []
    for _ in range(3):
        f = int(o.sum(0).argmax())
        ref.append(f)
        o[o[:, f]] = False
    assert vt.select_frames(T(ov), 3) == ref
    # overlap table vs brute force (get_rgbd_overlap_subcloud.py:68-138)
    sph = syn.raw_sphere(seed=2, radius=0.9, density=1500.0)
    views = syn.sphere_views(sph, nv=3, h=30, w=40)
    base = sph["points"][rng.permutation(sph["points"].shape[0])[:800]]
    got = vt.frame_overlaps(T(base), T(views["depth"].astype(np.int16)), views["cam"], T(views["poses"])).cpu().numpy()
    from oracle import npref
    xyz, mask = npref.unproject_frames(views["cam"], views["depth"], views["poses"])
    want = np.zeros((800, 3), bool)
    for f in range(3):
        pix = xyz[f].reshape(-1, 3)[mask[f].reshape(-1)]
        d2 = ((pix[:, None, :] - base[None].astype(np.float64)) ** 2).sum(-1)
        nn = d2.argmin(1)
        ok = d2[np.arange(len(nn)), nn] <= 0.01
        want[nn[ok], f] = True
    assert np.array_equal(got, want)


def test_unprojection_and_frame_selection_vs_reference_functions():
    """a12 / f3 on the HIP path against G11 (the reference's own depth2xyz / select_frames / unproject function texts
    executed in the build container): mvk_unproject_depth within 1e-13 of the float64 values (evaluation order of the
    3x3 products: last bit), identical valid mask; the overlap table's unprojected points; greedy frame selection on
    device tensors."""
    ops = importlib.import_module(PKG + ".ops")
    vt = importlib.import_module(PKG + ".dropin.utils.voting")
    g = load_golden("g11_unproject_select")
    xyz, valid = ops.unproject_depth(T(g["depth"].astype(np.int16)), g["cam"], T(g["poses"]))
    assert xyz.dtype == torch.float64 and np.array_equal(valid.cpu().numpy(), g["mask"])
    err = rel_err(xyz.cpu().numpy(), g["xyz"])
    print("unprojection rel err vs the reference function: %.2e" % err)
    assert err < 1e-13
    pts = xyz.reshape(-1, 3)[valid.reshape(-1)].cpu().numpy()
    assert rel_err(pts, g["overlap_points"]) < 1e-13
    for i in range(4):
        want = g["selected%d" % i].tolist()
        assert vt.select_frames(T(g["table%d" % i]), len(want)) == want


def test_frame_overlap_table_at_the_reference_size_vs_scipy():
    """f3 at the size the reference's script uses (datasets/get_rgbd_overlap_subcloud.py:68-138: 6 000 base points,
    depth maps resized to 80 x 60): `frame_overlaps` against an independent nearest-neighbour search --
    scipy.spatial.cKDTree.query(k = 1, distance_upper_bound = 0.1) in float64 on the unprojected pixels of G11's kind of
    frames -- standing for open3d's `search_hybrid_vector_3d(p, 0.1, 1)` (nearest neighbour within a radius; open3d is
    not installed here: the hybrid search's own boundary convention at exactly 0.1 stays unpinned)."""
    import mvkpconv
    from scipy.spatial import cKDTree
    from oracle import npref
    syn = mvkpconv.sub("synthetic")
    vt = importlib.import_module(PKG + ".dropin.utils.voting")
    rng = np.random.default_rng(33)
    sph = syn.raw_sphere(seed=5, radius=2.0, density=3000.0)
    views = syn.sphere_views(sph, nv=8, h=120, w=160)
    depth = np.ascontiguousarray(views["depth"][:, ::2, ::2])                    # nearest-neighbour resize to 80 x 60 (:107)
    cam = views["cam"].copy()
    cam[0] /= 2                                                                  # intrinsics follow the resize (:87-88)
    cam[1] /= 2
    sub = sph["points"][rng.permutation(sph["points"].shape[0])[:6000]]          # num_base_pts = 6000 (:69)
    got = vt.frame_overlaps(T(sub), T(depth.astype(np.int16)), cam, T(views["poses"])).cpu().numpy()
    xyz, mask = npref.unproject_frames(cam, depth, views["poses"])
    tree = cKDTree(sub.astype(np.float64))
    want = np.zeros((6000, 8), bool)
    near_boundary = 0
    for f in range(8):
        pix = xyz[f].reshape(-1, 3)[mask[f].reshape(-1)]
        d, nn = tree.query(pix, k=1, distance_upper_bound=0.1)
        ok = np.isfinite(d)
        want[nn[ok], f] = True
        near_boundary += int((np.abs(d[ok] - 0.1) < 1e-6).sum())
    assert want.sum() > 500 and near_boundary == 0          # the fixture does not sit on the radius boundary
    assert np.array_equal(got, want)
    sel = vt.select_frames(T(got), 3)
    o, ref = want.copy(), []
    for _ in range(3):
        fidx = int(o.sum(0).argmax())
        ref.append(fidx)
        o[o[:, fidx]] = False
    assert sel == ref


def test_scene_load_subsampling_vs_reference_core():
    """datasets/scene_cache.subsample_scene (ScanNet_sphere_color.py:935-948: colours as features, labels,
    dl = 0.04) against the compiled reference core (G9): points, order, colour barycentres / 255, majority labels."""
    sc = importlib.import_module(PKG + ".dropin.datasets.scene_cache")
    g = load_golden("g9_ply")
    out = sc.subsample_scene({'points': g["scene_points"], 'colors': g["scene_colors"], 'seg_label': g["scene_labels"]}, 0.04)
    assert bits_equal(out['sub_points'], g["sub_points"])
    assert bits_equal(out['sub_colors'], g["sub_colors"]) and out['sub_colors'].dtype == np.float32
    assert np.array_equal(out['sub_labels'], g["sub_labels"])
    remap = np.arange(21)[::-1].copy()
    out2 = sc.subsample_scene({'points': g["scene_points"], 'colors': g["scene_colors"], 'seg_label': g["scene_labels"]},
                              0.04, label_map=remap)
    assert np.array_equal(out2['sub_labels'], remap[g["sub_labels"]])


def test_reprojection_indices_vs_sklearn_kdtree_golden(tmp_path):
    """proj_inds of a scene on the exact 1-NN kernel against scikit-learn's KDTree.query (G10, the
    reference's call at ScanNet_sphere_color.py:1087-1089), then the <scan>_proj.pkl / <scan>_KDTree.pkl files in
    the reference's schemas (a pickled sklearn KDTree answers the same query)."""
    import pickle
    sc = importlib.import_module(PKG + ".dropin.datasets.scene_cache")
    g = load_golden("g10_reprojection")
    proj = sc.reprojection_indices(g["sub_points"], g["points"])
    assert proj.dtype == np.int32 and np.array_equal(proj, g["proj_inds"])
    labels = np.arange(g["points"].shape[0]) % 21
    sc.save_projection(str(tmp_path), "scene0000_00", proj, labels)
    p2, l2 = sc.load_projection(str(tmp_path), "scene0000_00")
    assert np.array_equal(p2, proj) and np.array_equal(l2, labels)
    sc.save_search_tree(str(tmp_path), "scene0000_00", g["sub_points"])
    with open(sc.scene_paths(str(tmp_path), "scene0000_00")["kdtree"], "rb") as f:
        tree = pickle.load(f)
    assert np.array_equal(np.squeeze(tree.query(g["points"][:500], return_distance=False)), proj[:500])
    assert np.array_equal(np.asarray(tree.data, dtype=np.float32), g["sub_points"])
