"""CPU-only tests: the C-ABI library loads and exports every symbol include/mvkpconv.h declares (no
compute without a GPU), host logic mirrors the reference (kernel points, rotations, config, error
behaviour), and the N>1 gradient exchange works across 2 gloo ranks."""
import ctypes
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "enhancing-3d-point-cloud-segmentation-using-multi-modal-fusion-with-2d-images_amd"


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mvkpconv.h")).read()
    declared = set(re.findall(r"\b(mvk_[a-z0-9_]+)\s*\(", hdr))
    lib_mod = importlib.import_module(PKG + "._lib")
    assert os.path.exists(lib_mod.LIB_PATH), "run `python __graft_entry__.py build` first"
    raw = ctypes.CDLL(lib_mod.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "libmvkpconv.so lacks %s" % name
    assert declared == set(lib_mod.EXPORTS), "ctypes table and header disagree: %s" % (declared ^ set(lib_mod.EXPORTS))
    l = lib_mod.lib()
    assert l.mvk_abi_version() == lib_mod.ABI_VERSION
    assert l.mvk_grid_subsample_workspace(1000, 2, 3, 1) > 0 and l.mvk_radius_neighbors_workspace(10, 10, 1) > 0


def test_gather_launch_plan_is_a_host_function():
    """mvk_kpconv_gather_plan: the launch geometry of the gather kernel, no GPU call (bench.py finds the launch in the
    PMC profile by it). Round 5: layers of >= 5 channels with linear influence run on the MFMA gather -- one wave per
    point, four per workgroup; first layer of the early-fusion net (66 = 64 + 2 channels): five 16-channel accumulator
    tiles, one channel block; wide rows: blocks of 256 channels in gridDim.y; few points with long rows (coarse levels,
    the deformable layers' deform-radius rows): the four waves of a workgroup share ONE point."""
    ops = importlib.import_module(PKG + ".ops")
    p = ops.kpconv_gather_plan(19464, 19464, 58, 66)
    assert p["mfma"] == 1 and (p["lanes_per_point"], p["points_per_wave"], p["rows_per_batch"]) == (64, 1, 5)
    assert p["workgroups"] == 4866 and p["first_sharing_workgroup"] == -1 and p["waves_per_workgroup"] == 4
    assert p["grid_threads"] == p["workgroups"] * 256
    t = ops.kpconv_gather_plan(19464, 19464, 58, 32)
    assert t["mfma"] == 1 and t["rows_per_batch"] == 2 and t["workgroups"] == 4866
    w = ops.kpconv_gather_plan(1300, 1300, 50, 512)            # two channel blocks of 256; too many points to share
    assert w["mfma"] == 1 and w["rows_per_batch"] == 16 and w["workgroups"] == 325 * 2 and w["first_sharing_workgroup"] == -1
    c = ops.kpconv_gather_plan(180, 180, 51, 256)              # a coarse level: a workgroup per point
    assert c["mfma"] == 1 and c["first_sharing_workgroup"] == 0 and c["workgroups"] == 180
    d = ops.kpconv_gather_plan(750, 750, 420, 128, deformable=True)    # deform-radius rows: every workgroup shares
    assert d["mfma"] == 1 and d["first_sharing_workgroup"] == 0 and d["workgroups"] == 750 and d["waves_per_workgroup"] == 4
    n = ops.kpconv_gather_plan(100, 100, 20, 2)                                # narrow rows: two tiles, mostly idle
    assert n["mfma"] == 1 and n["rows_per_batch"] == 2 and n["workgroups"] == 25
    with pytest.raises(RuntimeError):
        ops.kpconv_gather_plan(100, 100, 20, 64, elem_bytes=2)                # the fp16-feature mode left the tree (round 5)


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, PKG)
    for dp, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# ", ""), \
                    "%s mentions the oracle outside a docstring" % f


def test_ops_refuse_cpu_tensors():
    ops = importlib.import_module(PKG + ".ops")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.group_points(torch.zeros(1, 2, 3), torch.zeros(1, 2, 2, dtype=torch.int64))


def test_kernel_points_match_reference_golden():
    """load_kernels consumes the global RNG like the reference: with the seed of the golden run the
    kernel points are bit-identical to the ones the reference's KPConv created (g4 fixture)."""
    from conftest import load_golden
    kp_mod = importlib.import_module(PKG + ".dropin.kernels.kernel_points")
    g = load_golden("g4_kpconv_config1")
    np.random.seed(0)
    kp = kp_mod.load_kernels(0.1, 15, dimension=3, fixed="center")      # the golden run passed the Python float 0.1
    assert kp.dtype == np.float32 and np.array_equal(kp, g["kernel_points"])


def test_rotations_are_orthonormal_and_match_axis_angle():
    kp_mod = importlib.import_module(PKG + ".dropin.kernels.kernel_points")
    rng = np.random.default_rng(0)
    ax = rng.normal(size=(16, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(0, 2 * np.pi, 16)
    R = kp_mod.create_3D_rotations(ax, ang)
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3), atol=1e-12)
    assert np.allclose(np.einsum("nij,nj->ni", R, ax), ax, atol=1e-12)          # axis is invariant
    assert np.allclose(np.trace(R, axis1=1, axis2=2), 1 + 2 * np.cos(ang))


def test_config_layer_bookkeeping():
    syn_cfg = importlib.import_module(PKG + ".dropin.utils.config")

    class C(syn_cfg.Config):
        architecture = ['simple', 'resnetb_strided', 'resnetb_deformable', 'resnetb_deformable_strided', 'resnetb',
                        'nearest_upsample', 'unary', 'nearest_upsample', 'unary']
    c = C()
    assert c.num_layers == 3 and c.deform_layers == [False, True, False]


def test_unknown_block_and_modes_raise_like_the_reference():
    blocks = importlib.import_module(PKG + ".dropin.models.blocks")
    with pytest.raises(ValueError, match="Unknown block name"):
        blocks.block_decider("nope", 0.1, 4, 8, 0, None)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module(PKG + ".dp")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    frozen = torch.nn.Linear(2, 2)
    for p in frozen.parameters():
        p.requires_grad = False
    x = torch.full((4, 5), float(rank + 1))
    net(x).sum().backward()
    red = dp.FlatAllReduce(list(net.parameters()) + list(frozen.parameters()), world)
    red()
    q.put((rank, [p.grad.numpy().copy() for p in net.parameters()], dp.shard_spheres(5, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_two_gloo_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    # reference: mean of the two ranks' gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    want = None
    for r in range(2):
        net.zero_grad()
        net(torch.full((4, 5), float(r + 1))).sum().backward()
        g = [p.grad.clone() for p in net.parameters()]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    want = [w / 2 for w in want]
    for rank, grads, shard in res:
        for a, b in zip(grads, want):
            assert torch.allclose(torch.from_numpy(a), b, atol=1e-6)
    assert res[0][2] == [0, 2, 4] and res[1][2] == [1, 3]


def test_dropin_wrappers_keep_the_reference_error_strings():
    """Argument validation of the cpp_wrappers drop-ins runs before anything touches the GPU and raises
    RuntimeError with the reference's messages (cpp_subsampling/wrapper.cpp:389-470,
    cpp_neighbors/wrapper.cpp:127-171)."""
    sub = importlib.import_module(PKG + ".dropin.cpp_wrappers.cpp_subsampling.grid_subsampling")
    nb = importlib.import_module(PKG + ".dropin.cpp_wrappers.cpp_neighbors.radius_neighbors")
    with pytest.raises(RuntimeError, match=r"points.shape is not \(N, 3\)"):
        sub.subsample(np.zeros((5, 2), np.float32))
    with pytest.raises(RuntimeError, match=r"points.shape is not \(N, 3\)"):
        sub.subsample_batch(np.zeros((5,), np.float32), np.array([5], np.int32))
    with pytest.raises(RuntimeError, match=r"features.shape is not \(N, d\)"):
        sub.subsample(np.zeros((5, 3), np.float32), features=np.zeros((4, 2), np.float32))
    with pytest.raises(RuntimeError, match="Error parsing method"):
        sub.subsample(np.zeros((5, 3), np.float32), method="median")
    with pytest.raises(TypeError):
        sub.subsample(np.zeros((5, 3), np.float32), np.zeros((5, 1), np.float32))      # options are keyword-only
    with pytest.raises(RuntimeError, match=r"query.shape is not \(N, 3\)"):
        nb.batch_query(np.zeros((5, 4), np.float32), np.zeros((5, 3), np.float32), [5], [5], radius=0.1)
    with pytest.raises(RuntimeError, match="different for queries and supports"):
        nb.batch_query(np.zeros((5, 3), np.float32), np.zeros((5, 3), np.float32), [5], [2, 3], radius=0.1)


def test_kernel_disposition_optimizer_for_uncached_sizes():
    kp_mod = importlib.import_module(PKG + ".dropin.kernels.kernel_points")
    p = kp_mod.optimize_disposition(9, 3, "center", iters=300)
    assert p.shape == (9, 3) and np.allclose(p[0], 0) and np.linalg.norm(p, axis=1).max() <= 0.66 + 1e-9
    d = np.linalg.norm(p[:, None] - p[None], axis=-1) + np.eye(9)
    assert d.min() > 0.2                                   # points repel each other


def test_dropin_resolves_like_the_reference_when_first_on_sys_path():
    """INTEGRATION.md route 1: with dropin/ first on sys.path the reference's own import statements
    (datasets/common.py:32-36, architectures_sphere.py:18-22) resolve to the drop-in modules."""
    code = r'''
import sys
sys.path.insert(0, %r)
import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling
import cpp_wrappers.cpp_neighbors.radius_neighbors as cpp_neighbors
from kernels.kernel_points import create_3D_rotations, load_kernels
from models.blocks import *
from models.architectures import KPFCNN, p2p_fitting_regularizer
from models.architectures_sphere import KPFCNN_featureAggre
from models.architectures_sphere_middle_fusion import KPFCNN_featureAggre as M
from models.architectures_sphere_late_fusion import KPFCNN_featureAggre as L
from mvpnet.models.mvpnet_3d import FeatureAggregation
from mvpnet.models.unet_resnet34 import UNetResNet34
from mvpnet.ops.group_points import group_points
from common.nn import SharedMLP
from datasets.common import grid_subsampling, batch_grid_subsampling, batch_neighbors
assert callable(cpp_subsampling.subsample) and callable(cpp_subsampling.subsample_batch) and callable(cpp_neighbors.batch_query)
assert KPConv.__module__ == "models.blocks" and FeatureAggregation(64).out_channels == 64
print("ok")
''' % os.path.join(ROOT, PKG, "dropin")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


# ------------------------------------------------------------------ on-disk formats

def _dropin(mod):
    import importlib
    return importlib.import_module("enhancing-3d-point-cloud-segmentation-using-multi-modal-fusion-with-2d-images_amd.dropin." + mod)


def test_ply_reads_reference_files_and_writes_the_same_bytes(tmp_path):
    """utils/ply.py drop-in against files written by the reference's write_ply (tests/golden/g9_*.ply):
    same values back, and for the same arrays byte-identical files (cloud with mixed field types, mesh)."""
    from conftest import load_golden
    ply = _dropin("utils.ply")
    g = load_golden("g9_ply")
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data = ply.read_ply(os.path.join(gold, "g9_cloud.ply"))
    assert data.dtype.names == ('x', 'y', 'z', 'red', 'green', 'blue', 'class', 'score')
    assert np.array_equal(data['x'], g["points"][:, 0]) and np.array_equal(data['x'], g["read_x"])
    assert np.array_equal(data['blue'], g["colors"][:, 2]) and np.array_equal(data['class'], g["labels"])
    assert np.array_equal(data['score'], g["score"]) and data['score'].dtype == np.float64
    vdata, faces = ply.read_ply(os.path.join(gold, "g9_mesh.ply"), triangular_mesh=True)
    assert np.array_equal(faces, g["faces"]) and np.array_equal(vdata['red'], g["mesh_red"])
    names = ['x', 'y', 'z', 'red', 'green', 'blue', 'class', 'score']
    out = str(tmp_path / "cloud")                                   # '.ply' is appended like the reference does
    assert ply.write_ply(out, [g["points"], g["colors"], g["labels"], g["score"]], names) is True
    assert open(out + ".ply", "rb").read() == open(os.path.join(gold, "g9_cloud.ply"), "rb").read()
    out = str(tmp_path / "mesh.ply")
    assert ply.write_ply(out, (g["points"], g["colors"]), names[:6], triangular_faces=g["faces"]) is True
    assert open(out, "rb").read() == open(os.path.join(gold, "g9_mesh.ply"), "rb").read()
    # the reference's soft failures
    assert ply.write_ply(out, [g["points"], g["labels"][:5]], ['x', 'y', 'z', 'c']) is False
    assert ply.write_ply(out, [g["points"]], ['x', 'y']) is False
    assert ply.write_ply(out, [np.zeros((3, 2, 2))], ['a']) is False
    bad = tmp_path / "ascii.ply"
    bad.write_text("ply\nformat ascii 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(ValueError):
        ply.read_ply(str(bad))


def test_scene_cache_pickle_schemas_round_trip(tmp_path):
    """preprocess cache (list of scan dicts), per-scene <scan>.pkl and <scan>_proj.pkl in the reference's
    schemas (mvpnet preprocess.py:177-186, ScanNet_sphere_color.py:985-991, :1087-1092)."""
    import pickle
    sc = _dropin("datasets.scene_cache")
    rng = np.random.default_rng(1)
    scans = [{'scan_id': 'scene%04d_00' % i, 'points': rng.random((50, 3)).astype(np.float32),
              'colors': rng.integers(0, 256, (50, 3)).astype(np.uint8), 'seg_label': rng.integers(0, 40, 50)}
             for i in range(3)]
    cache = str(tmp_path / "cache.pkl")
    sc.save_preprocess_cache(cache, scans)
    back = sc.load_preprocess_cache(cache)
    assert [d['scan_id'] for d in back] == [d['scan_id'] for d in scans]
    assert all(np.array_equal(a['colors'], b['colors']) and a['colors'].dtype == np.uint8 for a, b in zip(back, scans))
    with open(cache, 'wb') as f:
        pickle.dump([{'scan_id': 'x', 'points': np.zeros((4, 3), np.float32)}], f)
    with pytest.raises(ValueError):
        sc.load_preprocess_cache(cache)
    tree = str(tmp_path / "input_0.040")
    sub = {'sub_points': scans[0]['points'], 'sub_labels': np.arange(50, dtype=np.int32),
           'sub_colors': (scans[0]['colors'] / 255).astype(np.float32)}
    rgbd = {'scan_id': 'scene0000_00', 'sub_base_point_ind': np.arange(7), 'sub_pointwise_rgbd_overlap': np.ones((7, 50), bool),
            'frame_ids': ['0', '20'], 'cam_matrix': np.eye(4, dtype=np.float32)}
    sc.save_scene(tree, 'scene0000_00', sub, rgbd)
    with open(os.path.join(tree, 'scene0000_00.pkl'), 'rb') as f:       # what the reference's loader does (:912-917)
        raw = pickle.load(f)
    assert set(raw) == {'sub_points', 'sub_labels', 'sub_colors', 'rgbd_dict'} and raw['rgbd_dict']['frame_ids'] == ['0', '20']
    got = sc.load_scene(tree, 'scene0000_00')
    assert np.array_equal(got['sub_points'], sub['sub_points']) and got['rgbd_dict']['scan_id'] == 'scene0000_00'
    sc.save_projection(tree, 'scene0000_00', np.arange(50)[::-1], scans[0]['seg_label'])
    proj, lab = sc.load_projection(tree, 'scene0000_00')
    assert proj.dtype == np.int32 and np.array_equal(proj, np.arange(50)[::-1]) and np.array_equal(lab, scans[0]['seg_label'])


def test_sampler_calibration_controller_and_cache_files(tmp_path):
    """datasets/calibration.py against a literal NumPy replay of ScanNet_sphere_color.py:1380-1464 on a
    simulated sampler (spheres per batch ~ batch_limit / sphere size + noise), and the reference's cache
    file keys."""
    import pickle
    import types
    cal_mod = _dropin("datasets.calibration")
    cfg = types.SimpleNamespace(batch_num=5, in_radius=1.2, first_subsampling_dl=0.04, num_layers=3, conv_radius=2.5,
                                deform_radius=6.0, deform_layers=[False, False, True])
    rng = np.random.default_rng(3)

    def sampler(limit, r):
        b = max(1, int(limit // 20000 + r.integers(-1, 2)))
        mats = []
        for layer in range(3):
            n = 400 >> layer
            counts = r.integers(5, 60 + 100 * (layer == 2), n)
            m = np.full((n, int(counts.max())), n, dtype=np.int64)
            for i, c in enumerate(counts):
                m[i, :c] = r.integers(0, n, c)
            mats.append(m)
        return mats, b

    cal = cal_mod.Calibrator(cfg, batch_limit=50000)
    hist_n = int(np.ceil(4 / 3 * np.pi * (6.0 + 1) ** 3))
    hists = np.zeros((3, hist_n), np.int64)
    estim_b, T, finer, limit, errs, steps = 0.0, 10, False, 50000.0, [], 0
    r1, r2 = np.random.default_rng(9), np.random.default_rng(9)
    while not cal.converged and cal.steps < 5000:
        mats, b = sampler(cal.batch_limit, r1)
        cal.update([torch.from_numpy(m) for m in mats], b)
        mats2, b2 = sampler(limit, r2)                                 # the literal replay, same random stream
        assert b2 == b
        hists += np.vstack([np.bincount(np.sum(m < m.shape[0], axis=1), minlength=hist_n)[:hist_n] for m in mats2])
        estim_b += (b2 - estim_b) / T
        errs = (errs + [5 - estim_b])[-10:]
        limit += 100.0 * (5 - b2)
        if not finer and abs(estim_b - 5) < 1:
            T, finer = 100, True
        steps += 1
        assert limit == cal.batch_limit
        if finer and np.max(np.abs(errs)) < 0.1:
            break
    assert cal.converged and cal.steps == steps and abs(cal.estim_b - 5) < 0.2
    cs = np.cumsum(hists.T, axis=0)
    want = np.sum(cs < 0.9 * cs[hist_n - 1, :], axis=0)
    assert np.array_equal(cal.neighborhood_limits(0.9), want)
    cal_mod.save_calibration(cfg, str(tmp_path), cal.batch_limit, [int(x) for x in want])
    with open(tmp_path / "batch_limits.pkl", "rb") as f:
        assert pickle.load(f) == {"potentials_1.200_0.040_5": float(cal.batch_limit)}
    with open(tmp_path / "neighbors_limits.pkl", "rb") as f:
        assert sorted(pickle.load(f)) == ["0.040_0.100", "0.080_0.200", "0.160_0.960"]
    b, lim = cal_mod.load_calibration(cfg, str(tmp_path))
    assert b == float(cal.batch_limit) and lim == [int(x) for x in want]
    cfg.batch_num = 6
    assert cal_mod.load_calibration(cfg, str(tmp_path))[0] is None


def test_metrics_module_under_reference_names():
    """utils.metrics drop-in (fast_confusion, IoU_from_confusions) against the reference's outputs (G8) and
    its argument checks; non-contiguous label values go through the lookup-table branch."""
    from conftest import load_golden
    m = _dropin("utils.metrics")
    g = load_golden("g8_metrics")
    conf = m.fast_confusion(g["true"], g["pred"], np.arange(20, dtype=np.int32))
    assert isinstance(conf, np.ndarray) and np.array_equal(conf, g["confusion"])
    assert np.allclose(m.IoU_from_confusions(conf), g["iou"], rtol=0, atol=1e-12)
    remap = np.array([3, 7, 11, 40, 41] + list(range(50, 65)), dtype=np.int64)           # 20 arbitrary label values
    conf2 = m.fast_confusion(remap[g["true"]], remap[g["pred"]], remap)
    assert np.array_equal(conf2, g["confusion"])
    assert m.fast_confusion(g["true"], g["pred"]).shape[0] == len(np.unique(np.hstack([g["true"], g["pred"]])))
    with pytest.raises(ValueError):
        m.fast_confusion(g["true"].astype(np.float32), g["pred"])
    with pytest.raises(ValueError):
        m.fast_confusion(g["true"], g["pred"], np.array([0, 0, 1]))
    with pytest.raises(ValueError):
        m.fast_confusion(np.zeros((3, 3), np.int32), np.zeros(9, np.int32))


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` without a launcher must start two ranks as a child process (never report
    n_gpus 1): launch rehearsal with MVK_BENCH_DRY=1 (gloo, no GPU work)."""
    import json
    env = dict(os.environ, MVK_BENCH_DRY="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["ranks"] == 2 and len(res["config"]["ms_per_step_per_rank"]) == 2
    # a world size that contradicts --gpus is refused loudly
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                        env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stderr + r2.stdout)


def test_bench_result_line_is_bounded():
    """The line of record must stay under 4 KB whatever the run looked like (BENCH_r02: a 22 KB line was not
    parsed by the driver): a worst-case result -- 8 ranks, long kernel labels, every optional object present --
    through bench.result_line, and a line that cannot fit is refused, not printed."""
    import importlib
    import json
    bench = importlib.import_module("bench")
    prof = {}
    for i in range(120):            # 120 distinct launch shapes, as the random grid orientation produces
        prof[("k", i)] = {"kernel": "kpconv_gather_vec<NCH=1>(LPP=16,PPW=4,+2 trailing channels)", "launches": 16,
                          "total_ms": 1.0 + (i == 0), "each_ms": [0.0671234567] * 16, "bytes_per_launch": 309427088.123,
                          "shape": {"Nq": 19464 - i, "Ns": 19464, "H": 58 - i % 5, "H_eff": 42.590731607069465, "Cin": 66, "K": 15}}
    bench.pmc_traffic = lambda best: (205634042.0, "r02_pmc_gather.json (python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline)")
    res = {"metric": "input points/s through MV-KPConv KPFCNN forward+backward (pyramid + fusion + fwd + bwd + SGD)",
           "value": 4236201.534778994, "unit": "points/s", "n_gpus": 8, "steps": 20, "warmup": 5,
           "ms_per_step": 4.5946822501719, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "middle_fusion_kpfcnn5_sphere19k_x1_per_gpu_deformable", "points_per_step_per_gpu": 19464,
                      "views": 5, "image_hw": [120, 160], "parallelism": "dp8",
                      "execution": "hipGraph[net|chain|enc2d]+eager-rccl(3 graphs)", "ranks": 8, "backend": "rccl",
                      "ms_per_step_per_rank": [4.5946822501719] * 8, "final_loss": 2.951704263687134,
                      "capacity_overflow": False},
           "roofline": bench.roofline(prof),
           "contraction": bench.mfma_report({(19464, 990, 64): {"launches": 16, "total_ms": 0.7, "flops_per_launch": 2.0 * 19464 * 990 * 64}}),
           "cpu_baseline": {"value": 15689.054546085035, "unit": "points/s", "cores": 16, "kind": "port",
                            "sample": "x" * 260, "value_pyramid_on_workers": 17000.123456}}
    line = bench.result_line(res)
    assert len(line) < 4096 and "\n" not in line
    d = json.loads(line)
    assert d["roofline"]["bound"] == "hbm" and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-5
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert len(bench.gather_by_level(prof)) == 5          # aggregated per level class, not per launch shape
    res["config"]["note"] = "y" * 5000
    with pytest.raises(SystemExit):
        bench.result_line(res)


# ---------------------------------------------------------------- drop-in boundary (INTEGRATION.md route 1)

_STUB_TREE = {
    "datasets/__init__.py": "",
    "datasets/ScanNet_sphere_color.py": (
        "from datasets.common import PointCloudDataset, grid_subsampling\n"
        "from utils.mayavi_visu import *\n"
        "from utils.config import bcolors, Config\n"
        "class StubDataset(PointCloudDataset):\n"
        "    def __init__(self):\n"
        "        PointCloudDataset.__init__(self, 'stub')\n"
        "MARK = 'stub ScanNet_sphere_color'\n"),
    "datasets/common.py": "raise ImportError('the reference datasets.common must be shadowed by the drop-in')\n",
    "utils/__init__.py": "",
    "utils/config.py": ("class bcolors:\n    OKBLUE = 'b'\n    ENDC = 'e'\n"
                        "class Config:\n    architecture = []\n    stub_field = 17\n    def save(self):\n        return 'saved'\n"),
    "utils/mayavi_visu.py": "def show_ModelNet_models(*a):\n    return 'visu'\n",
    "utils/trainer.py": ("from utils.ply import read_ply, write_ply\nfrom utils.metrics import IoU_from_confusions, fast_confusion\n"
                         "from utils.config import Config\nfrom mvpnet.utils.visualize import *\n"
                         "class ModelTrainer:\n    pass\n"),
    "models/__init__.py": "",
    "models/blocks.py": "raise ImportError('the reference models.blocks must be shadowed by the drop-in')\n",
    "kernels/__init__.py": "",
    "cpp_wrappers/__init__.py": "",
}
_STUB_ROOT = {
    "mvpnet/__init__.py": "",
    "mvpnet/utils/__init__.py": "",
    "mvpnet/utils/visualize.py": "def visualize_labels(*a):\n    return 'labels'\n",
    "mvpnet/models/__init__.py": "",
    "mvpnet/models/mvpnet_2d.py": "MARK = 'stub mvpnet_2d'\n",
    "common/__init__.py": "",
    "common/utils/__init__.py": "",
    "common/utils/checkpoint.py": "MARK = 'stub checkpoint'\n",
}


def test_dropin_first_on_sys_path_still_reaches_the_reference_modules(tmp_path):
    """INTEGRATION.md route 1 with a stub 'reference' tree (own text) behind dropin/ on sys.path: the
    modules the drop-in mirrors win, every other submodule of the same packages -- datasets.ScanNet_sphere_color,
    utils.trainer, utils.mayavi_visu, mvpnet.utils.visualize, common.utils.* -- still imports, and
    utils.config stays the reference's own (its fields / save())."""
    kp, root = tmp_path / "KPConv-PyTorch", tmp_path
    for base, tree in ((kp, _STUB_TREE), (root, _STUB_ROOT)):
        for rel, text in tree.items():
            f = base / rel
            f.parent.mkdir(parents=True, exist_ok=True)
            f.write_text(text)
    prog = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.append(%r)\n"                 # what a user adds: dropin first, repo root for mvpnet/common
        "import datasets.ScanNet_sphere_color as S, utils.trainer as T, utils.config as Cf\n"
        "import datasets.common as DC, models.blocks as MB, mvpnet.utils.visualize as V, common.utils.checkpoint as CK\n"
        "import mvpnet.models.mvpnet_2d as M2, mvpnet.models.mvpnet_3d as M3, mvpnet.ops.group_points as GP\n"
        "from common.nn import SharedMLP\n"
        "import cpp_wrappers.cpp_subsampling.grid_subsampling as G, cpp_wrappers.cpp_neighbors.radius_neighbors as R\n"
        "assert S.MARK.startswith('stub') and CK.MARK.startswith('stub') and M2.MARK.startswith('stub')\n"
        "assert 'dropin' in DC.__file__ and 'dropin' in MB.__file__ and 'dropin' in M3.__file__ and 'dropin' in G.__file__\n"
        "assert hasattr(MB, 'KPConv') and hasattr(M3, 'FeatureAggregation') and hasattr(DC, 'ScanNetCustomBatch')\n"
        "assert Cf.Config.stub_field == 17 and Cf.Config().save() == 'saved' and Cf.bcolors.OKBLUE == 'b'\n"
        "d = S.StubDataset(); assert d.name == 'stub' and d.config.stub_field == 17\n"
        "assert T.visualize_labels() == 'labels' and S.show_ModelNet_models() == 'visu'\n"
        "print('FALLTHROUGH OK')\n") % (os.path.join(ROOT, PKG, "dropin"), str(root))
    r = subprocess.run([sys.executable, "-c", prog], cwd=str(kp), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert r.returncode == 0 and "FALLTHROUGH OK" in r.stdout, r.stderr[-3000:]


_WORKER_DATASET = (
    "import os, sys, importlib, numpy as np, torch\n"
    "from torch.utils.data import Dataset\n"
    "class Stub(Dataset):\n"
    "    # stands for the reference dataset: __getitem__ (potential_item) calls the pyramid builders, i.e. the library\n"
    "    def __init__(self, lib_module):\n"
    "        self.lib_module = lib_module\n"
    "    def __len__(self):\n"
    "        return 8\n"
    "    def __getitem__(self, i):\n"
    "        lib_mod = importlib.import_module(self.lib_module)\n"
    "        lib_mod._check_process()          # raises FORK_MESSAGE in a forked child of the library's owner\n"
    "        return np.array([i, os.getpid(), os.getppid()], np.int64)\n")


def test_unmodified_script_dataloader_workers_are_not_forked(tmp_path):
    """n2: with dropin/ first on sys.path, `import datasets...` makes `spawn` the default start method, so a
    DataLoader created exactly as the reference script does (train_ScanNet_sphere.py:365-377: num_workers > 0, no
    multiprocessing_context) hands its items to fresh processes in which the library call is legal -- while the same
    loader under `fork` (MVK_DATALOADER_START=keep) gets the loud fork error, not a hang."""
    (tmp_path / "stubds.py").write_text(_WORKER_DATASET)
    kp = tmp_path / "KPConv-PyTorch"
    for rel, text in _STUB_TREE.items():
        f = kp / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(text)
    script = tmp_path / "train_stub.py"
    script.write_text(
        "import os, sys\n"
        "sys.path.insert(0, %r); sys.path.append(%r); sys.path.append(%r)\n"
        "import multiprocessing as mp\n"
        "import torch\n"
        "from torch.utils.data import DataLoader\n"
        "import datasets.common as DC                      # the drop-in (route 1)\n"
        "import stubds\n"
        "if __name__ == '__main__':\n"
        "    lib_mod = __import__(%r, fromlist=['_lib'])._lib\n"
        "    lib_mod._owner_pid = os.getpid()               # the parent has 'initialised the library' (no GPU here)\n"
        "    loader = DataLoader(stubds.Stub(lib_mod.__name__), batch_size=1, num_workers=2)\n"
        "    try:\n"
        "        rows = [b[0].tolist() for b in loader]\n"
        "        assert sorted(r[0] for r in rows) == list(range(8))\n"
        "        assert all(r[1] != os.getpid() for r in rows)\n"
        "        print('WORKERS OK', mp.get_start_method())\n"
        "    except RuntimeError as e:\n"
        "        print('WORKERS FAILED', mp.get_start_method(), 'forked child' in str(e))\n"
        % (os.path.join(ROOT, PKG, "dropin"), str(tmp_path), ROOT, PKG))
    env = dict(os.environ)
    env.pop("MVK_DATALOADER_START", None)
    r = subprocess.run([sys.executable, str(script)], cwd=str(kp), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0 and "WORKERS OK spawn" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    r = subprocess.run([sys.executable, str(script)], cwd=str(kp), env=dict(env, MVK_DATALOADER_START="keep"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert "WORKERS FAILED fork True" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_scannet_custom_batch_round_trips_the_flat_list():
    """ScanNetCustomBatch(input_list): L = (len - 11) // 5 (ScanNet_sphere_color.py:1535), dtypes kept,
    pin_memory() / to() return self; the baseline variant with L = (len - 7) // 5."""
    common = importlib.import_module(PKG + ".dropin.datasets.common")
    rng = np.random.default_rng(0)
    L, n = 3, [50, 20, 8]
    pts = [rng.random((m, 3)).astype(np.float32) for m in n]
    nb = [np.concatenate([rng.integers(0, m - 5, (m - 5, 6)), rng.integers(m - 5, m + 1, (5, 6))]).astype(np.int64) for m in n]
    pools = [rng.integers(0, n[i] + 1, (n[i + 1], 5)).astype(np.int64) for i in range(L - 1)] + [np.zeros((0, 1), np.int64)]
    ups = [rng.integers(0, n[i + 1] + 1, (n[i], 4)).astype(np.int64) for i in range(L - 1)] + [np.zeros((0, 1), np.int64)]
    lens = [np.array([m - 5, 5], np.int32) for m in n]
    tail = [rng.random((1, 50, 3)).astype(np.float32), rng.random((2, 3, 4, 5, 3)).astype(np.float32),
            rng.random((2, 3, 3, 4, 5)).astype(np.float32), rng.integers(0, 20, 50).astype(np.int64),
            np.ones((2, 3), np.float32), np.tile(np.eye(3, dtype=np.float32), (2, 1, 1)), np.array([0, 1], np.int32),
            np.array([3, 4], np.int32), np.arange(50, dtype=np.int32), [np.zeros((45, 3), np.int64), np.zeros((5, 3), np.int64)],
            rng.random((50, 4)).astype(np.float32)]
    flat = pts + nb + pools + ups + lens + tail
    assert len(flat) == 5 * L + 11
    b = common.ScanNetCustomBatch([flat])
    assert len(b.points) == len(b.neighbors) == len(b.pools) == len(b.upsamples) == len(b.lengths) == L
    assert b.points[1].dtype == torch.float32 and b.neighbors[0].dtype == torch.int64 and b.lengths[0].dtype == torch.int32
    assert b.labels.dtype == torch.int64 and torch.equal(b.feature_3d, torch.from_numpy(tail[-1]))
    assert b.knn_list is tail[9] and np.array_equal(b.images.numpy(), tail[2])
    assert b.pin_memory() is b and b.to("cpu") is b
    back = b.points + b.neighbors + b.pools + b.upsamples + b.lengths
    assert all(np.array_equal(t.numpy(), a) for t, a in zip(back, flat[:5 * L]))
    un = b.unstack_neighbors(0)
    assert len(un) == 2 and un[0].shape == (45, 6) and un[1].max() < 5 and un[1].min() >= -1
    with pytest.raises(ValueError):
        common.ScanNetCustomBatch([flat[:-1]])
    base = common.ScanNetBaselineCustomBatch([flat[:5 * L] + [tail[-1], tail[3], tail[4], tail[5], tail[6], tail[7], tail[8]]])
    assert len(base.points) == L and base.features.shape == (50, 4) and base.input_inds.dtype == torch.int32


def test_point_cloud_dataset_augmentation_matches_the_reference_draw_order():
    """PointCloudDataset.augmentation_transform(_new): same RNG consumption and arithmetic as the cited
    lines (common.py:252-409), checked against a literal replay with the same seed."""
    common = importlib.import_module(PKG + ".dropin.datasets.common")
    ds = common.PointCloudDataset("x")
    ds.config.augment_rotation, ds.config.augment_scale_anisotropic = 'vertical', True
    ds.config.augment_symmetries, ds.config.augment_noise = [True, False, False], 0.001
    ds.config.augment_scale_min, ds.config.augment_scale_max = 0.8, 1.2
    pts = np.random.default_rng(1).random((40, 3)).astype(np.float32)
    xyz = np.random.default_rng(2).random((2, 3, 4, 3)).astype(np.float32)
    np.random.seed(7)
    a, scale, R, axyz = ds.augmentation_transform_new(pts, xyz)
    np.random.seed(7)
    theta = np.random.rand() * 2 * np.pi
    c, s = np.cos(theta), np.sin(theta)
    R0 = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    sc = np.random.rand(3) * 0.4 + 0.8
    sym = np.array([1, 0, 0], np.int32) * np.random.randint(2, size=3)
    sc = (sc * (1 - sym * 2)).astype(np.float32)
    noise = (np.random.randn(40, 3) * 0.001).astype(np.float32)
    assert np.array_equal(R, R0) and np.array_equal(scale, sc)
    assert np.array_equal(a, np.sum(np.expand_dims(pts, 2) * R0, axis=1) * sc + noise)
    assert np.array_equal(axyz, (np.sum(np.expand_dims(xyz.reshape(-1, 3), 2) * R0, axis=1) * sc).reshape(xyz.shape))
    assert ds.big_neighborhood_filter(np.zeros((4, 9)), 0).shape == (4, 9)
    ds.neighborhood_limits = [5]
    assert ds.big_neighborhood_filter(np.zeros((4, 9)), 0).shape == (4, 5)


def _forked_child_calls_lib(q):
    lib_mod = importlib.import_module(PKG + "._lib")
    try:
        lib_mod.lib()
        q.put("no error")
    except RuntimeError as e:
        q.put(str(e))


def test_host_logic_of_the_gather_work_lists_and_the_features_computed_ahead():
    """What does not need a GPU of round 3's host logic: the side channel that carries a level's work list from
    segmentation_inputs_sphere to the blocks is bounded and keyed by (device, address, rows); a block prefers the batch's
    own `orders`; lift_2d_features hands back features computed ahead only where the network declares that it detaches
    them; the rigid loss does not add a python zero."""
    import importlib
    import types
    ops = importlib.import_module(PKG + ".ops")
    blocks = importlib.import_module(PKG + ".dropin.models.blocks")
    fc = importlib.import_module(PKG + ".dropin.models.fusion_common")
    ops._WORK_ORDERS.clear()
    pts = [torch.zeros(5 + i, 3) for i in range(ops._WORK_ORDERS_KEPT + 6)]
    for p in pts:
        ops.remember_work_order(p, torch.arange(p.shape[0], dtype=torch.int32))
    assert len(ops._WORK_ORDERS) == ops._WORK_ORDERS_KEPT
    assert (None, pts[0].data_ptr(), 5) not in ops._WORK_ORDERS and (None, pts[-1].data_ptr(), pts[-1].shape[0]) in ops._WORK_ORDERS
    assert ops.work_order_for(pts[-1]) is None                       # CPU tensors never reach a kernel
    batch = types.SimpleNamespace(points=pts[:3], orders=["o0", "o1", None])
    assert blocks._work_order("resnetb", 0, batch) == "o0" and blocks._work_order("resnetb_strided", 0, batch) == "o1"
    assert blocks._work_order("resnetb", 2, batch) is None and blocks._work_order("resnetb_strided", 2, batch) is None
    ahead = torch.ones(4, 64)
    net = types.SimpleNamespace(fa_output_detached=True)
    assert fc.lift_2d_features(net, types.SimpleNamespace(feature_2d3d=ahead)) is ahead
    with pytest.raises(AttributeError):                               # a network that trains through the module recomputes
        fc.lift_2d_features(types.SimpleNamespace(fa_output_detached=False), types.SimpleNamespace(feature_2d3d=ahead))
    ops._WORK_ORDERS.clear()


def test_forked_child_fails_loudly():
    """A forked child of the process that loaded the HIP library must get a RuntimeError with the
    INTEGRATION note (spawn / main-process pyramid), never a hang inside HIP."""
    import multiprocessing as mp
    lib_mod = importlib.import_module(PKG + "._lib")
    lib_mod.lib()
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_forked_child_calls_lib, args=(q,))
    p.start()
    msg = q.get(timeout=60)
    p.join(60)
    assert "forked child" in msg and "spawn" in msg and "INTEGRATION.md" in msg


def _two_stage_worker(rank, world, port, q):
    """A small torch-only network with the cut protocol of run_encoder_decoder: two-stage backward + bucketed
    asynchronous all-reduce must give the gradients of plain backward + one all-reduce."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module(PKG + ".dp")
    torch.manual_seed(0)
    early = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh())
    late = torch.nn.Sequential(torch.nn.Linear(8 + 6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    x0 = torch.randn(7, 6, generator=torch.Generator().manual_seed(10 + rank))

    def forward(sever):
        skip = x0 * 1.0                                   # a tensor used on both sides of the cut
        skip.requires_grad_(False)
        s = skip + early[0].bias[:6].sum() * 0           # depends on an early parameter (gradient zero) -> requires grad
        h = early(s)
        if sever:
            orig = [h, s]
            leaves = [t.detach().requires_grad_() for t in orig]
            h, s_use = leaves
        else:
            orig = leaves = None
            s_use = s
        return late(torch.cat([h, s_use], 1)).pow(2).sum(), (orig, leaves)

    # reference: plain backward + all-reduce of everything
    loss, _ = forward(False)
    loss.backward()
    params = list(early.parameters()) + list(late.parameters())
    red = dp.FlatAllReduce(params, world)
    red()
    want = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    # two stages, two buckets, the late bucket in flight during stage 2
    loss, cut = forward(True)
    br = dp.BucketedAllReduce([list(late.parameters()), list(early.parameters())], world)

    def between():
        assert all(p.grad is not None for p in late.parameters()) and all(p.grad is None for p in early.parameters())
        br.pack(0)
        br.launch(0)
    dp.two_stage_backward(loss, cut, between=between)
    br.pack(1)
    br.launch(1)
    br.wait()
    br.unpack(0)
    br.unpack(1)
    ok = all(torch.allclose(p.grad, w, rtol=1e-6, atol=1e-7) for p, w in zip(params, want))
    # the same with the backward seeded by 1 / world: no division, .grad re-pointed at the bucket (no copy back)
    for p in params:
        p.grad = None
    loss, cut = forward(True)
    bp = dp.BucketedAllReduce([list(late.parameters()), list(early.parameters())], world, prescaled=True)

    def between_p():
        bp.pack(0)
        bp.launch(0)
    dp.two_stage_backward(loss, cut, between=between_p, seed=bp.seed(loss))
    bp.pack(1)
    bp.launch(1)
    bp.wait()
    bp.unpack(0)
    bp.unpack(1)
    ok = ok and all(torch.allclose(p.grad, w, rtol=1e-6, atol=1e-7) for p, w in zip(params, want))
    ok = ok and all(p.grad.untyped_storage().data_ptr() in (bp.flat[0].untyped_storage().data_ptr(),
                                                            bp.flat[1].untyped_storage().data_ptr()) for p in params)
    q.put((rank, ok, dp.cut_block_of_layer(['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb',
                                            'nearest_upsample', 'unary'], 2)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_stage_backward_with_bucketed_allreduce_two_gloo_ranks():
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_two_stage_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res) and res[0][2] == 5
