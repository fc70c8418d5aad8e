"""Deterministic synthetic ScanNet-shaped inputs: a box-world "room" sampled
as a point cloud, cropped to a sphere, plus pinhole RGB-D views of the same room rendered by
ray / box intersection. No dataset or checkpoint is needed (the GPU box has no network).

Host-side NumPy only generates RAW inputs (points, colours, labels, depth maps, poses); everything
on the hot path (subsampling, neighbours, unprojection, k-NN, network) then runs through the HIP
library (``build_batch``).
"""
import os

import numpy as np
import torch

from . import ops
from .dropin.datasets.common import pyramid_plan, random_grid_rotations, segmentation_inputs_sphere, SphereBatch
from .dropin.utils.config import Config

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)


# ------------------------------------------------------------------------------------ scene

def room_boxes(rng):
    """Axis-aligned boxes (lo, hi): floor, two walls, a table, clutter. Room frame: z up, floor z = 0."""
    boxes = [
        ([-3.0, -3.0, -0.05], [3.0, 3.0, 0.0]),        # floor
        ([-0.85, -3.0, 0.0], [-0.8, 3.0, 2.6]),        # wall x = -0.8
        ([-3.0, 0.9, 0.0], [3.0, 0.95, 2.6]),          # wall y = 0.9
        ([1.25, -3.0, 0.0], [1.3, 3.0, 2.6]),          # wall x = 1.25
        ([-3.0, -3.0, 1.9], [3.0, 3.0, 1.95]),         # low ceiling
        ([-0.5, -0.6, 0.72], [0.6, 0.3, 0.75]),        # table slab
    ]
    for lx, ly in ((-0.45, -0.55), (0.5, -0.55), (-0.45, 0.2), (0.5, 0.2)):   # table legs
        boxes.append(([lx, ly, 0.0], [lx + 0.05, ly + 0.05, 0.72]))
    for _ in range(30):                                                        # clutter
        c = np.array([rng.uniform(-0.7, 1.2), rng.uniform(-1.5, 0.8), 0.0])
        s = rng.uniform(0.15, 0.55, 3)
        z0 = rng.choice([0.0, 0.75]) if abs(c[0]) < 0.5 and -0.6 < c[1] < 0.3 else 0.0
        boxes.append(((c - [s[0] / 2, s[1] / 2, -z0]).tolist(), (c + [s[0] / 2, s[1] / 2, z0 + s[2]]).tolist()))
    return [(np.asarray(a, np.float64), np.asarray(b, np.float64)) for a, b in boxes]


def sample_box_surfaces(rng, boxes, density):
    """Uniform samples on all faces of all boxes (points per m^2 = density) + 5 mm Gaussian noise."""
    out = []
    for lo, hi in boxes:
        d = hi - lo
        for ax in range(3):
            a, b = (ax + 1) % 3, (ax + 2) % 3
            n = int(d[a] * d[b] * density) + 1
            for face in (lo[ax], hi[ax]):
                p = np.empty((n, 3))
                p[:, ax] = face
                p[:, a] = rng.uniform(lo[a], hi[a], n)
                p[:, b] = rng.uniform(lo[b], hi[b], n)
                out.append(p)
    p = np.concatenate(out, 0)
    return p + rng.normal(0, 0.005, p.shape)


def raw_sphere(seed=0, radius=1.2, density=6000.0, center=(0.2, -0.2, 0.8)):
    """Raw (un-subsampled) sphere in room (world) coordinates, colours U[0,1], labels U{0..19}."""
    rng = np.random.default_rng(seed)
    boxes = room_boxes(rng)
    p = sample_box_surfaces(rng, boxes, density)
    c = np.asarray(center)
    keep = np.sum((p - c) ** 2, axis=1) < radius ** 2
    p = p[keep]
    p = p[rng.permutation(p.shape[0])]
    colors = rng.random((p.shape[0], 3)).astype(np.float32)
    labels = rng.integers(0, 20, p.shape[0]).astype(np.int32)
    return dict(points=p.astype(np.float32), colors=colors, labels=labels, center=c, boxes=boxes, rng=rng)


# ------------------------------------------------------------------------------------ cameras

def look_at(eye, target, up=(0, 0, 1)):
    """camera-to-world pose (4x4 float32), camera looks along +z, x right, y down (ScanNet convention)."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, eye
    return T.astype(np.float32)


def render_depth(boxes, pose, cam, h, w):
    """Depth (uint16 millimetres, 0 = no hit) by ray / axis-aligned-box intersection (slab method)."""
    v, u = np.indices((h, w))
    d_cam = np.stack([(u - cam[0, 2]) / cam[0, 0], (v - cam[1, 2]) / cam[1, 1], np.ones_like(u, float)], -1)
    R, o = pose[:3, :3].astype(np.float64), pose[:3, 3].astype(np.float64)
    d = d_cam @ R.T                                   # world ray directions, |d_z_cam| = 1 -> t == depth
    best = np.full((h, w), np.inf)
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = 1.0 / d
        for lo, hi in boxes:
            t0 = (lo - o) * inv
            t1 = (hi - o) * inv
            tn = np.minimum(t0, t1).max(-1)
            tf = np.maximum(t0, t1).min(-1)
            hit = (tf >= np.maximum(tn, 1e-3))
            t = np.where(tn > 1e-3, tn, tf)
            best = np.where(hit & (t < best), t, best)
    depth = np.where(np.isfinite(best) & (best < 60.0), best, 0.0)
    return np.round(depth * 1000.0).astype(np.uint16)


def sphere_views(sphere, nv=3, h=120, w=160):
    """nv synthetic RGB-D frames looking at the sphere centre: images (nv,3,h,w) f32 normalised with
    the ImageNet statistics (train_ScanNet_sphere.py:340), depth (nv,h,w) u16 mm, poses (nv,4,4) f32,
    cam (3,3) f32 = ScanNet depth intrinsics / 4."""
    rng = sphere['rng']
    cam = np.array([[144.5, 0, 79.9], [0, 144.5, 59.9], [0, 0, 1]], np.float32)
    c = sphere['center']
    eyes = [c + [1.9 * np.cos(a), -1.9 * abs(np.sin(a)) - 0.3, 0.6 + 0.2 * i]
            for i, a in enumerate(np.linspace(0.3, 2.6, nv))]
    poses = np.stack([look_at(e, c) for e in eyes], 0)
    depth = np.stack([render_depth(sphere['boxes'], p, cam, h, w) for p in poses], 0)
    img = rng.random((nv, h, w, 3)).astype(np.float32)
    img = ((img - IMAGENET_MEAN) / IMAGENET_STD).transpose(0, 3, 1, 2).copy()
    return dict(images=img, depth=depth, poses=poses, cam=cam)


# ------------------------------------------------------------------------------------ configs

ARCH_RIGID = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb', 'resnetb_strided', 'resnetb',
              'resnetb', 'resnetb_strided', 'resnetb', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb',
              'nearest_upsample', 'unary', 'nearest_upsample', 'unary', 'nearest_upsample', 'unary',
              'nearest_upsample', 'unary']
# the deformable architecture the reference trains (train_ScanNet_sphere_middle_fusion.py:87-105,
# train_ScanNet_sphere_late_fusion.py:88-106): 10 encoder blocks, deformable from the third level on
ARCH_DEFORM = ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb_strided', 'resnetb_deformable',
               'resnetb_deformable_strided', 'resnetb_deformable', 'resnetb_deformable_strided', 'resnetb_deformable',
               'nearest_upsample', 'unary', 'nearest_upsample', 'unary', 'nearest_upsample', 'unary',
               'nearest_upsample', 'unary']


def make_config(variant="early", deformable=False, modulated=False):
    """The hyper-parameters of the reference's train scripts (train_ScanNet_sphere.py:39-264,
    train_ScanNet_baseline.py, ..._middle_fusion.py, ..._late_fusion.py) as a Config."""

    class _C(Config):
        dataset = 'ScanNet'
        dataset_task = 'cloud_segmentation'
        num_classes = 20
        architecture = ARCH_DEFORM if deformable else ARCH_RIGID
        in_radius = 1.2
        num_kernel_points = 15
        first_subsampling_dl = 0.04
        conv_radius = 2.5
        deform_radius = 6.0
        KP_extent = 1.2
        KP_influence = 'linear'
        aggregation_mode = 'sum'
        first_features_dim = 128
        use_batch_norm = True
        batch_norm_momentum = 0.02
        deform_fitting_mode = 'point2point'
        deform_fitting_power = 1.0
        deform_lr_factor = 0.1
        repulse_extent = 1.2
        learning_rate = 1e-2
        momentum = 0.98
        weight_decay = 1e-3
        grad_clip_norm = 100.0
        batch_num = 5

    c = _C()
    c.modulated = modulated
    c.variant = variant
    if variant == "baseline":
        c.in_features_dim = 2                      # 1 + z (train_ScanNet_baseline.py)
    elif variant == "early":
        c.early_fusion = True
        c.in_features_dim = 66                     # 1 + z + 64 (train_ScanNet_sphere.py:196)
    elif variant == "middle":
        c.middle_fusion = True
        c.in_features_dim_3d = 4                   # 1 + rgb
        c.in_features_dim_2d = 65                  # 1 + 64
        c.in_features_dim = 4
    elif variant == "late":
        c.late_fusion = True
        c.in_features_dim = 4
    else:
        raise ValueError(variant)
    return c


def build_model(config, device):
    from .dropin.models import architectures, architectures_sphere, architectures_sphere_middle_fusion, \
        architectures_sphere_late_fusion
    lbl = list(range(20))
    cls = {"baseline": architectures.KPFCNN, "early": architectures_sphere.KPFCNN_featureAggre,
           "middle": architectures_sphere_middle_fusion.KPFCNN_featureAggre,
           "late": architectures_sphere_late_fusion.KPFCNN_featureAggre}[config.variant]
    return cls(config, lbl, []).to(device)


# ------------------------------------------------------------------------------------ batches

def stage_spheres(spheres, device, views=None):
    """Raw host inputs -> HBM once, then the scene-load subsampling at first_subsampling_dl
    (load_subsampled_clouds' grid_subsampling with colours + labels, ScanNet_sphere_color.py:937-940)
    -- like the reference this happens once per cloud, not once per step."""
    st = dict(points=[], colors=[], labels=[], center=[])
    for s in spheres:
        p = torch.from_numpy(s['points']).to(device)
        c = torch.from_numpy(s['colors']).to(device)
        l = torch.from_numpy(s['labels']).to(device)
        sp, _, sc, sl = ops.grid_subsample_batch(p, [p.shape[0]], features=c, labels=l, dl=0.04 if 'dl' not in s else s['dl'])
        st['points'].append(sp), st['colors'].append(sc), st['labels'].append(sl[:, 0].long())
        st['center'].append(torch.tensor(s['center'], dtype=torch.float32, device=device))
    if views is not None:
        st['images'] = [torch.from_numpy(v['images']).to(device) for v in views]
        st['depth'] = [torch.from_numpy(v['depth'].astype(np.int16)).to(device) for v in views]
        st['poses'] = [torch.from_numpy(v['poses']).to(device) for v in views]
        st['cam'] = [v['cam'] for v in views]
    return st


def build_batch(config, staged, limits=None, index_dtype=torch.int32, rotations=None, status=None):
    """One pass of the input side of the hot path, all on the GPU, from the subsampled sphere clouds:
    centring + stacking (potential_item, ScanNet_sphere_color.py:600-719), pyramid
    (segmentation_inputs_sphere, datasets/common.py:779-900) and, for the fusion variants, depth
    unprojection + 3-NN pixel indices (get_rgbd_data :352-474)."""
    world = staged['points']
    pts = [p - c for p, c in zip(world, staged['center'])]      # input_points = points - center_point (:600)
    lens = [int(p.shape[0]) for p in pts]
    stacked = torch.cat(pts, 0)
    stacked_world = torch.cat(world, 0)
    pyr = segmentation_inputs_sphere(config, stacked, np.asarray(lens, np.int32), limits, index_dtype, rotations,
                                     status=status)
    ones = torch.ones_like(stacked[:, :1])
    labels = torch.cat(staged['labels'], 0)
    colors = torch.cat(staged['colors'], 0)
    z = stacked_world[:, 2:3]                       # height feature = world z (:634)
    v = config.variant
    if v == "baseline":
        return SphereBatch(pyr, labels, features=torch.cat([ones, z], 1)), lens
    feat3d = torch.cat([ones, z], 1) if v == "early" else torch.cat([ones, colors], 1)
    image_xyz, knn = [], []
    for i, pw in enumerate(world):
        xyz, valid = ops.unproject_depth(staged['depth'][i], staged['cam'][i], staged['poses'][i])
        knn.append(ops.knn_pixels(pw, xyz, valid, k=3).unsqueeze(0))          # get_rgbd_data :448-451
        image_xyz.append(xyz.to(torch.float32))                                # :454
    batch = SphereBatch(pyr, labels, feature_3d=feat3d, feat_aggre_points=stacked_world.unsqueeze(0),
                        image_xyz=torch.stack(image_xyz, 0), images=torch.stack(staged['images'], 0),
                        knn_list=knn)
    # pixel indices into the stacked (b*nv) views for the one-gather FeatureAggregation (fusion_common.lift_2d_features)
    per_sphere = int(np.prod(staged['depth'][0].shape))
    batch.knn_stacked = torch.cat([k[0] + i * per_sphere for i, k in enumerate(knn)], 0) if len(knn) > 1 else knn[0][0]
    return batch, lens


def calibrate_limits(config, staged, keep=0.9):
    """neighborhood_limits like the reference's calibration (ScanNet_sphere_color.py:1380-1464):
    per layer, the neighbour count below which `keep` of the conv neighbourhoods fall."""
    pts = [p - c for p, c in zip(staged['points'], staged['center'])]
    lens = [int(p.shape[0]) for p in pts]
    pyr = segmentation_inputs_sphere(config, torch.cat(pts, 0), np.asarray(lens, np.int32), None, torch.int32)
    limits = []
    for layer, nb in enumerate(pyr['neighbors']):
        ns = pyr['points'][layer].shape[0]
        if nb.shape[0] == 0:
            limits.append(1)
            continue
        counts = (nb < ns).sum(1).cpu().numpy()
        hist = np.bincount(counts, minlength=nb.shape[1] + 1)
        cum = np.cumsum(hist)
        limits.append(int(np.sum(cum < keep * cum[-1])))
    return limits


# ------------------------------------------------------------------------------------ capacity-padded batches

class StaticBatch:
    """Capacity-padded, fixed-address copy of a SphereBatch for hipGraph replay.

    Levels 1.. are padded to a fixed row capacity (their point counts change from step to step with
    the random grid orientation); padded points sit at 1e6, padded index rows are all-shadow, and
    `valid[capacity]` holds the real row count on the DEVICE for the masked BatchNorm kernels. Level 0
    keeps its exact size (the spheres of a batch fix it). Neighbour matrices get their full calibrated
    width (`limits`), shadow index = capacity of the support level."""

    def __init__(self, batch, limits, margin=1.12, caps=None):
        L = len(batch.points)
        dev = batch.points[0].device
        self.n0 = batch.points[0].shape[0]
        given, caps, used = caps, [], set()
        for l in range(L if given is None else 0):
            c = self.n0 if l == 0 else int(-(-int(batch.points[l].shape[0] * margin + 8) // 64) * 64)
            while c in used:
                c += 64
            used.add(c)
            caps.append(c)
        if given is not None:            # a second static set with the capacities of the first
            caps = list(given)
        self.caps, self.limits = caps, [int(x) for x in limits]
        it = batch.neighbors[0].dtype
        self.points = [torch.full((caps[l], 3), 1e6, device=dev) for l in range(L)]
        self.neighbors = [torch.full((caps[l], self.limits[l]), caps[l], dtype=it, device=dev) for l in range(L)]
        self.pools = [torch.full((caps[l + 1], self.limits[l]), caps[l], dtype=it, device=dev) if l + 1 < L
                      else batch.pools[l] for l in range(L)]
        self.upsamples = [torch.full((caps[l], self.limits[l + 1]), caps[l + 1], dtype=it, device=dev) if l + 1 < L
                          else batch.upsamples[l] for l in range(L)]
        self.lengths = batch.lengths
        # gather work lists (datasets/common.py `orders`): a permutation of ALL capacity rows, identity over the padding
        self._iota = [torch.arange(caps[l], dtype=torch.int32, device=dev) for l in range(L)]
        has = getattr(batch, 'orders', None)
        self.orders = [self._iota[l].clone() if (has and has[l] is not None) else None for l in range(L)] if has else None
        # transposed neighbour matrices of the rigid layers (datasets/common.py `rev_neighbors` / `rev_pools`): capacity
        # rows, width = the first batch's longest row + a third (rows longer than that raise the chain's overflow word)
        def rev_static(name, shadow_of):
            src = getattr(batch, name, None)
            if not src:
                return None
            mats = []
            for l in range(L):
                if l >= len(src) or src[l] is None:
                    mats.append(None)
                    continue
                w = min(ops.reverse_width_cap(), -(-int(src[l].shape[1] * 1.35 + 8) // 8) * 8)
                mats.append(torch.full((caps[l], w), caps[shadow_of(l)], dtype=torch.int32, device=dev))
            return mats
        self.rev_neighbors = rev_static('rev_neighbors', lambda l: l)
        self.rev_pools = rev_static('rev_pools', lambda l: min(l + 1, L - 1))
        # the blocks' max_pool / closest_pool find their reverse lists by the index matrix itself (ops.remember_reverse)
        for l in range(L):
            if self.rev_pools and self.rev_pools[l] is not None:
                ops.remember_reverse(self.pools[l], self.rev_pools[l])
        self.rev_ups = {}
        for l, src in (getattr(batch, 'rev_ups', None) or {}).items():        # deterministic mode only
            self.rev_ups[l] = torch.full((caps[l + 1], 64), caps[l], dtype=torch.int32, device=dev)
            ops.remember_reverse(self.upsamples[l], self.rev_ups[l], first_column=True)
        self.valid = {caps[l]: torch.zeros(1, dtype=torch.int32, device=dev) for l in range(L)}
        self._counts = [self.valid[caps[l]] for l in range(L)]
        for name in self._DENSE:
            v = getattr(batch, name, None)
            setattr(self, name, v.clone() if v is not None else None)
        self.knn_list = [k.clone() for k in batch.knn_list] if batch.knn_list is not None else None
        self.load(batch)

    # feature_2d: output of the frozen 2D encoder when it was run ahead of the network step
    # feature_2d3d: FeatureAggregation's output when it, too, was computed ahead (networks that detach it)
    # stacked_features: early fusion's network input [feature_3d | feature_2d3d], built with the latter
    _DENSE = ('labels', 'features', 'feature_3d', 'feat_aggre_points', 'image_xyz', 'images', 'feature_2d', 'feature_2d3d',
              'stacked_features', 'knn_stacked')

    def load(self, batch):
        """Copies one freshly built batch into the static buffers (raises if a level outgrew its capacity)."""
        L = len(batch.points)
        n = [int(p.shape[0]) for p in batch.points]
        if n[0] != self.n0 or any(n[l] > self.caps[l] for l in range(L)):
            raise RuntimeError("batch does not fit the captured capacities %s: %s" % (self.caps, n))
        for l in range(L):
            ops.pad_points(batch.points[l], self.points[l], 1e6, self._counts[l])
            if self.orders and self.orders[l] is not None:
                src = batch.orders[l] if getattr(batch, 'orders', None) else None
                self.orders[l].copy_(self._iota[l])
                if src is not None:
                    self.orders[l][:n[l]].copy_(src)
            self._put(self.neighbors[l], batch.neighbors[l], n[l], self.caps[l])
            if l + 1 < L:
                self._put(self.pools[l], batch.pools[l], n[l], self.caps[l])
                self._put(self.upsamples[l], batch.upsamples[l], n[l + 1], self.caps[l + 1])
            for dst, src, ql in ((self.rev_neighbors, getattr(batch, 'rev_neighbors', None), l),
                                 (self.rev_pools, getattr(batch, 'rev_pools', None), min(l + 1, L - 1))):
                if dst and dst[l] is not None:
                    if not src or src[l] is None:
                        raise RuntimeError("batch carries no reverse neighbour list for layer %d" % l)
                    self._put(dst[l], src[l], n[ql], self.caps[ql])
            if l in self.rev_ups:
                src = (getattr(batch, 'rev_ups', None) or {}).get(l)
                if src is None:
                    raise RuntimeError("batch carries no reverse upsampling list for layer %d" % l)
                self._put(self.rev_ups[l], src, n[l], self.caps[l])
        for name in self._DENSE:
            v = getattr(batch, name, None)
            if v is not None:
                getattr(self, name).copy_(v)
        if self.knn_list is not None:
            for dst, src in zip(self.knn_list, batch.knn_list):
                dst.copy_(src)

    @staticmethod
    def _put(dst, src, shadow_src, shadow_dst):
        if src.shape[1] > dst.shape[1]:
            raise RuntimeError("batch does not fit the captured neighbour widths: %d > %d" % (src.shape[1], dst.shape[1]))
        ops.pad_index_rows(src, shadow_src, dst, shadow_dst)


_REV_FUSED = os.environ.get("MVK_REV_FUSED", "1") == "1"      # development switch: 0 = two launches per reverse list, after its search


class DeviceInputChain:
    """The input side of one step -- centring, pyramid (oriented subsampling + neighbour searches),
    unprojection, 3-NN -- as a fixed sequence of launches with DEVICE-side counts that writes straight
    into a StaticBatch: no host read-back, no shape depends on data, so it can be captured as a branch of
    the network's hipGraph (the only way the two overlap on this runtime). Same results as
    build_batch + StaticBatch.load (tests/test_model_gpu.py). The per-level random grid orientations
    (datasets/common.py:89-108) are drawn on the host and copied into `rot` before each replay.
    `status` accumulates [max neighbour count, overflow]: overflow = a level outgrew its capacity or a
    query its in-kernel list; check it with ops.check_neighbor_status (loud, possibly one step late)."""

    def __init__(self, config, staged, limits, static):
        self.config, self.staged = config, staged
        self.plan = pyramid_plan(config)
        dev = staged['points'][0].device
        self.B = len(staged['points'])
        lens0 = [int(p.shape[0]) for p in staged['points']]
        L = len(static.points)
        self.lens = [torch.tensor(lens0, dtype=torch.int32, device=dev)] + \
                    [torch.zeros(self.B, dtype=torch.int32, device=dev) for _ in range(L - 1)]
        self.rot = torch.zeros((max(L - 1, 1), self.B, 3, 3), dtype=torch.float32, device=dev)
        self.rot_host = torch.zeros((max(L - 1, 1), self.B, 3, 3), dtype=torch.float32).pin_memory()
        self.status4 = torch.zeros(4, dtype=torch.int32, device=dev)      # [max neighbours, overflow | longest reverse row, overflow]
        self.status = self.status4[:2]
        self.rev_status = self.status4[2:]
        self.limits = [int(x) for x in limits]
        if static.neighbors[0].dtype != torch.int32:
            raise RuntimeError("DeviceInputChain writes int32 neighbour matrices")
        # size the shared workspaces once for the largest call (a regrown workspace would lose a reused grid)
        caps = static.caps
        ops._workspace("nb", max(ops.lib().mvk_radius_neighbors_workspace(caps[a], caps[b], self.B)
                                 for a in range(L) for b in range(L) if abs(a - b) <= 1), dev)
        ops._workspace("sub", ops.lib().mvk_grid_subsample_workspace(caps[0], self.B, 0, 0), dev)

    def _rev_counts(self, k, rows, device):
        """Counter slice of the k-th fused reverse list of a build: persistent zero words of its own (the lists of a pyramid
        are finished together at the end, so they cannot share counters like the two-launch form does)."""
        pool = self.__dict__.setdefault('_rev_count_pool', {})
        buf = pool.get(k)
        if buf is None or buf.numel() < rows:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("DeviceInputChain: build one batch eagerly before capturing (reverse-list counters)")
            buf = torch.zeros(int(rows), dtype=torch.int32, device=device)
            torch.cuda.current_stream(device).synchronize()
            pool[k] = buf
        return buf

    def draw_rotations(self, rotations=None, upload=True):
        """Host side of a step: one random rotation per cloud and level (or the given ones), staged in
        pinned memory and (upload) copied to the device asynchronously on the current stream. upload=False: the copy is
        a node of a captured graph (upload_rotations() under capture) that reads the pinned bytes at every replay."""
        L1 = self.rot.shape[0]
        for l in range(L1):
            R = random_grid_rotations(self.B) if rotations is None else rotations[l]
            self.rot_host[l].copy_(torch.from_numpy(np.ascontiguousarray(R, dtype=np.float32)))
        if upload:
            self.upload_rotations()

    def upload_rotations(self):
        self.rot.copy_(self.rot_host, non_blocking=True)

    def build(self, static):
        """Enqueues the whole chain on the current stream, writing into `static`."""
        cfg, st, plan, caps, lim = self.config, self.staged, self.plan, static.caps, self.limits
        world = st['points']
        stacked_world = torch.cat(world, 0)
        torch.cat([p - c for p, c in zip(world, st['center'])], 0, out=static.points[0][:static.n0])   # :600
        static._counts[0].fill_(static.n0)
        L = len(static.points)
        grid_of = (None, None)

        # Round 5: the searches fill the transposed matrices themselves and ONE launch finishes all of them (18 launches
        # -> 1, ops.reverse_finish_many); deterministic mode keeps the two-launch form, which also sorts the rows
        fused = _REV_FUSED and not ops.is_deterministic()
        pending = []

        def search(ql, sl, radius, out, limit_layer, rev=None, rev_shadow=None):
            nonlocal grid_of
            reuse = grid_of == (sl, np.float32(radius))
            grid_of = (sl, np.float32(radius))
            if rev is not None and fused and out.shape[1] <= 64:
                counts = self._rev_counts(len(pending), caps[sl], out.device)
                ops.radius_neighbors_dev(static.points[ql], static.points[sl], self.lens[ql], self.lens[sl], radius,
                                         out, caps[sl], self.status, reuse_grid=reuse, rev=rev, rev_counts=counts,
                                         rev_status=self.rev_status)
                pending.append((rev, counts, caps[sl], rev_shadow))
                return True
            ops.radius_neighbors_dev(static.points[ql], static.points[sl], self.lens[ql], self.lens[sl], radius,
                                     out, caps[sl], self.status, reuse_grid=reuse)
            return False

        for l in range(L):
            e = plan[l]
            if e['conv_r'] is not None:
                rv = static.rev_neighbors[l] if static.rev_neighbors else None
                done = search(l, l, e['conv_r'], static.neighbors[l], l, rev=rv, rev_shadow=caps[l])
                if static.orders and static.orders[l] is not None:     # the workspace holds this search's grid
                    ops.neighbors_cell_order(caps[l], caps[l], self.B, static.orders[l], self.lens[l])
                if rv is not None and not done:
                    ops.reverse_neighbors(static.neighbors[l], caps[l], out=rv, status=self.rev_status, shadow=caps[l])
            if e['pool'] and l + 1 < L:
                ops.grid_subsample_dev(static.points[l], self.lens[l], e['dl'], static.points[l + 1], self.lens[l + 1],
                                       self.status, rotations_dev=self.rot[l], total_out=static._counts[l + 1])
                rv = static.rev_pools[l] if static.rev_pools else None
                done = search(l + 1, l, e['pool_r'], static.pools[l], l, rev=rv, rev_shadow=caps[l + 1])
                if rv is not None and not done:
                    ops.reverse_neighbors(static.pools[l], caps[l], out=rv, status=self.rev_status, shadow=caps[l + 1])
                search(l, l + 1, e['up_r'], static.upsamples[l], l + 1)
                if l in static.rev_ups:
                    ops.reverse_neighbors(static.upsamples[l], caps[l + 1], out=static.rev_ups[l], status=self.rev_status,
                                          shadow=caps[l], first_column=True)
        if pending:
            ops.reverse_finish_many(pending, self.rev_status)
        torch.cat(st['labels'], 0, out=static.labels)
        ones = torch.ones_like(stacked_world[:, :1])
        z = stacked_world[:, 2:3]
        v = cfg.variant
        if v == "baseline":
            torch.cat([ones, z], 1, out=static.features)
            return
        torch.cat([ones, z] if v == "early" else [ones] + [torch.cat(st['colors'], 0)], 1, out=static.feature_3d)
        static.feat_aggre_points.copy_(stacked_world.unsqueeze(0))
        torch.stack(st['images'], 0, out=static.images)
        row0 = 0
        for i, pw in enumerate(world):
            xyz, valid = ops.unproject_depth(st['depth'][i], st['cam'][i], st['poses'][i])
            knn = ops.knn_pixels(pw, xyz, valid, k=3)
            static.knn_list[i].copy_(knn.unsqueeze(0))
            if getattr(static, 'knn_stacked', None) is not None:      # + the sphere's first view in the stacked feature map
                torch.add(knn, i * int(np.prod(st['depth'][i].shape)), out=static.knn_stacked[row0:row0 + pw.shape[0]])
            row0 += pw.shape[0]
            static.image_xyz[i].copy_(xyz.to(torch.float32))
