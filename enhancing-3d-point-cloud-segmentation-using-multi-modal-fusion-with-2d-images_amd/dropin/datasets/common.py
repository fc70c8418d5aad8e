"""Pyramid construction of MV-KPConv (reference KPConv-PyTorch/datasets/common.py): the wrappers
``grid_subsampling`` / ``batch_grid_subsampling`` / ``batch_neighbors`` (:44-196) and the per-layer
input builder ``segmentation_inputs_sphere`` (:779-900), device resident.

The reference builds the pyramid with NumPy + two CPU extensions inside DataLoader workers; here
the stacked points stay in HBM and every level is produced by the HIP kernels (one host sync per
subsampling level for the data-dependent point counts). NumPy inputs are accepted too (they are
staged once) and NumPy outputs can be requested with ``as_numpy=True`` for calibration code.
"""
import numpy as np
import torch

try:
    from .._native import ops
    from ..kernels.kernel_points import create_3D_rotations
except ImportError:
    from _native import ops
    from kernels.kernel_points import create_3D_rotations


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _t(a, dtype=None):
    if isinstance(a, torch.Tensor):
        return a if a.is_cuda else a.to(_dev())
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(_dev())


def grid_subsampling(points, features=None, labels=None, sampleDl=0.1, verbose=0):
    """common.py:44-74. Returns tensors for tensor input, NumPy for NumPy input."""
    as_np = not isinstance(points, torch.Tensor)
    n = points.shape[0]
    res = ops.grid_subsample_batch(_t(points, np.float32), [n],
                                   features=None if features is None else _t(features, np.float32),
                                   labels=None if labels is None else _t(labels, np.int32), dl=sampleDl)
    out = [res[0]] + list(res[2:])
    if as_np:
        out = [o.cpu().numpy() for o in out]
    return out[0] if len(out) == 1 else tuple(out)


def random_grid_rotations(B, rng=None):
    """One random rotation per cloud, float32 (common.py:89-108): the same three draws per call from
    the global NumPy RNG as the reference (or from ``rng`` when given)."""
    r = np.random if rng is None else rng
    theta = r.rand(B) * 2 * np.pi
    phi = (r.rand(B) - 0.5) * np.pi
    u = np.vstack([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)])
    alpha = r.rand(B) * 2 * np.pi
    return create_3D_rotations(u.T, alpha).astype(np.float32)


def batch_grid_subsampling(points, batches_len, features=None, labels=None, sampleDl=0.1, max_p=0, verbose=0,
                           random_grid_orient=True, R=None):
    """common.py:77-182: random per-cloud grid orientation, subsample, rotate the barycentres back.
    ``R`` (B,3,3 float32) overrides the random draw (used by the parity tests with captured matrices)."""
    as_np = not isinstance(points, torch.Tensor)
    pts = _t(points, np.float32)
    lens = np.ascontiguousarray(batches_len.cpu().numpy() if isinstance(batches_len, torch.Tensor) else batches_len,
                                dtype=np.int32)
    B = len(lens)
    if random_grid_orient and R is None:
        R = random_grid_rotations(B)
    res = ops.grid_subsample_batch(pts, lens, features=None if features is None else _t(features, np.float32),
                                   labels=None if labels is None else _t(labels, np.int32), dl=sampleDl, max_p=max_p,
                                   rotations=R if random_grid_orient else None)
    s_points, s_len = res[0], res[1]
    out = [s_points, s_len] + list(res[2:])
    if as_np:
        out = [o.cpu().numpy() if isinstance(o, torch.Tensor) else o for o in out]
    return tuple(out)


def batch_neighbors(queries, supports, q_batches, s_batches, radius, limit=None, status=None, reuse_grid=False):
    """common.py:185-196 (+ optional crop to ``limit`` columns = big_neighborhood_filter :411-421).
    status / reuse_grid: the enqueue-only mode of ops.radius_neighbors_batch."""
    as_np = not isinstance(queries, torch.Tensor)
    out = ops.radius_neighbors_batch(_t(queries, np.float32), _t(supports, np.float32), q_batches, s_batches,
                                     radius, limit=limit, status=status, reuse_grid=reuse_grid)
    return out.cpu().numpy() if as_np else out


def pyramid_plan(config):
    """The radii / cell sizes segmentation_inputs_sphere uses, one dict per pyramid layer
    (common.py:797-857): ``conv_r`` (None when the layer has no conv block), ``pool`` (whether a level
    follows), ``dl`` of the subsampling and the pool / upsample search radii."""
    r_normal = config.first_subsampling_dl * config.conv_radius
    plan, layer_blocks = [], []
    for block in config.architecture:
        if not ('pool' in block or 'strided' in block or 'global' in block or 'upsample' in block):
            layer_blocks.append(block)
            continue
        entry = dict(conv_r=None, pool=False, dl=None, pool_r=None, up_r=None)
        if layer_blocks:
            deform = bool(np.any(['deformable' in b for b in layer_blocks]))
            entry['conv_r'] = r_normal * config.deform_radius / config.conv_radius if deform else r_normal
        if 'pool' in block or 'strided' in block:
            r = r_normal * config.deform_radius / config.conv_radius if 'deformable' in block else r_normal
            entry.update(pool=True, dl=2 * r_normal / config.conv_radius, pool_r=r, up_r=2 * r)
        plan.append(entry)
        r_normal *= 2
        layer_blocks = []
        if 'global' in block or 'upsample' in block:
            break
    return plan


def segmentation_inputs_sphere(config, stacked_points, stack_lengths, neighborhood_limits=None,
                               index_dtype=torch.int64, rotations=None, status=None):
    """Per-layer network inputs (common.py:779-900): returns dict with lists ``points``, ``neighbors``,
    ``pools``, ``upsamples``, ``lengths`` (one entry per layer), all in HBM.

    neighborhood_limits: per-layer column caps (the calibrated 90th percentiles of the reference,
    :864-867); None keeps every neighbour. index_dtype: int64 like the reference (:874-876) or int32
    (half the index traffic; the kernels take both). rotations: optional list of (B,3,3) matrices, one
    per subsampling level, replacing the random grid orientation draws. status (int32 [2] HBM, needs
    neighborhood_limits): the 13 neighbour searches are only enqueued (full ``limit`` columns, no read-back
    per search; ops.check_neighbor_status(status) validates them later) and the searches that share
    supports and radius -- conv / pool of a level and the upsample search of the level above -- share
    one cell grid."""
    r_normal = config.first_subsampling_dl * config.conv_radius
    grid_of = [None, None]          # supports tensor and radius of the grid the neighbour workspace holds

    def neighbors(qp, sp, qb, sb, radius, layer):
        if status is None:
            return batch_neighbors(qp, sp, qb, sb, radius, limit=lim(layer)).to(index_dtype)
        reuse = grid_of[0] is sp and grid_of[1] == np.float32(radius)
        grid_of[0], grid_of[1] = sp, np.float32(radius)
        return batch_neighbors(qp, sp, qb, sb, radius, limit=lim(layer), status=status, reuse_grid=reuse).to(index_dtype)

    pts = _t(stacked_points, np.float32)
    lens = np.ascontiguousarray(stack_lengths.cpu().numpy() if isinstance(stack_lengths, torch.Tensor)
                                else stack_lengths, dtype=np.int32)
    out = dict(points=[], neighbors=[], pools=[], upsamples=[], lengths=[], deform_layers=[])
    layer_blocks = []
    level = 0
    dev = pts.device

    def lim(layer):
        if neighborhood_limits is None or len(neighborhood_limits) == 0:
            return None
        return int(neighborhood_limits[layer])

    def empty_idx():
        return torch.zeros((0, 1), dtype=index_dtype, device=dev)

    for block in config.architecture:
        if not ('pool' in block or 'strided' in block or 'global' in block or 'upsample' in block):
            layer_blocks.append(block)
            continue
        layer = len(out['points'])
        deform_layer = False
        if layer_blocks:
            if np.any(['deformable' in b for b in layer_blocks]):
                r = r_normal * config.deform_radius / config.conv_radius
                deform_layer = True
            else:
                r = r_normal
            conv_i = neighbors(pts, pts, lens, lens, r, layer)
        else:
            conv_i = empty_idx()
        if 'pool' in block or 'strided' in block:
            dl = 2 * r_normal / config.conv_radius
            R = rotations[level] if rotations is not None else None
            pool_p, pool_b = batch_grid_subsampling(pts, lens, sampleDl=dl, R=R)
            level += 1
            if 'deformable' in block:
                r = r_normal * config.deform_radius / config.conv_radius
                deform_layer = True
            else:
                r = r_normal
            pool_i = neighbors(pool_p, pts, pool_b, lens, r, layer)
            up_i = neighbors(pts, pool_p, lens, pool_b, 2 * r, layer + 1)
        else:
            pool_i, up_i = empty_idx(), empty_idx()
            pool_p = torch.zeros((0, 3), dtype=torch.float32, device=dev)
            pool_b = np.zeros((0,), dtype=np.int32)
        out['points'].append(pts)
        out['neighbors'].append(conv_i)
        out['pools'].append(pool_i)
        out['upsamples'].append(up_i)
        out['lengths'].append(torch.from_numpy(np.asarray(lens, dtype=np.int32)))
        out['deform_layers'].append(deform_layer)
        pts, lens = pool_p, pool_b
        r_normal *= 2
        layer_blocks = []
        if 'global' in block or 'upsample' in block:
            break
    return out


class SphereBatch:
    """Attribute contract of the reference's batch containers (ScanNetCustomBatch,
    datasets/ScanNet_sphere_color.py:1525-1619; baseline variant ScanNet_baseline_color.py:1198-1272):
    per-layer lists ``points / neighbors / pools / upsamples / lengths`` plus ``features`` (baseline) or
    ``feature_3d, feat_aggre_points, image_xyz, images, knn_list`` (fusion) and ``labels``."""

    def __init__(self, pyramid, labels, features=None, feature_3d=None, feat_aggre_points=None, image_xyz=None,
                 images=None, knn_list=None):
        self.points = pyramid['points']
        self.neighbors = pyramid['neighbors']
        self.pools = pyramid['pools']
        self.upsamples = pyramid['upsamples']
        self.lengths = pyramid['lengths']
        self.labels = labels
        self.features = features
        self.feature_3d = feature_3d
        self.feat_aggre_points = feat_aggre_points
        self.image_xyz = image_xyz
        self.images = images
        self.knn_list = knn_list

    def to(self, device):
        for name in ('points', 'neighbors', 'pools', 'upsamples', 'lengths'):
            setattr(self, name, [t.to(device) for t in getattr(self, name)])
        for name in ('labels', 'features', 'feature_3d', 'feat_aggre_points', 'image_xyz', 'images'):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, t.to(device))
        return self
