"""Pyramid construction of MV-KPConv (reference KPConv-PyTorch/datasets/common.py): the wrappers
``grid_subsampling`` / ``batch_grid_subsampling`` / ``batch_neighbors`` (:44-196) and the per-layer
input builder ``segmentation_inputs_sphere`` (:779-900), device resident.

The reference builds the pyramid with NumPy + two CPU extensions inside DataLoader workers; here
the stacked points stay in HBM and every level is produced by the HIP kernels (one host sync per
subsampling level for the data-dependent point counts). NumPy inputs are accepted too (they are
staged once) and NumPy outputs can be requested with ``as_numpy=True`` for calibration code.
"""
import os

import numpy as np
import torch

from torch.utils.data import Dataset

try:
    from .._native import ops
    from ..kernels.kernel_points import create_3D_rotations
    from ..utils.config import Config
except ImportError:
    from _native import ops
    from kernels.kernel_points import create_3D_rotations
    from utils.config import Config


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
    one cell grid.

    Not in the reference: ``orders`` -- per layer the level's rows sorted by the cells of its conv search's grid
    (ops.neighbors_cell_order; None for a layer without conv blocks), the work list the KPConv gather walks
    (MVK_GATHER_ORDER=0: no lists). The reference's flat batch list has no slot for them: each list is also
    remembered under its points tensor (ops.remember_work_order), where the blocks find it as long as that tensor
    reaches the network as it is (device-resident batches in the process that built them).
    ``rev_neighbors`` / ``rev_pools`` -- per layer the transposed conv / pool neighbour matrix of the rigid layers
    (ops.reverse_neighbors; None elsewhere): the feature gradient of those convolutions then runs as a gather in a fixed
    summation order instead of an atomic scatter (MVK_REVERSE_DX=0: none). With ``status`` they are built at a fixed width
    and ``rev_status`` collects their overflow word (ops.check_reverse_status)."""
    r_normal = config.first_subsampling_dl * config.conv_radius
    want_orders = os.environ.get("MVK_GATHER_ORDER", "1") != "0"
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
    out = dict(points=[], neighbors=[], pools=[], upsamples=[], lengths=[], deform_layers=[], orders=[],
               rev_neighbors=[], rev_pools=[])
    want_rev = ops.REVERSE_DX
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
            order = None
            if want_orders and pts.shape[0] > 0 and conv_i.shape[1] > 0:      # the workspace holds this search's grid
                order = ops.neighbors_cell_order(pts.shape[0], pts.shape[0], len(lens), device=pts.device)
                ops.remember_work_order(pts, order)      # for batch containers without an `orders` attribute
        else:
            conv_i = empty_idx()
            order = None
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
        # transposed relations for the gather-form feature gradient (ops.reverse_neighbors): rigid convolutions, and since
        # round 5 the deformable ones too (searched at the wide deform radius: rows of hundreds of entries;
        # MVK_REVERSE_DX_DEFORM=0 keeps their atomic scatter)
        rigid = want_rev and pts.is_cuda and (not deform_layer or ops.REVERSE_DX_DEFORM)

        def reverse(m):
            if not rigid or m.shape[0] == 0 or m.shape[1] == 0:
                return None
            if status is None:
                return ops.reverse_neighbors(m, pts.shape[0])               # exact width (one read-back)
            if 'rev_status' not in out:                                     # sync-free pyramid: fixed width, one status word
                out['rev_status'] = torch.zeros(2, dtype=torch.int32, device=dev)
            return ops.reverse_neighbors(m, pts.shape[0], width=min(ops.reverse_width_cap(), 2 * m.shape[1] + 16),
                                         status=out['rev_status'])

        out['rev_neighbors'].append(reverse(conv_i))
        out['rev_pools'].append(reverse(pool_i))
        if out['rev_neighbors'][-1] is not None:
            ops.remember_reverse(conv_i, out['rev_neighbors'][-1])          # for batch containers without the attribute
        if out['rev_pools'][-1] is not None:
            ops.remember_reverse(pool_i, out['rev_pools'][-1])              # max_pool's backward finds it by the matrix
        if ops.is_deterministic() and want_rev and pts.is_cuda and up_i.shape[0] > 0 and up_i.shape[1] > 0:
            # deterministic mode: the nearest-upsampling scatter of the decoder as a gather too
            if status is None:
                rev_up = ops.reverse_neighbors(up_i, pool_p.shape[0], first_column=True)
            else:
                if 'rev_status' not in out:
                    out['rev_status'] = torch.zeros(2, dtype=torch.int32, device=dev)
                rev_up = ops.reverse_neighbors(up_i, pool_p.shape[0], width=64, status=out['rev_status'], first_column=True)
            ops.remember_reverse(up_i, rev_up, first_column=True)
            out.setdefault('rev_ups', {})[layer] = rev_up
        out['points'].append(pts)
        out['neighbors'].append(conv_i)
        out['pools'].append(pool_i)
        out['upsamples'].append(up_i)
        out['lengths'].append(torch.from_numpy(np.asarray(lens, dtype=np.int32)))
        out['deform_layers'].append(deform_layer)
        out['orders'].append(order)
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
        self.orders = pyramid.get('orders')          # work lists of the KPConv gather (None: row order)
        self.rev_neighbors = pyramid.get('rev_neighbors')    # transposed neighbour matrices (gather-form feature gradient)
        self.rev_pools = pyramid.get('rev_pools')
        self.rev_status = pyramid.get('rev_status')      # sync-free pyramids: ops.check_reverse_status(batch.rev_status)
        self.rev_ups = pyramid.get('rev_ups')            # deterministic mode: {layer: transposed first column of upsamples[layer]}
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
        if self.orders:
            self.orders = [None if t is None else t.to(device) for t in self.orders]
        for name in ('rev_neighbors', 'rev_pools'):
            if getattr(self, name):
                setattr(self, name, [None if t is None else t.to(device) for t in getattr(self, name)])
        for name in ('labels', 'features', 'feature_3d', 'feat_aggre_points', 'image_xyz', 'images'):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, t.to(device))
        return self


# ---------------------------------------------------------------- batch containers (a17)

def _tensor(a):
    """NumPy array (what the reference's collate hands over) or tensor (CPU or already in HBM) -> tensor."""
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


def _pin(t):
    return t.pin_memory() if (not t.is_cuda and not t.is_pinned() and torch.cuda.is_available()) else t


class _FlatListBatch:
    """Shared part of the two reference containers: the flat ``input_list`` of one collated batch is
    ``points[L] + neighbors[L] + pools[L] + upsamples[L] + lengths[L] + tail`` (common.py:897-898 /
    :648-650 plus what potential_item appends, ScanNet_sphere_color.py:812), L = (len - TAIL) // 5."""
    TAIL = ()             # attribute names of the tail, in list order
    PLAIN = ()            # tail entries that stay Python objects (knn_list)

    def __init__(self, input_list):
        input_list = input_list[0]                                  # drop the DataLoader's batch dimension
        L = (len(input_list) - len(self.TAIL)) // 5
        if L < 1 or 5 * L + len(self.TAIL) != len(input_list):
            raise ValueError("batch list of %d entries is not 5*L+%d" % (len(input_list), len(self.TAIL)))
        for j, name in enumerate(('points', 'neighbors', 'pools', 'upsamples', 'lengths')):
            setattr(self, name, [_tensor(a) for a in input_list[j * L:(j + 1) * L]])
        for j, name in enumerate(self.TAIL):
            a = input_list[5 * L + j]
            setattr(self, name, a if name in self.PLAIN else _tensor(a))

    def _map(self, fn):
        for name in ('points', 'neighbors', 'pools', 'upsamples', 'lengths'):
            setattr(self, name, [fn(t) for t in getattr(self, name)])
        for name in self.TAIL:
            if name not in self.PLAIN:
                setattr(self, name, fn(getattr(self, name)))
        return self

    def pin_memory(self):
        """Manual pinning (ScanNet_sphere_color.py:1574-1597); tensors that already live in HBM stay there."""
        return self._map(_pin)

    def to(self, device):
        return self._map(lambda t: t.to(device, non_blocking=True))

    # ScanNet_sphere_color.py:1623-1690
    def unstack_points(self, layer=None):
        return self.unstack_elements('points', layer)

    def unstack_neighbors(self, layer=None):
        return self.unstack_elements('neighbors', layer)

    def unstack_pools(self, layer=None):
        return self.unstack_elements('pools', layer)

    def unstack_elements(self, element_name, layer=None, to_numpy=True):
        """Per-cloud pieces of one stacked list; neighbour / pool indices are rebased to their own cloud and
        shadow entries become -1 (on copies: the reference edits the batch's own tensors in place)."""
        if element_name not in ('points', 'neighbors', 'pools'):
            raise ValueError('Unknown element name: {:s}'.format(element_name))
        elements = self.pools[:-1] if element_name == 'pools' else getattr(self, element_name)
        all_p_list = []
        for layer_i, layer_elems in enumerate(elements):
            if layer is None or layer == layer_i:
                i0, p_list = 0, []
                lengths = self.lengths[layer_i + 1] if element_name == 'pools' else self.lengths[layer_i]
                for b_i, length in enumerate(lengths):
                    length = int(length)
                    elem = layer_elems[i0:i0 + length]
                    if element_name == 'neighbors':
                        elem = elem.clone()
                        elem[elem >= self.points[layer_i].shape[0]] = -1
                        elem[elem >= 0] -= i0
                    elif element_name == 'pools':
                        elem = elem.clone()
                        elem[elem >= self.points[layer_i].shape[0]] = -1
                        elem[elem >= 0] -= int(torch.sum(self.lengths[layer_i][:b_i]))
                    i0 += length
                    p_list.append(elem.cpu().numpy() if to_numpy else elem)
                if layer == layer_i:
                    return p_list
                all_p_list.append(p_list)
        return all_p_list


class ScanNetCustomBatch(_FlatListBatch):
    """Fusion batch (datasets/ScanNet_sphere_color.py:1525-1619): ``ScanNetCustomBatch(input_list)`` with
    L = (len - 11) // 5; accepts the reference's NumPy arrays as well as tensors already in HBM."""
    TAIL = ('feat_aggre_points', 'image_xyz', 'images', 'labels', 'scales', 'rots', 'cloud_inds', 'center_inds',
            'input_inds', 'knn_list', 'feature_3d')
    PLAIN = ('knn_list',)
    features = None


class ScanNetBaselineCustomBatch(_FlatListBatch):
    """Baseline batch (datasets/ScanNet_baseline_color.py:1198-1272): L = (len - 7) // 5."""
    TAIL = ('features', 'labels', 'scales', 'rots', 'cloud_inds', 'center_inds', 'input_inds')
    feature_3d = feat_aggre_points = image_xyz = images = knn_list = None


def ScanNetCollate(batch_data):
    """collate_fn of the loaders (ScanNet_sphere_color.py:1693-1694)."""
    return ScanNetCustomBatch(batch_data)


# ---------------------------------------------------------------- dataset parent class (common.py:205-900)

class PointCloudDataset(Dataset):
    """Parent class of the reference's datasets (datasets/common.py:205-900) with the pyramid builders on
    the HIP kernels. The ``*_inputs`` methods return the reference's flat list; arrays are NumPy (as the
    reference's) unless ``self.device_resident`` is set, in which case the per-layer tensors stay in HBM
    (the batch containers above take either)."""

    device_resident = False

    def __init__(self, name):
        self.name = name
        self.path = ''
        self.label_to_names = {}
        self.num_classes = 0
        self.label_values = np.zeros((0,), dtype=np.int32)
        self.label_names = []
        self.label_to_idx = {}
        self.name_to_label = {}
        self.config = Config()
        self.neighborhood_limits = []

    def __len__(self):
        return 0

    def __getitem__(self, idx):
        return 0

    def init_labels(self):
        """common.py:239-250 (the ScanNet benchmark's 21 names are hard-wired there)."""
        self.num_classes = len(self.label_to_names)
        self.label_values = np.sort([k for k, v in self.label_to_names.items()])
        self.label_names = ['unclassified', 'wall', 'floor', 'cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window',
                            'bookshelf', 'picture', 'counter', 'desk', 'curtain', 'refridgerator', 'showercurtain', 'toilet',
                            'sink', 'bathtub', 'otherfurniture']
        self.label_to_idx = {l: i for i, l in enumerate(self.label_values)}

    def _draw_augmentation(self, n, dim):
        """The draws of common.py:259-308 in the reference's order from the global NumPy RNG: rotation
        (vertical: 1 draw, all: 3), scale (anisotropic: dim draws; isotropic: 1 -- the reference computes
        ``rand*(max-min) - min`` there, kept), symmetries (dim draws), noise (n*dim normals)."""
        c = self.config
        R = np.eye(dim)
        if dim == 3:
            if c.augment_rotation == 'vertical':
                theta = np.random.rand() * 2 * np.pi
                cs, sn = np.cos(theta), np.sin(theta)
                R = np.array([[cs, -sn, 0], [sn, cs, 0], [0, 0, 1]], dtype=np.float32)
            elif c.augment_rotation == 'all':
                theta = np.random.rand() * 2 * np.pi
                phi = (np.random.rand() - 0.5) * np.pi
                u = np.array([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)])
                alpha = np.random.rand() * 2 * np.pi
                R = create_3D_rotations(np.reshape(u, (1, -1)), np.reshape(alpha, (1, -1)))[0]
        R = R.astype(np.float32)
        lo, hi = c.augment_scale_min, c.augment_scale_max
        if c.augment_scale_anisotropic:
            scale = np.random.rand(dim) * (hi - lo) + lo
        else:
            scale = np.random.rand() * (hi - lo) - lo
        sym = np.array(c.augment_symmetries).astype(np.int32)
        sym *= np.random.randint(2, size=dim)
        scale = (scale * (1 - sym * 2)).astype(np.float32)
        noise = (np.random.randn(n, dim) * c.augment_noise).astype(np.float32)
        return R, scale, noise

    @staticmethod
    def _augment_normals(normals, R, scale):
        ns = scale[[1, 2, 0]] * scale[[2, 0, 1]]
        an = np.dot(normals, R) * ns
        return an * (1 / (np.linalg.norm(an, axis=1, keepdims=True) + 1e-6))

    def augmentation_transform(self, points, normals=None, verbose=False):
        """common.py:252-329."""
        R, scale, noise = self._draw_augmentation(points.shape[0], points.shape[1])
        aug = np.sum(np.expand_dims(points, 2) * R, axis=1) * scale + noise
        if normals is None:
            return aug, scale, R
        return aug, self._augment_normals(normals, R, scale), scale, R

    def augmentation_transform_new(self, points, image_xyz, normals=None):
        """common.py:331-409: the same transform applied to the unprojected pixels (without the noise)."""
        R, scale, noise = self._draw_augmentation(points.shape[0], points.shape[1])
        aug = np.sum(np.expand_dims(points, 2) * R, axis=1) * scale + noise
        shape = image_xyz.shape
        aug_xyz = (np.sum(np.expand_dims(image_xyz.reshape([-1, 3]), 2) * R, axis=1) * scale).reshape(shape)
        if normals is None:
            return aug, scale, R, aug_xyz
        return aug, self._augment_normals(normals, R, scale), scale, R, aug_xyz

    def big_neighborhood_filter(self, neighbors, layer):
        """common.py:411-421."""
        if len(self.neighborhood_limits) > 0:
            return neighbors[:, :self.neighborhood_limits[layer]]
        return neighbors

    def _pyramid_list(self, stacked_points, stack_lengths):
        pyr = segmentation_inputs_sphere(self.config, stacked_points, stack_lengths,
                                         self.neighborhood_limits if len(self.neighborhood_limits) > 0 else None,
                                         torch.int64)
        groups = [pyr['points'], pyr['neighbors'], pyr['pools'], pyr['upsamples'], pyr['lengths']]
        if not self.device_resident:
            groups = [[t.cpu().numpy() for t in g] for g in groups]
        return [t for g in groups for t in g]

    def segmentation_inputs(self, stacked_points, stacked_features, labels, stack_lengths):
        """common.py:536-650: ``points + neighbors + pools + upsamples + lengths + [features, labels]``."""
        return self._pyramid_list(stacked_points, stack_lengths) + [stacked_features, labels]

    def segmentation_inputs_sphere(self, stacked_points, stacked_image_xyz, stacked_images, stacked_feature_points,
                                   labels, stack_lengths):
        """common.py:779-900: ``... + [feature points, image_xyz, images, labels]``."""
        return self._pyramid_list(stacked_points, stack_lengths) + [stacked_feature_points, stacked_image_xyz,
                                                                     stacked_images, labels]
