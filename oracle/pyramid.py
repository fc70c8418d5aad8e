"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the pyramid builder
(KPConv-PyTorch/datasets/common.py:77-182 batch_grid_subsampling incl. the random grid
rotation, :185-196 batch_neighbors, :411-421 big_neighborhood_filter, :779-900
segmentation_inputs_sphere) on top of the C oracle (or the compiled reference core, impl="ref")."""
import numpy as np

from . import cport


def create_3D_rotations(axis, angle):
    """kernels/kernel_points.py:44-75 (Rodrigues form)."""
    c, s = np.cos(angle), np.sin(angle)
    v = 1 - c
    x, y, z = axis[:, 0], axis[:, 1], axis[:, 2]
    R = np.stack([c + v * x * x, v * x * y - s * z, v * x * z + s * y,
                  v * x * y + s * z, c + v * y * y, v * y * z - s * x,
                  v * x * z - s * y, v * y * z + s * x, c + v * z * z], axis=1)
    return R.reshape(-1, 3, 3)


def draw_rotations(B, rng=np.random):
    """common.py:89-108: three draws of size B from the global RNG."""
    theta = rng.rand(B) * 2 * np.pi
    phi = (rng.rand(B) - 0.5) * np.pi
    u = np.vstack([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)])
    alpha = rng.rand(B) * 2 * np.pi
    return create_3D_rotations(u.T, alpha).astype(np.float32)


def batch_grid_subsampling(points, lens, dl, R=None, impl="oracle"):
    """common.py:110-135 (points-only path)."""
    pts = points
    if R is not None:
        pts = points.copy()
        i0 = 0
        for bi, length in enumerate(lens):
            pts[i0:i0 + length, :] = np.sum(np.expand_dims(pts[i0:i0 + length, :], 2) * R[bi], axis=1)   # :118
            i0 += length
    s_points, s_len = cport.subsample_batch(pts, lens, dl=dl, impl=impl)
    if R is not None:
        i0 = 0
        for bi, length in enumerate(s_len):
            s_points[i0:i0 + length, :] = np.sum(np.expand_dims(s_points[i0:i0 + length, :], 2) * R[bi].T, axis=1)  # :134
            i0 += length
    return s_points, s_len


def segmentation_inputs(config, stacked_points, stack_lengths, limits=None, rotations=None, impl="oracle"):
    """common.py:779-900. rotations: list of (B,3,3) float32 per subsampling level (None = no rotation)."""
    r_normal = config.first_subsampling_dl * config.conv_radius
    pts, lens = stacked_points, np.asarray(stack_lengths, np.int32)
    out = dict(points=[], neighbors=[], pools=[], upsamples=[], lengths=[])
    layer_blocks, level = [], 0

    def crop(m, layer):
        return m if limits is None or len(limits) == 0 else m[:, :limits[layer]]

    for block in config.architecture:
        if not ('pool' in block or 'strided' in block or 'global' in block or 'upsample' in block):
            layer_blocks.append(block)
            continue
        layer = len(out['points'])
        if layer_blocks:
            r = r_normal * config.deform_radius / config.conv_radius if any('deformable' in b for b in layer_blocks) else r_normal
            conv_i = cport.radius_neighbors_batch(pts, pts, lens, lens, r, impl=impl)
        else:
            conv_i = np.zeros((0, 1), np.int32)
        if 'pool' in block or 'strided' in block:
            dl = 2 * r_normal / config.conv_radius
            R = rotations[level] if rotations is not None else None
            pool_p, pool_b = batch_grid_subsampling(pts, lens, dl, R, impl)
            level += 1
            r = r_normal * config.deform_radius / config.conv_radius if 'deformable' in block else r_normal
            pool_i = cport.radius_neighbors_batch(pool_p, pts, pool_b, lens, r, impl=impl)
            up_i = cport.radius_neighbors_batch(pts, pool_p, lens, pool_b, 2 * r, impl=impl)
        else:
            pool_i, up_i = np.zeros((0, 1), np.int32), np.zeros((0, 1), np.int32)
            pool_p, pool_b = np.zeros((0, 3), np.float32), np.zeros((0,), np.int32)
        conv_i, pool_i = crop(conv_i, layer), crop(pool_i, layer)
        if up_i.shape[0] > 0:
            up_i = crop(up_i, layer + 1)
        out['points'].append(pts)
        out['neighbors'].append(conv_i.astype(np.int64))
        out['pools'].append(pool_i.astype(np.int64))
        out['upsamples'].append(up_i.astype(np.int64))
        out['lengths'].append(lens)
        pts, lens = pool_p, pool_b
        r_normal *= 2
        layer_blocks = []
        if 'global' in block or 'upsample' in block:
            break
    return out
