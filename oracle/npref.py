"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the floating-point part of
the MV-KPConv hot path. Pinned by tests/golden (generated from the reference's
own Python modules, see tests/golden/make_golden.py).

All paths below are relative to /root/reference/.
"""
import numpy as np


# ---------------------------------------------------------------------------
# KPConv (KPConv-PyTorch/models/blocks.py:237-374, rigid + deformable/modulated)
# ---------------------------------------------------------------------------

def kp_influence(d2, extent, influence):
    """blocks.py:329-346."""
    if influence == "constant":
        return np.ones_like(d2)
    if influence == "linear":
        return np.maximum(1 - np.sqrt(d2) / d2.dtype.type(extent), 0).astype(d2.dtype)
    if influence == "gaussian":
        sig = extent * 0.3
        return np.exp(-d2 / d2.dtype.type(2 * sig ** 2 + 1e-9)).astype(d2.dtype)  # blocks.py:69-76
    raise ValueError("Unknown influence function type (config.KP_influence)")


def kpconv_weights(q, s, idx, kp, extent, influence="linear", aggregation="sum", offsets=None):
    """Influence weights w[n,h,k] (blocks.py:277-351). Returns (w, d2).

    offsets: optional [Nq,K,3] already scaled by KP_extent (blocks.py:266,287)."""
    dt = q.dtype
    Ns = s.shape[0]
    sp = np.concatenate([s, np.full((1, 3), 1e6, dt)], 0)          # :277
    rel = sp[idx] - q[:, None, :]                                     # :280-283
    kpts = kp[None, None] if offsets is None else (offsets + kp)[:, None]   # :287-290
    diff = rel[:, :, None, :] - kpts                                  # :293-294
    d2 = np.sum(diff ** 2, axis=3, dtype=dt)                          # :297
    w = kp_influence(d2, extent, influence)
    if aggregation == "closest":                                      # :349-351
        nn = np.argmin(d2, axis=2)
        w = w * np.eye(kp.shape[0], dtype=dt)[nn]
    elif aggregation != "sum":
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    return w, d2


def kpconv_forward(q, s, idx, x, kp, W, extent, influence="linear", aggregation="sum",
                   offsets=None, modulations=None, return_A=False):
    """y[n,:] = sum_k (sum_h w[n,h,k] x+[idx[n,h]]) @ W[k]   (SURVEY.md A.4).

    The deformable neighbour re-compaction (blocks.py:300-325) only removes
    neighbours whose weight is exactly 0 under the linear influence, so the
    dense formula below is the same function (checked against the reference by
    the golden fixtures)."""
    w, d2 = kpconv_weights(q, s, idx, kp, extent, influence, aggregation, offsets)
    xp = np.concatenate([x, np.zeros((1, x.shape[1]), x.dtype)], 0)  # :357
    nx = xp[idx]                                                      # :360  [N,H,Cin]
    A = np.einsum("nhk,nhc->nkc", w, nx)                              # :363
    if modulations is not None:
        A = A * modulations[:, :, None]                               # :366-367
    y = np.einsum("nkc,kco->no", A, W)                                # :370-374
    return (y, A, w) if return_A else y


def kpconv_backward(q, s, idx, x, kp, W, extent, g, influence="linear", aggregation="sum"):
    """Rigid KPConv backward: returns (dx [Ns,Cin], dW [K,Cin,Cout])."""
    w, _ = kpconv_weights(q, s, idx, kp, extent, influence, aggregation)
    Ns = s.shape[0]
    xp = np.concatenate([x, np.zeros((1, x.shape[1]), x.dtype)], 0)
    A = np.einsum("nhk,nhc->nkc", w, xp[idx])
    dW = np.einsum("nkc,no->kco", A, g)
    dA = np.einsum("no,kco->nkc", g, W)
    contrib = np.einsum("nhk,nkc->nhc", w, dA)
    dxp = np.zeros_like(xp)
    np.add.at(dxp, idx.reshape(-1), contrib.reshape(-1, x.shape[1]))
    return dxp[:Ns], dW


# ---------------------------------------------------------------------------
# pooling helpers (blocks.py:79-133)
# ---------------------------------------------------------------------------

def max_pool(x, inds):
    xp = np.concatenate([x, np.zeros((1, x.shape[1]), x.dtype)], 0)   # :103
    return xp[inds].max(axis=1)                                       # :106-110


def closest_pool(x, inds):
    xp = np.concatenate([x, np.zeros((1, x.shape[1]), x.dtype)], 0)   # :88
    return xp[inds[:, 0]]                                             # :91


# ---------------------------------------------------------------------------
# fusion inputs (KPConv-PyTorch/datasets/ScanNet_sphere_color.py)
# ---------------------------------------------------------------------------

def depth2xyz(cam_matrix, depth):
    """ScanNet_sphere_color.py:66-72 (result is float64: int64 pixel grid x float32 inverse)."""
    v, u = np.indices(depth.shape)
    u, v = u.ravel(), v.ravel()
    uv1 = np.stack([u, v, np.ones_like(u)], axis=1)
    return (np.linalg.inv(cam_matrix[:3, :3]).dot(uv1.T) * depth.ravel()).T


def unproject_frames(cam_matrix, depths_mm, poses):
    """ScanNet_sphere_color.py:409-417 per frame: depth u16 mm -> world xyz (float64) + valid mask.

    depths_mm (nv,h,w) uint16, poses (nv,4,4) float32, cam_matrix float32 (already rescaled)."""
    xyz, mask = [], []
    for d, pose in zip(depths_mm, poses):
        depth = np.asarray(d, dtype=np.float32) / 1000.                 # :410
        p = depth2xyz(cam_matrix, depth)                               # :413
        m = p[:, 2] > 0                                                # :415
        p = np.matmul(p, pose[:3, :3].T) + pose[:3, 3]                 # :417
        xyz.append(p.reshape(d.shape + (3,)))
        mask.append(m.reshape(d.shape))
    return np.stack(xyz, 0), np.stack(mask, 0)


def knn_pixels(sphere_points, image_xyz, image_mask, k=3):
    """ScanNet_sphere_color.py:436-451: exact k-NN (float64) of every sphere point among
    the valid unprojected pixels, remapped to flat pixel index view*h*w + row*w + col."""
    from . import cport
    flat_xyz = image_xyz.reshape(-1, 3)
    ind_all = np.nonzero(image_mask.reshape(-1))[0]
    idx, _ = cport.knn_f64(sphere_points.astype(np.float64), flat_xyz[ind_all], k)
    return ind_all[idx].astype(np.int64)


# ---------------------------------------------------------------------------
# group_points (mvpnet/ops/group_points.py:20-31, cuda/group_points_kernel.cu:25-47,50-89;
# restatement in mvpnet/ops/tests/test_group_points.py:6-12)
# ---------------------------------------------------------------------------

def group_points(points, index):
    """points (B,C,N1), index (B,N2,K) -> (B,C,N2,K)."""
    B, Cc, _ = points.shape
    out = np.empty((B, Cc) + index.shape[1:], points.dtype)
    for b in range(B):
        out[b] = points[b][:, index[b]]
    return out


def group_points_backward(grad_out, index, n1):
    B, Cc = grad_out.shape[:2]
    gi = np.zeros((B, Cc, n1), grad_out.dtype)
    for b in range(B):
        for c in range(Cc):
            np.add.at(gi[b, c], index[b].reshape(-1), grad_out[b, c].reshape(-1))
    return gi


# ---------------------------------------------------------------------------
# sphere picking (KPConv-PyTorch/datasets/ScanNet_sphere_color.py:556-597)
# ---------------------------------------------------------------------------

def ball_members(points, center, radius):
    """sklearn KDTree.query_radius membership: float64 rdist = dx^2+dy^2+dz^2 (in that order) <= r^2;
    returns ascending indices and rdist."""
    p = np.asarray(points, np.float64)
    c = np.asarray(center, np.float64).reshape(3)
    d = p - c
    rd = np.zeros(p.shape[0])
    rd += d[:, 0] * d[:, 0]
    rd += d[:, 1] * d[:, 1]
    rd += d[:, 2] * d[:, 2]
    idx = np.nonzero(rd <= radius * radius)[0]
    return idx, rd[idx]


def sphere_pick(pot_points, potentials, min_pot, argmin_pot, input_points, in_radius, mask_margin=0.1):
    """One critical-section iteration (:556-584) + the input-region queries (:592-597), numpy.
    potentials / min_pot / argmin_pot are updated in place; returns (cloud_ind, point_ind, center,
    input_inds, mask_inds) with ascending indices."""
    cloud_ind = int(np.argmin(min_pot))
    point_ind = int(argmin_pot[cloud_ind])
    center = np.asarray(pot_points[cloud_ind], np.float64)[point_ind]
    idx, rd = ball_members(pot_points[cloud_ind], center, in_radius)
    d2s = np.square(np.sqrt(rd))                                   # query_radius returns sqrt(rdist); :576 squares it
    tukeys = np.square(1 - d2s / np.square(in_radius))             # :578
    tukeys[d2s > np.square(in_radius)] = 0                         # :579
    potentials[cloud_ind][idx] += tukeys                           # :581
    m = int(np.argmin(potentials[cloud_ind]))
    min_pot[cloud_ind] = potentials[cloud_ind][m]
    argmin_pot[cloud_ind] = m
    inp, _ = ball_members(input_points[cloud_ind], center, in_radius)
    msk, _ = ball_members(input_points[cloud_ind], center, in_radius + mask_margin)
    return cloud_ind, point_ind, center, inp, msk
