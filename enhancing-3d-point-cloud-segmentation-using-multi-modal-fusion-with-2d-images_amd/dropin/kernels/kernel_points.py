"""Kernel-point dispositions (reference KPConv-PyTorch/kernels/kernel_points.py).

``load_kernels`` reproduces the reference's behaviour (:408-490): cached disposition (here an
.npy data file next to this module instead of a PLY under the CWD), one global-``np.random`` draw
for the z-rotation, N(0, 0.01) noise, scale by the radius, rotate, cast to float32 -- so that with
the same ``np.random.seed`` the kernel points are identical to the reference's. When no cached
disposition exists one is created with a small repulsion optimiser (the published KPConv recipe,
written from the paper; the reference's optimiser is :258-405).
"""
import os

import numpy as np

_DISPO_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dispositions")


def create_3D_rotations(axis, angle):
    """Rotation matrices from axes [N,3] and angles [N] (Rodrigues form; reference :44-75)."""
    axis = np.asarray(axis)
    angle = np.asarray(angle)
    c, s = np.cos(angle), np.sin(angle)
    v = 1 - c
    x, y, z = axis[:, 0], axis[:, 1], axis[:, 2]
    R = np.stack([c + v * x * x, v * x * y - s * z, v * x * z + s * y,
                  v * x * y + s * z, c + v * y * y, v * y * z - s * x,
                  v * x * z - s * y, v * y * z + s * x, c + v * z * z], axis=1)
    return R.reshape(-1, 3, 3)


def optimize_disposition(num_kpoints, dimension=3, fixed="center", iters=2000, seed=0):
    """Unit-ball kernel disposition by gradient descent on the KPConv energy: pairwise 1/d repulsion
    plus a quadratic attraction to the centre; 'center' pins point 0 at the origin, 'verticals'
    additionally pins points 1, 2 on the vertical axis."""
    rng = np.random.RandomState(seed)
    p = rng.rand(num_kpoints, dimension) * 2 - 1
    p = p[np.linalg.norm(p, axis=1) < 1][:num_kpoints]
    while p.shape[0] < num_kpoints:
        q = rng.rand(num_kpoints, dimension) * 2 - 1
        p = np.concatenate([p, q[np.linalg.norm(q, axis=1) < 1]], 0)[:num_kpoints]
    if fixed in ("center", "verticals"):
        p[0] = 0
    if fixed == "verticals":
        p[1:3] = 0
        p[1, -1], p[2, -1] = 0.5, -0.5
    step = 0.01
    for it in range(iters):
        d = p[:, None, :] - p[None]
        n2 = (d ** 2).sum(-1) + 1e-9
        np.fill_diagonal(n2, np.inf)
        grad = -(d / n2[..., None] ** 1.5).sum(1) + 2 * p * 0.5
        if fixed in ("center", "verticals"):
            grad[0] = 0
        if fixed == "verticals":
            grad[1:3, :-1] = 0
        gn = np.linalg.norm(grad, axis=1, keepdims=True)
        p -= step * grad / np.maximum(gn, 1e-9) * np.minimum(gn, 1.0)
        step *= 0.9995
    r = np.linalg.norm(p, axis=1).max()
    return p / max(r, 1e-9) * 0.66


def load_kernels(radius, num_kpoints, dimension, fixed, lloyd=False):
    name = "k_{:03d}_{:s}_{:d}D.npy".format(num_kpoints, fixed, dimension)
    path = os.path.join(_DISPO_DIR, name)
    if os.path.exists(path):
        kernel_points = np.load(path)
    else:
        kernel_points = optimize_disposition(num_kpoints, dimension, fixed)
        try:
            os.makedirs(_DISPO_DIR, exist_ok=True)
            np.save(path, kernel_points)
        except OSError:
            pass
    # same sequence of global-RNG draws as the reference (:454-484)
    R = np.eye(dimension)
    theta = np.random.rand() * 2 * np.pi
    if dimension == 2:
        if fixed != "vertical":
            c, s = np.cos(theta), np.sin(theta)
            R = np.array([[c, -s], [s, c]], dtype=np.float32)
    elif dimension == 3:
        if fixed != "vertical":
            c, s = np.cos(theta), np.sin(theta)
            R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
        else:
            phi = (np.random.rand() - 0.5) * np.pi
            u = np.array([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)])
            alpha = np.random.rand() * 2 * np.pi
            R = create_3D_rotations(u.reshape(1, -1), np.reshape(alpha, (1,)))[0].astype(np.float32)
    kernel_points = kernel_points + np.random.normal(scale=0.01, size=kernel_points.shape)
    kernel_points = radius * kernel_points
    kernel_points = np.matmul(kernel_points, R)
    return kernel_points.astype(np.float32)
