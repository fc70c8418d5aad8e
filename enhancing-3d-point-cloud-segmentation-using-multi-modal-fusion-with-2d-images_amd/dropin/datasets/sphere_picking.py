"""Potentials-based sphere picking of the MV-KPConv dataset (reference
KPConv-PyTorch/datasets/ScanNet_sphere_color.py:556-597 inside ``potential_item``), device resident:
arg-min of the sampling potentials -> sphere centre -> Tukey update of the potentials around it ->
indices of the input points inside the sphere (and inside the slightly larger mask sphere).

The reference keeps one sklearn KDTree per cloud and calls ``query_radius``; here the radius queries
are ordered-compaction scans (csrc/fusion.hip: ball_*_kernel) with the same float64 membership test
(rdist <= r^2), and the potentials live in HBM. Index ORDER differs from the reference (ascending here,
KD-tree traversal order there); the SET is identical (tests/golden g7 from sklearn itself).
"""
import numpy as np
import torch

try:
    from .._native import ops
except ImportError:
    from _native import ops


class PotentialSphereSampler:

    def __init__(self, pot_points, input_points, in_radius, mask_margin=0.1, init_potentials=None, rng=None):
        """pot_points / input_points: lists (one per cloud) of [n,3] float32 arrays or tensors (coarse
        potential points, ScanNet_sphere_color.py:1039-1042, and the dl-subsampled input clouds).
        init_potentials: optional list of float64 arrays; default rand * 1e-3 like the reference (:330-334)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        to_dev = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, np.float32))).to(dev)
        self.pot_points = [to_dev(p) for p in pot_points]
        self.input_points = [to_dev(p) for p in input_points]
        self.in_radius = float(in_radius)
        self.mask_margin = float(mask_margin)
        r = np.random if rng is None else rng
        if init_potentials is None:
            init_potentials = [r.rand(p.shape[0]) * 1e-3 for p in self.pot_points]
        self.potentials = [torch.from_numpy(np.ascontiguousarray(p, np.float64)).to(dev) for p in init_potentials]
        mins = [torch.min(p, 0) for p in self.potentials]
        self.min_potentials = torch.stack([m.values for m in mins])          # (:335-341)
        self.argmin_potentials = torch.stack([m.indices for m in mins])

    def pick(self):
        """One iteration of the critical section (:556-584) + the two input-region queries (:592-597).
        Returns dict(cloud_ind, point_ind, center (np.float64 [3]), input_inds, mask_inds) with the index
        tensors in HBM (int64, ascending)."""
        cloud_ind = int(torch.argmin(self.min_potentials))
        point_ind = int(self.argmin_potentials[cloud_ind])
        center = self.pot_points[cloud_ind][point_ind].double().cpu().numpy()
        pot = self.potentials[cloud_ind]
        ops.tukey_update(self.pot_points[cloud_ind], center, self.in_radius, pot)
        min_ind = torch.argmin(pot)
        self.min_potentials[cloud_ind] = pot[min_ind]
        self.argmin_potentials[cloud_ind] = min_ind
        pts = self.input_points[cloud_ind]
        return dict(cloud_ind=cloud_ind, point_ind=point_ind, center=center,
                    input_inds=ops.ball_query(pts, center, self.in_radius),
                    mask_inds=ops.ball_query(pts, center, self.in_radius + self.mask_margin))
