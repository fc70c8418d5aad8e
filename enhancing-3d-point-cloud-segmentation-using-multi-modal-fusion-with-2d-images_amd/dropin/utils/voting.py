"""Test-time voting, metrics and frame selection of MV-KPConv, device resident.

* ``vote_update``          smoothed probability accumulation  test_probs[c][inds] = a*old + (1-a)*new
                           (reference KPConv-PyTorch/utils/tester.py:166-185, test_smooth = 0.95)
* ``confusion`` / ``IoU_from_confusions``   (utils/metrics.py:35-80 fast_confusion, :206-232)
* ``select_frames``        greedy maximum-coverage frame choice on a bool overlap table
                           (datasets/ScanNet_sphere_color.py:53-63)
* ``frame_overlaps``       which base points each RGB-D frame sees: nearest base point within 0.1 of every
                           valid unprojected pixel (datasets/get_rgbd_overlap_subcloud.py:68-138; open3d's
                           hybrid 1-NN search there, the exact float64 k-NN kernel of csrc/fusion.hip here)
All tensors stay in HBM; only scalars come back to the host.
"""
import numpy as np
import torch

try:
    from .._native import ops
except ImportError:
    from _native import ops


def vote_update(test_probs, inds, probs, smooth=0.95, points=None, radius_ratio=None, in_radius=None):
    """In place on test_probs [N_cloud, C]; inds [n] int64 (indices of the sphere's points in the cloud),
    probs [n, C]. Optional inner-sphere mask (tester.py:176-179): only points with |p|^2 <
    (radius_ratio * in_radius)^2 vote."""
    if radius_ratio is not None and 0 < radius_ratio < 1:
        mask = torch.sum(points ** 2, dim=1) < (radius_ratio * in_radius) ** 2
        inds, probs = inds[mask], probs[mask]
    test_probs[inds] = smooth * test_probs[inds] + (1 - smooth) * probs
    return test_probs


def confusion(true, pred, num_classes):
    """[C,C] int64 confusion matrix, rows = truth, columns = prediction (metrics.py fast_confusion with
    label_values = 0..C-1)."""
    true, pred = true.reshape(-1).long(), pred.reshape(-1).long()
    return torch.bincount(true * num_classes + pred, minlength=num_classes ** 2).reshape(num_classes, num_classes)


def IoU_from_confusions(confusions):
    """metrics.py:206-232 on tensors or arrays ([..., C, C] -> [..., C])."""
    c = torch.as_tensor(confusions).double()
    TP = torch.diagonal(c, dim1=-2, dim2=-1)
    TP_plus_FN = c.sum(-1)
    TP_plus_FP = c.sum(-2)
    IoU = TP / (TP_plus_FP + TP_plus_FN - TP + 1e-6)
    mask = (TP_plus_FN < 1e-3).double()
    counts = (1 - mask).sum(-1, keepdim=True)
    mIoU = IoU.sum(-1, keepdim=True) / (counts + 1e-6)
    return IoU + mask * mIoU


def select_frames(rgbd_overlap, num_rgbd_frames):
    """Greedy max-coverage (ScanNet_sphere_color.py:53-63): repeatedly take the frame that covers most
    still-uncovered base points. rgbd_overlap [nb, nf] bool (tensor or array); returns a list of ints."""
    ov = torch.as_tensor(rgbd_overlap).clone().bool()
    selected = []
    for _ in range(num_rgbd_frames):
        frame_idx = int(torch.argmax(ov.sum(0)))          # first maximum, like np.argmax
        selected.append(frame_idx)
        ov[ov[:, frame_idx].clone()] = False
    return selected


def frame_overlaps(base_points, depth_mm, cam_matrix, poses, radius=0.1):
    """overlaps [nb, nf] bool: base point b is the nearest base point (within `radius`) of at least one
    valid unprojected pixel of frame f. base_points [nb,3] f32 (HBM), depth_mm (nf,h,w) int16/uint16 (HBM),
    poses (nf,4,4) f32 (HBM)."""
    nb = base_points.shape[0]
    xyz, valid = ops.unproject_depth(depth_mm, cam_matrix, poses)          # (nf,h,w,3) f64
    nf = xyz.shape[0]
    out = torch.zeros((nb, nf), dtype=torch.bool, device=base_points.device)
    keys = base_points.double().reshape(1, nb, 1, 3)                        # "image" of nb keys, all valid
    kmask = torch.ones((1, nb, 1), dtype=torch.bool, device=base_points.device)
    for f in range(nf):
        pix = xyz[f].reshape(-1, 3)[valid[f].reshape(-1)]
        if pix.shape[0] == 0:
            continue
        nn = ops.knn_pixels(pix.float(), keys, kmask, k=1)[:, 0]
        d2 = ((pix - base_points.double()[nn]) ** 2).sum(1)
        hit = nn[d2 <= radius * radius]
        out[hit, f] = True
    return out
