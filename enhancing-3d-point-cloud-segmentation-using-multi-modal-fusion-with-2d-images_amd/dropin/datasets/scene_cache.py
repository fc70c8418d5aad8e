"""On-disk formats either side of the sphere pipeline, all plain pickles:

* the preprocess cache of mvpnet/data/preprocess/preprocess.py:177-186 -- a list of
  ``{'scan_id', 'points' f32 (n,3), 'colors' u8 (n,3), 'seg_label'}`` dicts, one per scan;
* the per-scene files of KPConv-PyTorch/datasets/ScanNet_sphere_color.py:904-991 under
  ``input_<dl>/``: ``<scan>.pkl`` = ``{'sub_points', 'sub_labels', 'sub_colors', 'rgbd_dict'}`` and
  ``<scan>_proj.pkl`` = ``(proj_inds, labels)`` (:1074-1092), proj_inds = the nearest subsampled point of every
  original point (``reprojection_indices``: the exact float64 1-NN kernel instead of ``KDTree.query``, :1087-1089).
  The reference also pickles a scikit-learn KDTree per scene (``<scan>_KDTree.pkl``, :947, :985): sphere picking
  runs on the GPU ball query here (sphere_picking.PotentialSphereSampler) and does not need it, but
  ``save_search_tree`` writes the same pickle (host side, scikit-learn) so that a directory prepared here also
  serves the reference's loader, and a directory prepared by the reference loads as is.

subsample_scene is the compute step between the two (scene-load subsampling with colours as features and
labels, :935-948) on the HIP subsampling kernel."""
import os
import pickle

import numpy as np
import torch

try:
    from .._native import ops
except ImportError:      # flat layout (dropin/ on sys.path)
    from _native import ops

CACHE_KEYS = ('scan_id', 'points', 'colors', 'seg_label')
SCENE_KEYS = ('sub_points', 'sub_labels', 'sub_colors', 'rgbd_dict')


def load_preprocess_cache(path):
    """The list of per-scan dicts written by mvpnet's preprocess.py; checks the schema."""
    with open(path, 'rb') as f:
        data = pickle.load(f)
    for d in data:
        missing = [k for k in CACHE_KEYS if k not in d]
        if missing:
            raise ValueError('preprocess cache entry %r lacks %s' % (d.get('scan_id'), missing))
        if np.asarray(d['points']).ndim != 2 or np.asarray(d['points']).shape[1] != 3:
            raise ValueError('Wrong dimensions : points.shape is not (N, 3)')
    return data


def save_preprocess_cache(path, scans):
    with open(path, 'wb') as f:
        pickle.dump([{k: d[k] for k in CACHE_KEYS} for d in scans], f, protocol=pickle.HIGHEST_PROTOCOL)


def subsample_scene(scan, dl, label_map=None, device=None):
    """points / colors / seg_label of one cache entry -> the reference's scene-level arrays
    (ScanNet_sphere_color.py:935-948): float32 barycentres, float32 colours in [0,1] (barycentre / 255),
    int32 majority labels, in the reference's output order. label_map: the nyu40 -> ScanNet lookup
    applied to seg_label first (:938)."""
    dev = torch.device(device if device is not None else 'cuda')
    points = np.ascontiguousarray(scan['points'], dtype=np.float32)
    colors = np.ascontiguousarray(scan['colors'])
    labels = np.asarray(scan['seg_label'])
    if label_map is not None:
        labels = np.asarray(label_map)[labels]
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    p, _, f, l = ops.grid_subsample_batch(torch.from_numpy(points).to(dev), [points.shape[0]],
                                          features=torch.from_numpy(colors.astype(np.float32)).to(dev),
                                          labels=torch.from_numpy(labels).to(dev), dl=dl)
    # the division runs in NumPy like the reference's (:944): the tensor library divides by a scalar through a
    # reciprocal multiply, which is not the correctly rounded quotient
    return {'sub_points': p.cpu().numpy(), 'sub_colors': f.cpu().numpy() / 255,
            'sub_labels': np.squeeze(l.cpu().numpy())}


def scene_paths(tree_path, cloud_name):
    return {'scene': os.path.join(tree_path, '%s.pkl' % cloud_name),
            'kdtree': os.path.join(tree_path, '%s_KDTree.pkl' % cloud_name),
            'proj': os.path.join(tree_path, '%s_proj.pkl' % cloud_name)}


def save_scene(tree_path, cloud_name, sub_data, rgbd_dict=None):
    os.makedirs(tree_path, exist_ok=True)
    out = {'sub_points': sub_data['sub_points'], 'sub_labels': sub_data['sub_labels'],
           'sub_colors': sub_data['sub_colors'],
           'rgbd_dict': rgbd_dict if rgbd_dict is not None else sub_data.get('rgbd_dict', {'scan_id': cloud_name})}
    with open(scene_paths(tree_path, cloud_name)['scene'], 'wb') as f:
        pickle.dump(out, f)
    return out


def load_scene(tree_path, cloud_name):
    """{'sub_points', 'sub_labels', 'sub_colors', 'rgbd_dict'} of one scene; files written by the
    reference before it stored sub_points in the pickle (only in its KDTree) are rejected loudly."""
    with open(scene_paths(tree_path, cloud_name)['scene'], 'rb') as f:
        data = pickle.load(f)
    missing = [k for k in SCENE_KEYS if k not in data]
    if missing:
        raise ValueError('scene file of %s lacks %s' % (cloud_name, missing))
    return data


def save_projection(tree_path, cloud_name, proj_inds, labels):
    """<scan>_proj.pkl = (proj_inds, labels) (:1087-1092): nearest subsampled point of every original point."""
    os.makedirs(tree_path, exist_ok=True)
    with open(scene_paths(tree_path, cloud_name)['proj'], 'wb') as f:
        pickle.dump([np.asarray(proj_inds, dtype=np.int32), np.asarray(labels)], f)


def load_projection(tree_path, cloud_name):
    with open(scene_paths(tree_path, cloud_name)['proj'], 'rb') as f:
        proj_inds, labels = pickle.load(f)
    return proj_inds, labels


def reprojection_indices(sub_points, points, device=None):
    """proj_inds of a scene (ScanNet_sphere_color.py:1087-1089: ``KDTree(sub_points).query(points)``): for every
    ORIGINAL point the index of its nearest subsampled point, int32 [n]. Exact float64 distances of the float32
    coordinates like scikit-learn's tree (ties: lowest index), on the k-NN kernel of csrc/fusion.hip."""
    dev = torch.device(device if device is not None else 'cuda')
    keys = torch.as_tensor(np.ascontiguousarray(sub_points, dtype=np.float32)).to(dev)
    qs = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float32)).to(dev)
    if keys.shape[0] == 0:
        raise ValueError('reprojection_indices: no subsampled points')
    mask = torch.ones((1, keys.shape[0], 1), dtype=torch.bool, device=dev)
    nn = ops.knn_pixels(qs, keys.double().reshape(1, -1, 1, 3), mask, k=1)[:, 0]
    return nn.to(torch.int32).cpu().numpy()


def save_search_tree(tree_path, cloud_name, sub_points, leaf_size=10):
    """<scan>_KDTree.pkl = pickle of sklearn.neighbors.KDTree(sub_points, leaf_size=10) (:947, :985-986). Host-side
    file format only (needs scikit-learn, like the reference)."""
    from sklearn.neighbors import KDTree
    os.makedirs(tree_path, exist_ok=True)
    tree = KDTree(np.asarray(sub_points, dtype=np.float32), leaf_size=leaf_size)
    with open(scene_paths(tree_path, cloud_name)['kdtree'], 'wb') as f:
        pickle.dump(tree, f)
    return tree
