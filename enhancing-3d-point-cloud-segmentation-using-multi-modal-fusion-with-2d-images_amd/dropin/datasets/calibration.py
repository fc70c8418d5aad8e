"""Sampler calibration of the reference's ScanNetSampler.calibration
(KPConv-PyTorch/datasets/ScanNet_sphere_color.py:1272-1521):

* batch_limit -- the cap on stacked points per batch -- is steered by a proportional controller
  (Kp = 100) until the low-passed number of spheres per batch sits on config.batch_num
  (time constant 10 batches, 100 once within 1 of the target; converged when the last ten smoothed
  errors are all below 0.1);
* neighborhood_limits -- per layer, the neighbour count that leaves `untouched_ratio` of the
  neighbourhoods uncropped -- from a histogram of the real-neighbour counts of every batch seen.

Both are cached in ``batch_limits.pkl`` / ``neighbors_limits.pkl`` under the dataset path with the
reference's keys, so either implementation picks up the other's files. The histogram is accumulated on
the GPU (the index matrices already live there); the controller is host arithmetic."""
import os
import pickle

import numpy as np
import torch


def batch_limit_key(config, use_potentials=True):
    return '{:s}_{:.3f}_{:.3f}_{:d}'.format('potentials' if use_potentials else 'random', config.in_radius,
                                            config.first_subsampling_dl, config.batch_num)


def neighbor_limit_keys(config):
    keys = []
    for layer in range(config.num_layers):
        dl = config.first_subsampling_dl * (2 ** layer)
        r = dl * (config.deform_radius if config.deform_layers[layer] else config.conv_radius)
        keys.append('{:.3f}_{:.3f}'.format(dl, r))
    return keys


def _load_dict(path):
    if os.path.exists(path):
        with open(path, 'rb') as f:
            return pickle.load(f)
    return {}


def load_calibration(config, path, use_potentials=True):
    """(batch_limit or None, neighborhood_limits or None) from the cache files of `path`."""
    b = _load_dict(os.path.join(path, 'batch_limits.pkl')).get(batch_limit_key(config, use_potentials))
    d = _load_dict(os.path.join(path, 'neighbors_limits.pkl'))
    keys = neighbor_limit_keys(config)
    limits = [d[k] for k in keys] if all(k in d for k in keys) else None
    return b, limits


def save_calibration(config, path, batch_limit, neighborhood_limits, use_potentials=True):
    os.makedirs(path, exist_ok=True)
    bfile, nfile = os.path.join(path, 'batch_limits.pkl'), os.path.join(path, 'neighbors_limits.pkl')
    bd, nd = _load_dict(bfile), _load_dict(nfile)
    bd[batch_limit_key(config, use_potentials)] = float(batch_limit)
    for k, v in zip(neighbor_limit_keys(config), neighborhood_limits):
        nd[k] = v
    with open(bfile, 'wb') as f:
        pickle.dump(bd, f)
    with open(nfile, 'wb') as f:
        pickle.dump(nd, f)


class Calibrator:
    """Feed it the batches a sampler produces; read `batch_limit` before drawing the next one.

        cal = Calibrator(config, batch_limit=start)
        while not cal.converged:
            batch = make_batch(limit=cal.batch_limit)
            cal.update(batch.neighbors, n_spheres)
        limits = cal.neighborhood_limits(0.9)
    """

    def __init__(self, config, batch_limit, Kp=100.0):
        self.target = config.batch_num
        self.batch_limit = float(batch_limit)
        self.Kp = float(Kp)
        self.hist_n = int(np.ceil(4 / 3 * np.pi * (config.deform_radius + 1) ** 3))      # :1383
        self.hists = None                                   # [num_layers, hist_n] int64, on the GPU once known
        self.num_layers = config.num_layers
        self.estim_b, self.low_pass_T, self.finer = 0.0, 10, False
        self.smooth_errors, self.converged, self.steps = [], False, 0

    def update(self, neighbors, n_spheres):
        """neighbors: the per-layer index matrices of one batch; n_spheres: its number of stacked clouds."""
        for layer, nb in enumerate(neighbors[:self.num_layers]):
            if nb.numel() == 0:
                continue
            if not isinstance(nb, torch.Tensor):
                nb = torch.as_tensor(nb)
            if self.hists is None:
                self.hists = torch.zeros((self.num_layers, self.hist_n), dtype=torch.int64, device=nb.device)
            counts = (nb < nb.shape[0]).sum(dim=1)          # the reference compares with the row count (:1414)
            h = torch.bincount(counts, minlength=self.hist_n)[:self.hist_n]
            self.hists[layer] += h.to(self.hists.device)
        b = float(n_spheres)
        self.estim_b += (b - self.estim_b) / self.low_pass_T                     # :1422
        self.smooth_errors = (self.smooth_errors + [self.target - self.estim_b])[-10:]
        self.batch_limit += self.Kp * (self.target - b)                          # :1433
        if not self.finer and abs(self.estim_b - self.target) < 1:               # :1436-1438
            self.low_pass_T, self.finer = 100, True
        if self.finer and max(abs(e) for e in self.smooth_errors) < 0.1:         # :1441-1443
            self.converged = True
        self.steps += 1
        return self.batch_limit

    def neighborhood_limits(self, untouched_ratio=0.9):
        """:1462-1464: per layer, the number of histogram bins whose cumulated count stays below the ratio."""
        h = self.hists.cpu().numpy() if self.hists is not None else np.zeros((self.num_layers, self.hist_n), np.int64)
        cumsum = np.cumsum(h.T, axis=0)
        return np.sum(cumsum < (untouched_ratio * cumsum[self.hist_n - 1, :]), axis=0)
