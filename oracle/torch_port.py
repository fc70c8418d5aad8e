"""TEST INFRASTRUCTURE ONLY -- unfused PyTorch-ops restatement of the network side of the hot path,
run on the CPU: KPConv.forward as the reference's ~15 ATen ops (KPConv-PyTorch/models/blocks.py:
277-374), the blocks (:430-694), KPFCNN wiring (models/architectures.py:322-343 and the three
architectures_sphere*.py forward passes), group_points (mvpnet/ops/cuda/group_points_kernel.cu:
41-44 = expand + gather) and FeatureAggregation (mvpnet/models/mvpnet_3d.py:40-64).

It is functional: it takes the PRODUCT model's ``state_dict`` (same parameter names as the
reference) and a batch of CPU tensors, so that (a) tests can compare the HIP path end to end with
identical weights and (b) bench.py can time "the reference's PyTorch path on the host cores"
(cpu_baseline.kind = "port"). Pinned against the reference's own modules by tests/golden G4/G5.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- KPConv, op by op (blocks.py:237-374)

def kpconv_unfused(q_pts, s_pts, neighb_inds, x, kernel_points, weights, KP_extent, KP_influence='linear',
                   aggregation_mode='sum', offset_features=None, modulated=False, K=15):
    """Returns (y, min_d2, deformed_KP). offset_features: [N, 3K(+K)] for the deformable variants."""
    deformable = offset_features is not None
    modulations = None
    if deformable:
        if modulated:
            unscaled = offset_features[:, :3 * K].view(-1, K, 3)
            modulations = 2 * torch.sigmoid(offset_features[:, 3 * K:])
        else:
            unscaled = offset_features.view(-1, K, 3)
        offsets = unscaled * KP_extent
    s_pts = torch.cat((s_pts, torch.zeros_like(s_pts[:1, :]) + 1e6), 0)            # :277
    neighbors = s_pts[neighb_inds, :]                                               # :280
    neighbors = neighbors - q_pts.unsqueeze(1)                                      # :283
    min_d2 = deformed_KP = None
    if deformable:
        deformed_KP = offsets + kernel_points                                       # :287
        deformed_K_points = deformed_KP.unsqueeze(1)
    else:
        deformed_K_points = kernel_points
    differences = neighbors.unsqueeze(2) - deformed_K_points                        # :293-294
    sq_distances = torch.sum(differences ** 2, dim=3)                               # :297
    if deformable:
        min_d2, _ = torch.min(sq_distances, dim=1)                                  # :303
        in_range = torch.any(sq_distances < KP_extent ** 2, dim=2).type(torch.int32)
        new_max_neighb = torch.max(torch.sum(in_range, dim=1))
        neighb_row_bool, neighb_row_inds = torch.topk(in_range, new_max_neighb.item(), dim=1)
        new_neighb_inds = neighb_inds.gather(1, neighb_row_inds, sparse_grad=False)
        neighb_row_inds = neighb_row_inds.unsqueeze(2).expand(-1, -1, K)
        sq_distances = sq_distances.gather(1, neighb_row_inds, sparse_grad=False)
        new_neighb_inds = new_neighb_inds * neighb_row_bool.long()
        new_neighb_inds = new_neighb_inds - (neighb_row_bool.type(torch.int64) - 1) * int(s_pts.shape[0] - 1)
    else:
        new_neighb_inds = neighb_inds
    if KP_influence == 'constant':                                                  # :330-344
        all_weights = torch.ones_like(sq_distances)
    elif KP_influence == 'linear':
        all_weights = torch.clamp(1 - torch.sqrt(sq_distances) / KP_extent, min=0.0)
    elif KP_influence == 'gaussian':
        all_weights = torch.exp(-sq_distances / (2 * (KP_extent * 0.3) ** 2 + 1e-9))
    else:
        raise ValueError('Unknown influence function type (config.KP_influence)')
    all_weights = torch.transpose(all_weights, 1, 2)
    if aggregation_mode == 'closest':                                               # :349-351
        nn1 = torch.argmin(sq_distances, dim=2)
        all_weights = all_weights * torch.transpose(F.one_hot(nn1, K), 1, 2).float()
    elif aggregation_mode != 'sum':
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    x = torch.cat((x, torch.zeros_like(x[:1, :])), 0)                               # :357
    neighb_x = x[new_neighb_inds]                                                   # :360
    weighted = torch.matmul(all_weights, neighb_x)                                  # :363
    if deformable and modulated:
        weighted = weighted * modulations.unsqueeze(2)                              # :366-367
    weighted = weighted.permute((1, 0, 2))                                          # :370
    return torch.sum(torch.matmul(weighted, weights), dim=0), min_d2, deformed_KP   # :371-374


# ---------------------------------------------------------------- blocks, functional over a state dict

class _Net:
    """Walks the architecture like the reference's module tree does, reading parameters by their
    reference names from a flat state dict; BatchNorm in training mode (batch statistics)."""

    def __init__(self, sd, config, training=True):
        self.sd = sd
        self.c = config
        self.training = training
        self.reg_terms = []     # (min_d2, deformed_KP, extent) of deformable KPConvs, for the regulariser
        self.trace = {}         # block prefix -> output (for layer-by-layer comparisons in tests)

    def bn(self, x, prefix):
        if self.c.use_batch_norm:
            p = prefix + '.batch_norm.'
            return F.batch_norm(x, self.sd[p + 'running_mean'].clone(), self.sd[p + 'running_var'].clone(),
                                self.sd[p + 'weight'], self.sd[p + 'bias'], self.training,
                                self.c.batch_norm_momentum, 1e-5)
        return x + self.sd[prefix + '.bias']

    def unary(self, x, prefix, use_bn, relu=True):
        x = F.linear(x, self.sd[prefix + '.mlp.weight'])
        if use_bn:
            x = self.bn(x, prefix + '.batch_norm')
        else:
            x = x + self.sd[prefix + '.batch_norm.bias']
        return F.leaky_relu(x, 0.1) if relu else x

    def kpconv(self, prefix, q, s, inds, x, radius, deformable):
        c = self.c
        extent = radius * c.KP_extent / c.conv_radius
        off_feat = None
        if deformable:
            po = prefix + '.offset_conv'
            off_feat, _, _ = kpconv_unfused(q, s, inds, x, self.sd[po + '.kernel_points'], self.sd[po + '.weights'],
                                            extent, c.KP_influence, c.aggregation_mode, K=c.num_kernel_points)
            off_feat = off_feat + self.sd[prefix + '.offset_bias']
        y, min_d2, dkp = kpconv_unfused(q, s, inds, x, self.sd[prefix + '.kernel_points'], self.sd[prefix + '.weights'],
                                        extent, c.KP_influence, c.aggregation_mode, off_feat, c.modulated,
                                        c.num_kernel_points)
        if deformable:
            self.reg_terms.append((min_d2, dkp, extent))
        return y

    def block(self, name, prefix, x, batch, layer, radius):
        y = self._block(name, prefix, x, batch, layer, radius)
        self.trace[prefix] = y
        return y

    def _block(self, name, prefix, x, batch, layer, radius):
        if name == 'unary':
            return self.unary(x, prefix, self.c.use_batch_norm)
        if name == 'nearest_upsample':
            xp = torch.cat((x, torch.zeros_like(x[:1, :])), 0)
            return xp[batch['upsamples'][layer - 1][:, 0]]                          # blocks.py:79-91
        if 'strided' in name:
            q, s, inds = batch['points'][layer + 1], batch['points'][layer], batch['pools'][layer]
        else:
            q, s, inds = batch['points'][layer], batch['points'][layer], batch['neighbors'][layer]
        deform = 'deform' in name
        if name.startswith('simple'):
            y = self.kpconv(prefix + '.KPConv', q, s, inds, x, radius, deform)
            return F.leaky_relu(self.bn(y, prefix + '.batch_norm'), 0.1)
        if name.startswith('resnetb'):
            feats = x
            if (prefix + '.unary1.mlp.weight') in self.sd:
                x = self.unary(x, prefix + '.unary1', self.c.use_batch_norm)
            x = self.kpconv(prefix + '.KPConv', q, s, inds, x, radius, deform)
            x = F.leaky_relu(self.bn(x, prefix + '.batch_norm_conv'), 0.1)
            x = self.unary(x, prefix + '.unary2', self.c.use_batch_norm, relu=False)
            if 'strided' in name:                                                   # max_pool, blocks.py:94-110
                fp = torch.cat((feats, torch.zeros_like(feats[:1, :])), 0)
                shortcut = fp[inds].max(1)[0]
            else:
                shortcut = feats
            if (prefix + '.unary_shortcut.mlp.weight') in self.sd:
                shortcut = self.unary(shortcut, prefix + '.unary_shortcut', self.c.use_batch_norm, relu=False)
            return F.leaky_relu(x + shortcut, 0.1)
        raise ValueError('Unknown block name in the architecture definition : ' + name)

    def encoder(self, x, batch, prefix='encoder_blocks'):
        c = self.c
        layer, r = 0, c.first_subsampling_dl * c.conv_radius
        skips, skip_at = [], []
        for i, name in enumerate(c.architecture):
            if any(t in name for t in ('pool', 'strided', 'upsample', 'global')):
                skip_at.append(i)
            if 'upsample' in name:
                break
            if i in skip_at:
                skips.append(x)
            x = self.block(name, '%s.%d' % (prefix, i), x, batch, layer, r)
            if 'pool' in name or 'strided' in name:
                layer += 1
                r *= 2
        return x, skips, layer, r

    def decoder(self, x, skips, batch, layer, r):
        c = self.c
        start = next(i for i, n in enumerate(c.architecture) if 'upsample' in n)
        for j, name in enumerate(c.architecture[start:]):
            if j > 0 and 'upsample' in c.architecture[start + j - 1]:
                x = torch.cat([x, skips.pop()], dim=1)
            x = self.block(name, 'decoder_blocks.%d' % j, x, batch, layer, r)
            if 'upsample' in name:
                layer -= 1
                r *= 0.5
        return x


def group_points(points, index):
    """expand + gather (group_points_kernel.cu:41-44; test restatement test_group_points.py:6-12)."""
    b, c, n1 = points.shape
    _, n2, k = index.shape
    return points.unsqueeze(2).expand(b, c, n2, n1).gather(3, index.unsqueeze(1).expand(b, c, n2, k))


def feature_aggregation(sd, prefix, src_xyz, tgt_xyz, feature, training=True, n_layers=3):
    """mvpnet_3d.py:40-64 with SharedMLP = (1x1 conv, BN2d, ReLU) x 3, reduction = sum over k."""
    diff = src_xyz - tgt_xyz.unsqueeze(-1)
    dist = torch.sum(diff ** 2, dim=1, keepdim=True)
    x = torch.cat([feature, diff, dist], dim=1)
    for i in range(n_layers):
        p = '%s.mlp.%d.' % (prefix, i)
        x = F.conv2d(x, sd[p + 'conv.weight'])
        x = F.batch_norm(x, sd[p + 'bn.running_mean'].clone(), sd[p + 'bn.running_var'].clone(),
                         sd[p + 'bn.weight'], sd[p + 'bn.bias'], training, 0.1, 1e-5)
        x = F.relu(x)
    return torch.sum(x, 3)


def lift_2d(sd, batch, net_2d, training=True):
    """architectures_sphere.py:246-284 with a given 2D encoder module (CPU, eval, frozen). When the
    batch carries 'feature_2d' (the 2D encoder's output computed elsewhere, (b*nv, 64, h, w)) the
    encoder is skipped -- it is a PyTorch library network outside the hot path's kernels."""
    images = batch['images']
    b, nv, _, h, w = images.shape
    if batch.get('feature_2d') is not None:
        f2d = batch['feature_2d']
    else:
        with torch.no_grad():
            f2d = net_2d({'image': images.reshape(-1, 3, h, w)})['feature']
    f2d = f2d.reshape(b, nv, -1, h, w).transpose(1, 2).contiguous().reshape(b, -1, nv * h * w)
    xyz = batch['image_xyz'].permute(0, 4, 1, 2, 3).reshape(b, 3, nv * h * w)
    fl, xl = [], []
    for i in range(b):
        knn = batch['knn_list'][i].long()
        fl.append(group_points(f2d[i:i + 1], knn))
        xl.append(group_points(xyz[i:i + 1], knn))
    f = feature_aggregation(sd, 'feat_aggreg', torch.cat(xl, 2), batch['feat_aggre_points'].transpose(1, 2),
                            torch.cat(fl, 2), training)
    return f.permute(0, 2, 1).reshape(-1, 64)


def forward(sd, config, batch, net_2d=None, training=True, trace=None):
    """Logits of the variant named by config.variant; returns (logits, regulariser terms).
    trace: optional dict that receives every block's output keyed by its module path."""
    net = _Net(sd, config, training)
    if trace is not None:
        net.trace = trace
    v = config.variant
    if v == 'baseline':
        x = batch['features']
    else:
        f2d3d = lift_2d(sd, batch, net_2d, training)
    if v == 'early':
        x = torch.cat((batch['feature_3d'], f2d3d), dim=1).detach()
    if v in ('baseline', 'early', 'late'):
        if v == 'late':
            x = batch['feature_3d']
        x, skips, layer, r = net.encoder(x, batch)
        x = net.decoder(x, skips, batch, layer, r)
        if v == 'late':
            x = net.unary(x, 'transform_mlp', False)
            x = torch.cat((x, f2d3d), dim=1)
    elif v == 'middle':
        ones = torch.ones_like(batch['feat_aggre_points'][0, :, :1])
        x2 = torch.cat((ones, f2d3d), dim=1).detach()
        x3, skips3, layer, r = net.encoder(batch['feature_3d'], batch, 'encoder_blocks_3d')
        # 2d encoder: skip features are concatenated onto the 3d ones in order
        x2o, skips2, _, _ = net.encoder(x2, batch, 'encoder_blocks_2d')
        skips = [torch.cat([a, b], dim=1) for a, b in zip(skips3, skips2)]
        x = torch.mean(torch.stack([x3, x2o]), 0)
        x = net.decoder(x, skips, batch, layer, r)
    x = net.unary(x, 'head_mlp', False)
    x = net.unary(x, 'head_softmax', False)
    return x, net.reg_terms


def loss_fn(logits, labels, reg_terms, config):
    """CrossEntropy(ignore_index=-1) + p2p_fitting_regularizer (architectures.py:25-58, :345-378).
    Labels are assumed to be already in [0, C-1] (synthetic data has no ignored labels)."""
    out = F.cross_entropy(logits.t().unsqueeze(0), labels.unsqueeze(0), ignore_index=-1)
    fit = rep = 0
    K = config.num_kernel_points
    for min_d2, dkp, extent in reg_terms:
        fit = fit + F.l1_loss(min_d2 / (extent ** 2), torch.zeros_like(min_d2))
        locs = dkp / extent
        for i in range(K):
            other = torch.cat([locs[:, :i, :], locs[:, i + 1:, :]], dim=1).detach()
            d = torch.sqrt(torch.sum((other - locs[:, i:i + 1, :]) ** 2, dim=2))
            rl = torch.sum(torch.clamp_max(d - config.repulse_extent, max=0.0) ** 2, dim=1)
            rep = rep + F.l1_loss(rl, torch.zeros_like(rl)) / K
    return out + config.deform_fitting_power * (2 * fit + rep)


def batch_to_cpu(batch):
    """Product SphereBatch (HBM) -> dict of CPU tensors with int64 indices."""
    d = {}
    for name in ('points', 'neighbors', 'pools', 'upsamples'):
        d[name] = [t.detach().cpu().long() if 'point' not in name else t.detach().cpu() for t in getattr(batch, name)]
    for name in ('labels', 'features', 'feature_3d', 'feat_aggre_points', 'image_xyz', 'images'):
        t = getattr(batch, name, None)
        if t is not None:
            d[name] = t.detach().cpu()
    if getattr(batch, 'knn_list', None) is not None:
        d['knn_list'] = [k.detach().cpu() if isinstance(k, torch.Tensor) else torch.from_numpy(k) for k in batch.knn_list]
    return d
