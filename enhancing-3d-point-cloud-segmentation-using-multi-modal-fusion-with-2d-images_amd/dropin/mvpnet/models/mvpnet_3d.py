"""FeatureAggregation (reference mvpnet/models/mvpnet_3d.py:12-70): per (point, neighbour) feature
[feat_2d | dxyz | |dxyz|^2] -> SharedMLP (1x1 conv + BN + ReLU) -> reduction over k.
Sub-module names (mlp.{i}.conv / mlp.{i}.bn) match the reference so MVPNet checkpoints load."""
import torch
from torch import nn
import torch.nn.functional as F

try:
    from ...common.nn import SharedMLP, xavier_uniform
    from ..._native import ops
except ImportError:
    from common.nn import SharedMLP, xavier_uniform
    from _native import ops


class FeatureAggregation(nn.Module):
    """Feature Aggregation inspired by ContFuse"""

    def __init__(self, in_channels, mlp_channels=(64, 64, 64), reduction='sum', use_relation=True):
        super(FeatureAggregation, self).__init__()
        self.in_channels = in_channels
        self.use_relation = use_relation
        if mlp_channels:
            self.out_channels = mlp_channels[-1]
            self.mlp = SharedMLP(in_channels + (4 if use_relation else 0), mlp_channels, ndim=2, bn=True)
        else:
            self.out_channels = in_channels
            self.mlp = None
        if reduction == 'sum':
            self.reduction = torch.sum
        elif reduction == 'max':
            self.reduction = lambda x, dim: torch.max(x, dim)[0]
        self.reset_parameters()

    def forward(self, src_xyz, tgt_xyz, feature):
        """src_xyz (b,3,np,k), tgt_xyz (b,3,np), feature (b,c,np,k) -> (b,out,np)."""
        if self.mlp is None:
            return self.reduction(feature, 3)
        x = feature
        if self.use_relation:
            diff_xyz = src_xyz - tgt_xyz.unsqueeze(-1)
            distance = torch.sum(diff_xyz ** 2, dim=1, keepdim=True)
            x = torch.cat([feature, diff_xyz, distance], dim=1)
        return self.reduction(self.mlp(x), 3)

    def forward_fused(self, feature_2d, image_xyz, knn, points):
        """Same function as forward() fed by the two group_points calls of the network
        (architectures_sphere.py:266-284), for ONE sphere and without materialising the grouped tensors:
        one HIP gather kernel builds the (C+4, np*k) MLP input straight from the encoder's (nv,C,h,w)
        feature map, the three 1x1 convolutions run on the f32 MFMA GEMM over rows, BatchNorm over the
        np*k rows (identical statistics to BatchNorm2d over (1,C,np,k)), sum over k.
        feature_2d (nv,C,h,w), image_xyz (nv,h,w,3), knn (np,k) int64, points (np,3) -> (np, out)."""
        if self.mlp is None or not self.use_relation or self.reduction is not torch.sum:
            raise RuntimeError("forward_fused covers the MV-KPConv configuration (relation features, MLP, sum)")
        n_pts, k = knn.shape
        x = ops.fa_gather(feature_2d, image_xyz, knn, points)           # [C+4, np*k] channel-major
        transposed = True
        for layer in self.mlp:
            w = layer.conv.weight
            bn = layer.bn
            rows = x.shape[1] if transposed else x.shape[0]
            if bn.training:   # HIP BatchNorm + ReLU over the np*k rows (all valid unless the level is capacity padded);
                n_valid = ops.row_count_for(rows)          # its column statistics come out of the GEMM's epilogue
                if n_valid is None:
                    n_valid = ops.full_count(rows, x.device)
                x = ops.linear(x, w.view(w.shape[0], w.shape[1]), x_is_transposed=transposed, stats_n_valid=n_valid)
                x = ops.bn_lrelu(x, n_valid, bn, 0.0)
                transposed = False
                continue
            x = ops.linear(x, w.view(w.shape[0], w.shape[1]), x_is_transposed=transposed)   # [np*k, out]
            transposed = False
            x = F.relu(F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, bn.momentum, bn.eps))
        return x.view(n_pts, k, -1).sum(dim=1)

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
                xavier_uniform(m)
