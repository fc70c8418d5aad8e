"""FeatureAggregation (reference mvpnet/models/mvpnet_3d.py:12-70): per (point, neighbour) feature
[feat_2d | dxyz | |dxyz|^2] -> SharedMLP (1x1 conv + BN + ReLU) -> reduction over k.
Sub-module names (mlp.{i}.conv / mlp.{i}.bn) match the reference so MVPNet checkpoints load."""
import torch
from torch import nn

try:
    from ...common.nn import SharedMLP, xavier_uniform
except ImportError:
    from common.nn import SharedMLP, xavier_uniform


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

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
                xavier_uniform(m)
