"""MV-KPConv middle fusion (reference KPConv-PyTorch/models/architectures_sphere_middle_fusion.py):
two encoders (3D features / lifted 2D features), skip features concatenated, bottleneck features
averaged (:308), one decoder."""
import torch
import torch.nn as nn

try:
    from .architectures import (p2p_fitting_regularizer, build_encoder, build_decoder, run_decoder,
                                _SegmentationLossMixin)
    from .blocks import UnaryBlock
    from .fusion_common import build_2d_branch, lift_2d_features
except ImportError:
    from models.architectures import (p2p_fitting_regularizer, build_encoder, build_decoder, run_decoder,
                                      _SegmentationLossMixin)
    from models.blocks import UnaryBlock
    from models.fusion_common import build_2d_branch, lift_2d_features


class KPFCNN_featureAggre(_SegmentationLossMixin, nn.Module):
    fa_output_detached = True      # forward() detaches the lifted features: see fusion_common.lift_2d_features

    def __init__(self, config, lbl_values, ign_lbls):
        super(KPFCNN_featureAggre, self).__init__()
        self.K = config.num_kernel_points
        self.C = len(lbl_values) - len(ign_lbls)
        (self.encoder_blocks_3d, self.encoder_skips, dims3, in3, out_dim, layer, r) = build_encoder(
            config, config.in_features_dim_3d)
        (self.encoder_blocks_2d, _, dims2, in2, _, _, _) = build_encoder(config, config.in_features_dim_2d)
        self.encoder_skip_dims = [a + b for a, b in zip(dims3, dims2)]      # concatenated skips (:83)
        self.decoder_blocks, self.decoder_concats, out_dim = build_decoder(
            config, in3 + in2, out_dim, layer, r, self.encoder_skip_dims)   # reference quirk (:144): the
        # first decoder block is an upsample (no weights), so the doubled in_dim only matters through
        # the skip arithmetic -- kept as in the reference for state-dict compatibility.
        self.head_mlp = UnaryBlock(out_dim, config.first_features_dim, False, 0)
        self.head_softmax = UnaryBlock(config.first_features_dim, self.C, False, 0)
        self._init_losses(config, lbl_values, ign_lbls)
        build_2d_branch(self, config)

    def forward(self, batch, config):
        feature_2d3d = lift_2d_features(self, batch)
        ones = torch.ones_like(batch.feat_aggre_points[:, :, :1].squeeze(0))     # (np, 1)
        x_2d = torch.cat((ones, feature_2d3d), dim=1).detach()                   # 65 (`.clone().detach()`)
        x_3d = batch.feature_3d.detach()                                         # e.g. 4
        skip_x = []
        for block_i, block_op in enumerate(self.encoder_blocks_3d):
            if block_i in self.encoder_skips:
                skip_x.append(x_3d)
            x_3d = block_op(x_3d, batch)
        index = 0
        for block_i, block_op in enumerate(self.encoder_blocks_2d):
            if block_i in self.encoder_skips:
                skip_x[index] = torch.cat([skip_x[index], x_2d], dim=1)
                index += 1
            x_2d = block_op(x_2d, batch)
        x = torch.mean(torch.stack([x_3d, x_2d]), 0)
        x = run_decoder(self, x, skip_x, batch)
        return self.head_softmax(self.head_mlp(x, batch), batch)
