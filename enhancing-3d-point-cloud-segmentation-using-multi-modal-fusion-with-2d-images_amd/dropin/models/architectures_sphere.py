"""MV-KPConv early fusion (reference KPConv-PyTorch/models/architectures_sphere.py:61-370):
lifted 2D features are concatenated with the 3D input features and fed to one KPFCNN."""
import torch
import torch.nn as nn

try:
    from .architectures import (p2p_fitting_regularizer, build_encoder, build_decoder, run_encoder_decoder,
                                _SegmentationLossMixin)
    from .blocks import UnaryBlock
    from .fusion_common import build_2d_branch, lift_2d_features
except ImportError:
    from models.architectures import (p2p_fitting_regularizer, build_encoder, build_decoder,
                                      run_encoder_decoder, _SegmentationLossMixin)
    from models.blocks import UnaryBlock
    from models.fusion_common import build_2d_branch, lift_2d_features


class KPFCNN_featureAggre(_SegmentationLossMixin, nn.Module):
    fa_output_detached = True      # forward() detaches the lifted features (:295): see fusion_common.lift_2d_features

    def __init__(self, config, lbl_values, ign_lbls):
        super(KPFCNN_featureAggre, self).__init__()
        self.K = config.num_kernel_points
        self.C = len(lbl_values) - len(ign_lbls)
        (self.encoder_blocks, self.encoder_skips, self.encoder_skip_dims,
         in_dim, out_dim, layer, r) = build_encoder(config, config.in_features_dim)
        self.decoder_blocks, self.decoder_concats, out_dim = build_decoder(
            config, in_dim, out_dim, layer, r, self.encoder_skip_dims)
        self.head_mlp = UnaryBlock(out_dim, config.first_features_dim, False, 0)
        self.head_softmax = UnaryBlock(config.first_features_dim, self.C, False, 0)
        self._init_losses(config, lbl_values, ign_lbls)
        build_2d_branch(self, config)

    def forward(self, batch, config):
        stacked = getattr(batch, 'stacked_features', None)                  # built ahead with the lifted features (bench.py)
        if stacked is None:
            feature_2d3d = lift_2d_features(self, batch)                   # (np, 64)
            stacked = torch.cat((batch.feature_3d, feature_2d3d), dim=1)   # e.g. 1 + z + 64 = 66 (:290-291)
        x = stacked.detach()               # :295 (`.clone().detach()`: no grad reaches the 2D branch; nothing below
                                           # writes x in place, so the copy itself is not needed)
        x = run_encoder_decoder(self, x, batch)
        return self.head_softmax(self.head_mlp(x, batch), batch)
