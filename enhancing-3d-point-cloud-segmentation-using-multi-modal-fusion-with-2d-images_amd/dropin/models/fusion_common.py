"""2D -> 3D lifting shared by the three MV-KPConv fusion variants
(reference KPConv-PyTorch/models/architectures_sphere.py:242-284 and the identical blocks in
architectures_sphere_middle_fusion.py:229-266 / architectures_sphere_late_fusion.py:233-270)."""
import os

import numpy as np
import torch

try:
    from ..mvpnet.models.mvpnet_3d import FeatureAggregation
    from ..mvpnet.models.unet_resnet34 import UNetResNet34
    from ..mvpnet.ops.group_points import group_points
except ImportError:
    from mvpnet.models.mvpnet_3d import FeatureAggregation
    from mvpnet.models.unet_resnet34 import UNetResNet34
    from mvpnet.ops.group_points import group_points


def build_2d_branch(net, config):
    """FeatureAggregation(64) + frozen UNetResNet34 in eval mode (architectures_sphere.py:205-237).
    The 2D checkpoint config.path_2D is loaded like the reference does (:228-230) and a missing file fails
    as loudly as its torch.load; only an EMPTY path (synthetic benchmarks: no checkpoint can be
    downloaded) keeps the random initialisation."""
    net.feat_aggreg = FeatureAggregation(64)
    net.net_2d = UNetResNet34(20, p=0.5, pretrained=True)
    path = getattr(config, 'path_2D', '')
    if path:
        if not os.path.exists(path):
            raise FileNotFoundError("config.path_2D = %r does not exist (the frozen 2D encoder would silently stay "
                                    "randomly initialised)" % path)
        checkpoint = torch.load(path, map_location=torch.device("cpu"))
        net.net_2d.load_state_dict(checkpoint['model'])
    for _, params in net.net_2d.named_parameters():
        params.requires_grad = False
    for _, m in net.net_2d._modules.items():
        m.train(False)


def lift_2d_features(net, batch, fused=None):
    """images -> UNet features -> k nearest pixels of every point -> FeatureAggregation.
    Returns feature_2d3d (np, 64).

    fused (default on the GPU): FeatureAggregation.forward_fused -- a single HIP gather kernel over the stacked points of
    ALL spheres + MFMA linear layers, BatchNorm statistics over all of them like the reference; fused=False: the
    reference's op sequence (per-sphere group_points, concatenation, the module's tensor-op forward)."""
    ahead = getattr(batch, 'feature_2d3d', None)
    if ahead is not None and getattr(net, 'fa_output_detached', False):
        # the lifted features of this batch were computed ahead of the step (bench.py: beside the previous step, like the
        # frozen encoder's and the pyramid). Sound only where the network detaches them (early / middle fusion,
        # architectures_sphere.py:295): nothing trainable is upstream of that point, FeatureAggregation included.
        return ahead
    images = batch.images                                   # (b, nv, 3, h, w)
    b, nv, _, h, w = images.size()
    images = images.reshape([-1] + list(images.shape[2:]))
    feature_2d = getattr(batch, 'feature_2d', None)         # the frozen encoder may have been run ahead (bench.py)
    if feature_2d is None:
        feature_2d = net.net_2d({'image': images})['feature']  # (b*nv, c, h, w), no grad (frozen)

    def knn_of(i):
        knn = batch.knn_list[i]
        if isinstance(knn, np.ndarray):
            knn = torch.from_numpy(knn)
        return knn.long().to(feature_2d.device)

    if fused is None:
        fused = feature_2d.is_cuda
    if fused:
        # The (b*nv, c, h, w) feature map is ONE stack of views: the pixel indices of sphere i (flat over ITS nv views,
        # ScanNet_sphere_color.py:436-451) shift by i*nv*h*w and the per-sphere group_points loop of the reference
        # (:262-279) becomes one gather over the stacked points; BatchNorm then spans all sum(np)*k rows, as it does in
        # the reference, where the grouped tensors are concatenated before FeatureAggregation (:278-283).
        knn_all = getattr(batch, 'knn_stacked', None)        # built with the offsets by the input side when present
        if knn_all is None:
            per = []
            for i in range(b):
                knn = knn_of(i)
                per.append((knn[0] if knn.dim() == 3 else knn) + i * nv * h * w)
            knn_all = per[0] if b == 1 else torch.cat(per, dim=0)
        return net.feat_aggreg.forward_fused(feature_2d, batch.image_xyz.reshape(b * nv, h, w, 3), knn_all,
                                             batch.feat_aggre_points[0])

    feature_2d = feature_2d.reshape(b, nv, -1, h, w).transpose(1, 2).contiguous().reshape(b, -1, nv * h * w)
    image_xyz = batch.image_xyz.permute(0, 4, 1, 2, 3).reshape(b, 3, nv * h * w)
    feats, xyzs = [], []
    for i in range(b):
        knn = knn_of(i)
        if knn.dim() == 2:
            knn = knn.unsqueeze(0)                           # (1, s_np, k)
        feats.append(group_points(feature_2d[i:i + 1], knn))
        with torch.no_grad():
            xyzs.append(group_points(image_xyz[i:i + 1].contiguous(), knn))
    input_feature_2d = torch.cat(feats, dim=2)              # (1, c, np, k)
    input_image_xyz = torch.cat(xyzs, dim=2)                # (1, 3, np, k)
    points = batch.feat_aggre_points.transpose(1, 2)        # (1, 3, np)
    feature_2d3d = net.feat_aggreg(input_image_xyz, points, input_feature_2d)   # (1, 64, np)
    return feature_2d3d.permute(0, 2, 1).reshape(-1, 64)
