"""2D encoder of MV-KPConv: UNet on a ResNet34 trunk (reference mvpnet/models/unet_resnet34.py:9-125).
Stays a PyTorch-ROCm (MIOpen) network per the scope contract; torchvision is not available on the
target image, so the ResNet34 trunk is built here with torchvision's parameter names
(encoderN.M.conv1/bn1/conv2/bn2/downsample.{0,1}) to stay checkpoint compatible."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

try:
    from ..._native import ops
except ImportError:
    from _native import ops

_FROZEN_FAST = os.environ.get("MVK_FROZEN_ENCODER_FAST", "1") == "1"


def _fold(conv_w, bn, conv_b=None, transposed=False):
    """Eval-mode BatchNorm folded into the preceding convolution: w' = w * g / sqrt(var + eps) per output channel,
    b' = beta + (b - mean) * g / sqrt(var + eps)."""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    shape = (1, -1, 1, 1) if transposed else (-1, 1, 1, 1)
    w = (conv_w.detach() * scale.view(shape)).contiguous(memory_format=torch.channels_last)
    b0 = conv_b.detach() if conv_b is not None else torch.zeros_like(scale)
    return w, (bn.bias.detach() + (b0 - bn.running_mean.detach()) * scale).contiguous()


class BasicBlock(nn.Module):
    def __init__(self, c_in, c_out, stride=1):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Conv2d(c_in, c_out, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c_out)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(c_out, c_out, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c_out)
        self.downsample = None
        if stride != 1 or c_in != c_out:
            self.downsample = nn.Sequential(nn.Conv2d(c_in, c_out, 1, stride, bias=False), nn.BatchNorm2d(c_out))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


def _stage(c_in, c_out, n, stride):
    return nn.Sequential(*[BasicBlock(c_in if i == 0 else c_out, c_out, stride if i == 0 else 1) for i in range(n)])


class UNetResNet34(nn.Module):
    def __init__(self, num_classes, p=0.0, pretrained=True):
        super(UNetResNet34, self).__init__()
        self.num_classes = num_classes
        # encoder (stride-1 stem, :19-28); `pretrained` weights cannot be fetched offline -> random init
        self.encoder0 = nn.Conv2d(3, 64, kernel_size=7, stride=1, padding=3, bias=False)
        self.bn = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.encoder1 = _stage(64, 64, 3, 1)
        self.encoder2 = _stage(64, 128, 4, 2)
        self.encoder3 = _stage(128, 256, 6, 2)
        self.encoder4 = _stage(256, 512, 3, 2)
        # decoder (:34-41)
        self.deconv4 = self.get_deconv(512, 256)
        self.decoder3 = self.get_conv(512, 256)
        self.deconv3 = self.get_deconv(256, 128)
        self.decoder2 = self.get_conv(256, 128)
        self.deconv2 = self.get_deconv(128, 64)
        self.decoder1 = self.get_conv(128, 64)
        self.deconv1 = self.get_deconv(64, 64)
        self.decoder0 = self.get_conv(128, 64)
        self.logit = nn.Conv2d(64, num_classes, 1, bias=True)
        self.dropout = nn.Dropout(p=p) if p > 0.0 else None

    @staticmethod
    def get_deconv(c_in, c_out):
        return nn.Sequential(nn.ConvTranspose2d(c_in, c_out, kernel_size=2, stride=2), nn.BatchNorm2d(c_out),
                             nn.ReLU(inplace=True))

    @staticmethod
    def get_conv(c_in, c_out):
        return nn.Sequential(nn.Conv2d(c_in, c_out, kernel_size=3, padding=1), nn.BatchNorm2d(c_out),
                             nn.ReLU(inplace=True))

    # ---- frozen fast path --------------------------------------------------------------------------------
    # In every MV-KPConv variant this network is frozen and in eval mode (architectures_sphere.py:232-237): a pure
    # function of the images. Its convolutions stay MIOpen calls; the BatchNorms are folded into their weights and
    # what is left of conv -> BN -> [+ identity] -> ReLU is one HIP launch (ops.bias_act_nhwc) on channels-last
    # tensors (no layout transposes around the implicit-GEMM kernels): ~90 launches instead of ~220 per call.
    # 'seg_logit' is not computed on this path (no caller of the fusion networks reads it).
    def _frozen_ok(self, x):
        if not (_FROZEN_FAST and x.is_cuda and x.dtype == torch.float32):
            return False
        if any(m.training for m in self.modules() if isinstance(m, (nn.BatchNorm2d, nn.Dropout))):
            return False
        return (not torch.is_grad_enabled()) or not (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def _folded(self):
        ver = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
        cache = getattr(self, "_fold_cache", None)
        if cache is not None and cache[0] == ver:
            return cache[1]
        f = {"stem": _fold(self.encoder0.weight, self.bn)}
        for si in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(self, "encoder%d" % si)):
                f[(si, bi, 1)] = _fold(blk.conv1.weight, blk.bn1)
                f[(si, bi, 2)] = _fold(blk.conv2.weight, blk.bn2)
                if blk.downsample is not None:
                    f[(si, bi, 0)] = _fold(blk.downsample[0].weight, blk.downsample[1])
        for name in ("deconv4", "deconv3", "deconv2", "deconv1"):
            m = getattr(self, name)
            f[name] = _fold(m[0].weight, m[1], m[0].bias, transposed=True)
        for name in ("decoder3", "decoder2", "decoder1", "decoder0"):
            m = getattr(self, name)
            f[name] = _fold(m[0].weight, m[1], m[0].bias)
        self._fold_cache = (ver, f)
        return f

    def _frozen_features(self, x):
        f = self._folded()
        h, w = x.shape[2], x.shape[3]
        pad_h, pad_w = (-h) % 16, (-w) % 16
        if pad_h or pad_w:
            x = F.pad(x, [0, pad_w, 0, pad_h])
        x = x.contiguous(memory_format=torch.channels_last)
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)      # a no-op when the library kept the layout
        wt, b = f["stem"]
        x = ops.bias_act_nhwc(cl(F.conv2d(x, wt, None, 1, 3)), b)
        feats = [x]
        x = self.maxpool(x)
        for si in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(self, "encoder%d" % si)):
                stride = blk.conv1.stride
                w1, b1 = f[(si, bi, 1)]
                w2, b2 = f[(si, bi, 2)]
                out = ops.bias_act_nhwc(cl(F.conv2d(x, w1, None, stride, 1)), b1)
                out = cl(F.conv2d(out, w2, None, 1, 1))
                if blk.downsample is not None:
                    wd, bd = f[(si, bi, 0)]
                    x = ops.bias_act_nhwc(out, b2, res=cl(F.conv2d(x, wd, None, stride, 0)), bias2=bd)
                else:
                    x = ops.bias_act_nhwc(out, b2, res=x)
            if si < 4:
                feats.append(x)
        for dec, de, skip in (("decoder3", "deconv4", 3), ("decoder2", "deconv3", 2), ("decoder1", "deconv2", 1),
                              ("decoder0", "deconv1", 0)):
            wt, b = f[de]
            up = ops.bias_act_nhwc(cl(F.conv_transpose2d(x, wt, None, 2)), b)
            wt, b = f[dec]
            x = ops.bias_act_nhwc(cl(F.conv2d(cl(torch.cat([up, feats[skip]], dim=1)), wt, None, 1, 1)), b)
        if pad_h or pad_w:
            x = x[:, :, 0:h, 0:w]
        return x

    def forward(self, data_dict):
        x = data_dict['image']
        if self._frozen_ok(x):
            with torch.no_grad():
                return {'seg_logit': None, 'feature': self._frozen_features(x)}
        return self._forward_modules(x)

    def _forward_modules(self, x):
        """The module-by-module forward of the reference (unet_resnet34.py:88-125): training / fine-tuning, CPU."""
        h, w = x.shape[2], x.shape[3]
        pad_h, pad_w = (-h) % 16, (-w) % 16
        if pad_h or pad_w:
            x = F.pad(x, [0, pad_w, 0, pad_h])
        feats = []
        x = self.relu(self.bn(self.encoder0(x)))
        feats.append(x)
        x = self.encoder1(self.maxpool(x))
        feats.append(x)
        x = self.encoder2(x)
        feats.append(x)
        x = self.encoder3(x)
        if self.dropout is not None:
            x = self.dropout(x)
        feats.append(x)
        x = self.encoder4(x)
        if self.dropout is not None:
            x = self.dropout(x)
        x = self.decoder3(torch.cat([self.deconv4(x), feats[3]], dim=1))
        x = self.decoder2(torch.cat([self.deconv3(x), feats[2]], dim=1))
        x = self.decoder1(torch.cat([self.deconv2(x), feats[1]], dim=1))
        x = self.decoder0(torch.cat([self.deconv1(x), feats[0]], dim=1))
        if pad_h or pad_w:
            x = x[:, :, 0:h, 0:w]
        return {'seg_logit': self.logit(x), 'feature': x}
