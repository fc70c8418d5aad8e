"""2D encoder of MV-KPConv: UNet on a ResNet34 trunk (reference mvpnet/models/unet_resnet34.py:9-125).
Stays a PyTorch-ROCm (MIOpen) network per the scope contract; torchvision is not available on the
target image, so the ResNet34 trunk is built here with torchvision's parameter names
(encoderN.M.conv1/bn1/conv2/bn2/downsample.{0,1}) to stay checkpoint compatible."""
import torch
import torch.nn as nn
import torch.nn.functional as F


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

    def forward(self, data_dict):
        x = data_dict['image']
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
