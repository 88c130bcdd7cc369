"""DeepLabV3+ with the WideResNet-38 trunk (`DeepWV3Plus`), mirroring the reference's
models/deeplabv3/deeplabv3.py:21-162 module tree (names and shapes) so that the teacher runs in
PyTorch-ROCm and the student's parameters keep the reference's checkpoint keys."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .wider_resnet import build_trunk


def _upsample(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def _conv_bn_relu(cin, cout, k, dilation=1):
    pad = dilation if k == 3 else 0
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=k, dilation=dilation, padding=pad, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class _AtrousSpatialPyramidPoolingModule(nn.Module):
    """image pooling + 1x1 + three dilated 3x3 branches, concatenated (image branch first)."""

    def __init__(self, in_dim, reduction_dim=256, output_stride=16, rates=(6, 12, 18)):
        super().__init__()
        if output_stride == 8:
            rates = [2 * r for r in rates]
        elif output_stride != 16:
            raise ValueError(f"output stride of {output_stride} not supported")
        self.features = nn.ModuleList([_conv_bn_relu(in_dim, reduction_dim, 1)] +
                                      [_conv_bn_relu(in_dim, reduction_dim, 3, r) for r in rates])
        self.img_pooling = nn.AdaptiveAvgPool2d(1)
        self.img_conv = _conv_bn_relu(in_dim, reduction_dim, 1)

    def forward(self, x):
        img = _upsample(self.img_conv(self.img_pooling(x)), x.shape[2:])
        return torch.cat([img] + [f(x) for f in self.features], 1)


class DeepWV3Plus(nn.Module):
    def __init__(self, num_classes, trunk="WideResnet38", criterion=None):
        super().__init__()
        if criterion is not None:
            raise NotImplementedError("supervised criterion / ImageNet trunk loading is outside the KD hot path")
        for name, mod in build_trunk().items():
            self.add_module(name, mod)
        self.aspp = _AtrousSpatialPyramidPoolingModule(4096, 256, output_stride=8)
        self.bot_fine = nn.Conv2d(128, 48, kernel_size=1, bias=False)
        self.bot_aspp = nn.Conv2d(1280, 256, kernel_size=1, bias=False)
        self.final = nn.Sequential(
            nn.Conv2d(256 + 48, 256, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
            nn.Conv2d(256, num_classes, kernel_size=1, bias=False))
        for m in self.final.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, inp, gts=None):
        size = inp.shape[2:]
        x = self.mod1(inp)
        m2 = self.mod2(self.pool2(x))
        x = self.mod3(self.pool3(m2))
        x = self.mod7(self.mod6(self.mod5(self.mod4(x))))
        dec0_up = self.bot_aspp(self.aspp(x))
        dec0 = torch.cat([self.bot_fine(m2), _upsample(dec0_up, m2.shape[2:])], 1)
        return _upsample(self.final(dec0), size)
