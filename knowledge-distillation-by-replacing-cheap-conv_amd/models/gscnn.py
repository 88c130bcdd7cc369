"""Gated-SCNN (`GSCNN`, BASELINE config 5) as a parameter container with the reference's module tree and checkpoint keys
(models/gscnn/gscnn.py:183-325, models/gscnn/gate_spatial_conv.py:17-65, models/encoders/Resnet.py:64-99): the WideResNet-38
trunk of DeepWV3Plus plus a full-resolution shape stream (three BasicBlocks, three gated convs, a Canny prior) whose edge
attention feeds an extra ASPP branch.

Inside DepthwiseStudent both teacher and student run through engine.StudentEngine (HIP kernels; the Canny map is computed
on the device by kd_canny instead of the reference's per-step host round trip through cv2).  `forward` below is the plain
torch restatement used by `teacher_backend="torch"` and by host-side tests; it takes the edge map from `canny_fn`."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .deeplabv3 import _AtrousSpatialPyramidPoolingModule, _conv_bn_relu, _upsample
from .wider_resnet import build_trunk


class BasicBlock(nn.Module):
    """conv3x3-BN-ReLU-conv3x3-BN, identity shortcut, ReLU (stride 1, no downsample: the only form GSCNN uses)."""

    def __init__(self, planes):
        super().__init__()
        self.conv1 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        return self.relu(self.bn2(self.conv2(out)) + x)


class GatedSpatialConv2d(nn.Module):
    """out = W (feat * (alpha + 1)), alpha = sigmoid(BN(conv1x1(relu(conv1x1(BN([feat; gate]))))))."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 1, 1))
        nn.init.xavier_normal_(self.weight)
        c = in_channels + 1
        self._gate_conv = nn.Sequential(nn.BatchNorm2d(c), nn.Conv2d(c, c, 1), nn.ReLU(), nn.Conv2d(c, 1, 1), nn.BatchNorm2d(1),
                                        nn.Sigmoid())

    def forward(self, input_features, gating_features):
        alphas = self._gate_conv(torch.cat([input_features, gating_features], dim=1))
        return F.conv2d(input_features * (alphas + 1), self.weight)


class _EdgeASPP(_AtrousSpatialPyramidPoolingModule):
    """ASPP with the edge branch: cat[image pooling, edge, 1x1, rates] (gscnn.py:112-181)."""

    def __init__(self, in_dim, reduction_dim=256, output_stride=16, rates=(6, 12, 18)):
        super().__init__(in_dim, reduction_dim, output_stride, rates)
        self.edge_conv = _conv_bn_relu(1, reduction_dim, 1)

    def forward(self, x, edge):
        size = x.shape[2:]
        img = _upsample(self.img_conv(self.img_pooling(x)), size)
        e = self.edge_conv(_upsample(edge, size))
        return torch.cat([img, e] + [f(x) for f in self.features], 1)


class GSCNN(nn.Module):
    def __init__(self, num_classes, trunk=None, criterion=None):
        super().__init__()
        if criterion is not None:
            raise NotImplementedError("supervised criterion inside the model is outside the KD hot path")
        self.num_classes = num_classes
        for name, mod in build_trunk().items():
            self.add_module(name, mod)
        self.dsn1 = nn.Conv2d(64, 1, 1)
        self.dsn3 = nn.Conv2d(256, 1, 1)
        self.dsn4 = nn.Conv2d(512, 1, 1)
        self.dsn7 = nn.Conv2d(4096, 1, 1)
        self.res1 = BasicBlock(64)
        self.d1 = nn.Conv2d(64, 32, 1)
        self.res2 = BasicBlock(32)
        self.d2 = nn.Conv2d(32, 16, 1)
        self.res3 = BasicBlock(16)
        self.d3 = nn.Conv2d(16, 8, 1)
        self.fuse = nn.Conv2d(8, 1, kernel_size=1, padding=0, bias=False)
        self.cw = nn.Conv2d(2, 1, kernel_size=1, padding=0, bias=False)
        self.gate1 = GatedSpatialConv2d(32, 32)
        self.gate2 = GatedSpatialConv2d(16, 16)
        self.gate3 = GatedSpatialConv2d(8, 8)
        self.aspp = _EdgeASPP(4096, 256, output_stride=8)
        self.bot_fine = nn.Conv2d(128, 48, kernel_size=1, bias=False)
        self.bot_aspp = nn.Conv2d(1280 + 256, 256, kernel_size=1, bias=False)
        self.final_seg = nn.Sequential(
            nn.Conv2d(256 + 48, 256, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
            nn.Conv2d(256, num_classes, kernel_size=1, bias=False))
        self.canny_fn = None   # (N,3,H,W) batch -> (N,1,H,W) 0/255 map; default: the device kernel

    def _canny(self, inp):
        if self.canny_fn is not None:
            return self.canny_fn(inp)
        if not inp.is_cuda:
            raise RuntimeError("GSCNN.forward on host tensors needs `canny_fn` (the edge prior is a device kernel here; the "
                               "reference calls cv2.Canny on the host, models/gscnn/gscnn.py:284-288)")
        from .. import ops
        return ops.canny(inp.detach().float().contiguous()).unsqueeze(1)

    def forward(self, inp, gts=None):
        size = inp.shape[2:]
        up = lambda t: _upsample(t, size)
        m1 = self.mod1(inp)
        m2 = self.mod2(self.pool2(m1))
        m3 = self.mod3(self.pool3(m2))
        m4 = self.mod4(m3)
        m7 = self.mod7(self.mod6(self.mod5(m4)))
        s3, s4, s7 = up(self.dsn3(m3)), up(self.dsn4(m4)), up(self.dsn7(m7))
        canny = self._canny(inp).to(m1.dtype)
        cs = self.gate1(self.d1(up(self.res1(up(m1)))), s3)
        cs = self.gate2(self.d2(up(self.res2(cs))), s4)
        cs = self.gate3(self.d3(up(self.res3(cs))), s7)
        edge_out = torch.sigmoid(up(self.fuse(cs)))
        acts = torch.sigmoid(self.cw(torch.cat((edge_out, canny), dim=1)))
        x = self.aspp(m7, acts)
        dec0_up = self.bot_aspp(x)
        dec0 = torch.cat([self.bot_fine(m2), _upsample(dec0_up, m2.shape[2:])], 1)
        return F.interpolate(self.final_seg(dec0), size=size, mode="bilinear", align_corners=False)
