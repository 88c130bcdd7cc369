"""WideResNet-38 (A2 variant) trunk as a plain torch.nn parameter container / teacher graph.

Behavioural mirror of the reference's models/encoders/wider_resnet.py
(IdentityResidualBlock :119-182, WiderResNetA2 :267-378) with identical
sub-module names, so checkpoints and dotted block names such as
'mod4.block2.convs.conv2' resolve the same way.  Written table-driven; the
student never executes these modules (see engine.py) -- the frozen teacher does,
through PyTorch-ROCm.
"""
from collections import OrderedDict

import torch.nn as nn

# (mid..., out) channels per module mod2..mod7; two entries = two 3x3 convs, three = bottleneck
MODULE_CHANNELS = [(128, 128), (256, 256), (512, 512), (512, 1024), (512, 1024, 2048), (1024, 2048, 4096)]
WRN38_STRUCTURE = [3, 3, 6, 3, 1, 1]
DROPOUT_P = {4: 0.3, 5: 0.5}  # mod6 / mod7 (Dropout2d: the reference rebinds nn.Dropout globally, :301)


def bnrelu(channels):
    return nn.Sequential(nn.BatchNorm2d(channels), nn.ReLU(inplace=True))


class IdentityResidualBlock(nn.Module):
    """Pre-activation residual block: out = convs(bn1(x)) + (proj_conv(bn1(x)) | x)."""

    def __init__(self, in_channels, channels, stride=1, dilation=1, dropout_p=None):
        super().__init__()
        if len(channels) not in (2, 3):
            raise ValueError("channels must contain either two or three values")
        self.bn1 = bnrelu(in_channels)
        k = [3, 3] if len(channels) == 2 else [1, 3, 1]
        cin = [in_channels] + list(channels[:-1])
        layers = []
        for i, (ci, co, ks) in enumerate(zip(cin, channels, k)):
            if i > 0:
                layers.append((f"bn{i + 1}", bnrelu(ci)))
                if i == len(k) - 1 and dropout_p is not None:
                    layers.append(("dropout", nn.Dropout2d(p=dropout_p)))
            pad = dilation if ks == 3 else 0
            layers.append((f"conv{i + 1}", nn.Conv2d(ci, co, ks, stride=stride if i == 0 else 1, padding=pad,
                                                     dilation=dilation if ks == 3 else 1, bias=False)))
        self.convs = nn.Sequential(OrderedDict(layers))
        if stride != 1 or in_channels != channels[-1]:
            self.proj_conv = nn.Conv2d(in_channels, channels[-1], 1, stride=stride, padding=0, bias=False)

    def forward(self, x):
        if hasattr(self, "proj_conv"):
            a = self.bn1(x)
            shortcut = self.proj_conv(a)
        else:
            shortcut = x.clone()
            a = self.bn1(x)
        out = self.convs(a)
        out.add_(shortcut)  # in place: a forward hook on the last conv observes the block output
        return out


def build_trunk(structure=WRN38_STRUCTURE):
    """Returns an OrderedDict of the trunk's children: mod1, pool2, mod2, pool3, mod3 ... mod7 (output stride 8)."""
    mods = OrderedDict()
    mods["mod1"] = nn.Sequential(OrderedDict([("conv1", nn.Conv2d(3, 64, 3, stride=1, padding=1, bias=False))]))
    cin = 64
    for mod_id, num in enumerate(structure):
        blocks = []
        for block_id in range(num):
            dil = 2 if mod_id == 3 else (4 if mod_id > 3 else 1)
            stride = 2 if (block_id == 0 and mod_id == 2) else 1
            blocks.append((f"block{block_id + 1}",
                           IdentityResidualBlock(cin, MODULE_CHANNELS[mod_id], stride=stride, dilation=dil,
                                                 dropout_p=DROPOUT_P.get(mod_id))))
            cin = MODULE_CHANNELS[mod_id][-1]
        if mod_id < 2:
            mods[f"pool{mod_id + 2}"] = nn.MaxPool2d(3, stride=2, padding=1)
        mods[f"mod{mod_id + 2}"] = nn.Sequential(OrderedDict(blocks))
    return mods
