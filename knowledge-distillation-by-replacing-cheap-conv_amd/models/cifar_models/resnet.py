"""CIFAR ResNet-20/32/44/56/110 (He et al., option-A parameter-free shortcuts) with the reference's module names
(models/cifar_models/resnet.py:57-140: conv1, bn1, layer{1,2,3}.{i}.{conv1,bn1,conv2,bn2}, linear), so the shipped
`checkpoints/cifar10/resnet20.th` ('module.'-prefixed state dict) loads through forgiving_state_restore.
Used by BASELINE config 0 (KLDiv-only KD, ClassificationTrainer).  Convolutions and BatchNorm (training and eval mode,
forward and backward) are nn_hip.Conv2d / nn_hip.BatchNorm2d, i.e. the small-shape HIP kernels on GPU tensors -- no MIOpen
kernel runs; the parameter-free glue (residual add + ReLU, option-A shortcut, 8x8 average pool) and the 64x10 linear layer
are torch tensor ops."""
import torch.nn as nn
import torch.nn.functional as F

from ...nn_hip import BatchNorm2d, Conv2d


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = Conv2d(in_planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.pad_planes = planes // 4 if (stride != 1 or in_planes != planes) else 0
        self.shortcut = nn.Sequential()   # option A has no parameters; kept for state-dict/module-tree parity

    def forward(self, x):
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out))
        sc = x
        if self.pad_planes:
            sc = F.pad(x[:, :, ::2, ::2], (0, 0, 0, 0, self.pad_planes, self.pad_planes), "constant", 0)
        return F.relu(out + sc)


class ResNet(nn.Module):
    def __init__(self, num_blocks, num_classes=10):
        super().__init__()
        self.conv1 = Conv2d(3, 16, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = BatchNorm2d(16)
        in_planes = 16
        for i, (planes, stride) in enumerate(((16, 1), (32, 2), (64, 2))):
            blocks = []
            for s in [stride] + [1] * (num_blocks[i] - 1):
                blocks.append(BasicBlock(in_planes, planes, s))
                in_planes = planes
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.linear = nn.Linear(64, num_classes)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        out = self.bn1(self.conv1(x), relu=True)
        out = self.layer3(self.layer2(self.layer1(out)))
        out = F.avg_pool2d(out, out.size()[3]).flatten(1)
        return self.linear(out)


def resnet20():
    return ResNet([3, 3, 3])


def resnet32():
    return ResNet([5, 5, 5])


def resnet44():
    return ResNet([7, 7, 7])


def resnet56():
    return ResNet([9, 9, 9])


def resnet110():
    return ResNet([18, 18, 18])
