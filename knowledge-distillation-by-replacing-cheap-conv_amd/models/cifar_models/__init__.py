from .resnet import ResNet, resnet20, resnet32, resnet44, resnet56, resnet110  # noqa: F401
