"""Model zoo entry points resolved by name from configs (`config['teacher']['type']`), mirroring the reference's
models/__init__.py for the classes on the KD hot path."""
import logging
from functools import reduce

import torch
from torch import nn

from . import cifar_models, metric  # noqa: F401
from .deeplabv3 import DeepWV3Plus  # noqa: F401
from .gscnn import GSCNN  # noqa: F401


def forgiving_state_restore(net, loaded_dict):
    """Partial load: keep only entries whose name and shape match (reference models/__init__.py:61-88), with
    auto-detection of DataParallel ('module.'-prefixed) checkpoints."""
    keys = list(loaded_dict.keys())
    is_parallel = len(keys) > 0 and all(k.startswith('module.') for k in keys)
    if is_parallel:
        loaded_dict = {k[len('module.'):]: v for k, v in loaded_dict.items()}
    own = net.state_dict()
    for k in own:
        if k in loaded_dict and own[k].size() == loaded_dict[k].size():
            own[k] = loaded_dict[k]
        else:
            logging.info("Skipped loading parameter %s", k)
    net.load_state_dict(own)
    return net


def restore_snapshot(net, optimizer, snapshot, restore_optimizer_bool):
    checkpoint = torch.load(snapshot, map_location=torch.device('cpu'), weights_only=False)
    if optimizer is not None and 'optimizer' in checkpoint and restore_optimizer_bool:
        optimizer.load_state_dict(checkpoint['optimizer'])
    net = forgiving_state_restore(net, checkpoint['state_dict'] if 'state_dict' in checkpoint else checkpoint)
    return net, optimizer


def load_weights(snapshot_file, net, optimizer=None, restore_optimizer_bool=False):
    return restore_snapshot(net, optimizer, snapshot_file, restore_optimizer_bool)
