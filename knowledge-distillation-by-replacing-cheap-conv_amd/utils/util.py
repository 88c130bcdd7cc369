"""Host-side bookkeeping used by the trainers (reference utils/util.py), re-designed so that the hot loop never
forces a device->host transfer: metric values may be 0-dim device tensors and are summed on the device; the mIoU
confusion matrix is built on the device (the reference copies both full logit tensors to the host every step,
trainer/layerwise_trainer.py:249-250 -> utils/util.py:108-128)."""
import json
from collections import OrderedDict
from itertools import repeat
from pathlib import Path

import torch


def read_json(fname):
    with Path(fname).open('rt') as handle:
        return json.load(handle, object_hook=OrderedDict)


def write_json(content, fname):
    with Path(fname).open('wt') as handle:
        json.dump(content, handle, indent=4, sort_keys=False)


def ensure_dir(dirname):
    Path(dirname).mkdir(parents=True, exist_ok=True)


def inf_loop(data_loader):
    """Endless wrapper around a data loader."""
    for loader in repeat(data_loader):
        yield from loader


class MetricTracker:
    """Running averages keyed by name; values may be python numbers or 0-dim tensors (kept on their device until read)."""

    def __init__(self, *keys, writer=None):
        self.writer = writer
        self._keys = list(keys)
        self.reset()

    def reset(self):
        self._total = {k: 0.0 for k in self._keys}
        self._count = {k: 0 for k in self._keys}
        self._pending = []   # (key, writer mode, writer step, 0-dim device tensor) not yet sent to the writer

    def update(self, key, value, n=1):
        """The reference pushes every update to TensorBoard immediately (utils/util.py:60-63), which for a device scalar
        is a host sync per logged key per step.  Device scalars are buffered with the writer's current step instead and
        flush() sends them all with ONE stacked device->host copy (at log points / epoch end)."""
        if torch.is_tensor(value):
            value = value.detach().float()
            if self.writer is not None:
                self._pending.append((key, getattr(self.writer, 'mode', ''), getattr(self.writer, 'step', 0), value))
        elif self.writer is not None:
            self.writer.add_scalar(key, value)
        self._total[key] = self._total[key] + value * n
        self._count[key] += n

    def flush(self):
        if not self._pending:
            return
        pend, self._pending = self._pending, []
        vals = torch.stack([p[3].reshape(()) for p in pend]).cpu().tolist()
        keep = (getattr(self.writer, 'mode', ''), getattr(self.writer, 'step', 0))
        for (key, mode, step, _), v in zip(pend, vals):
            if hasattr(self.writer, 'mode'):
                self.writer.mode, self.writer.step = mode, step
            self.writer.add_scalar(key, v)
        if hasattr(self.writer, 'mode'):
            self.writer.mode, self.writer.step = keep

    def avg(self, key):
        c = self._count[key]
        t = self._total[key]
        if torch.is_tensor(t):
            t = t.item()   # the only synchronisation point: when somebody asks for the number
        return t / c if c else 0.0

    def result(self):
        return {k: self.avg(k) for k in self._keys}


class CityscapesMetricTracker:
    """Confusion-matrix mIoU over the 19 train ids; pixels labelled ignore_index are dropped."""
    class_names = ["road", "sidewalk", "building", "wall", "fence", "pole", "traffic_light", "traffic_sight", "vegetation",
                   "terrain", "sky", "person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle"]
    num_classes = len(class_names)

    def __init__(self, writer=None, ignore_index=255):
        self.writer = writer
        self.ignore_index = ignore_index
        self.reset()

    def reset(self):
        self.conf = None

    def update(self, outputs, labels):
        """outputs (N,C,H,W) logits on any device, labels (N,H,W); stays on the outputs' device. Does NOT modify labels
        (the reference rewrites 255 -> 19 in the caller's tensor)."""
        outputs = outputs.detach()
        if outputs.is_cuda:
            # one fused HIP pass (argmax + int64 histogram, kd_confusion): exact, no temporaries, no host transfer
            from .. import ops
            if outputs.shape[1] != self.num_classes:
                raise ValueError(f"expected {self.num_classes}-class logits, got {outputs.shape[1]} channels")
            tgt = labels.to(device=outputs.device, dtype=torch.int64)
            if self.conf is None:
                self.conf = torch.zeros((self.num_classes, self.num_classes), dtype=torch.int64, device=outputs.device)
            ops.confusion(outputs, tgt, self.conf, accumulate=True)
            return
        # host tensors (n_gpu = 0 runs): the reference's own numpy recipe, in torch
        pred = torch.argmax(outputs, dim=1).reshape(-1)
        tgt = labels.reshape(-1)
        mask = (tgt >= 0) & (tgt < self.num_classes)
        idx = self.num_classes * tgt[mask].long() + pred[mask]
        hist = torch.bincount(idx, minlength=self.num_classes ** 2).reshape(self.num_classes, self.num_classes)
        self.conf = hist if self.conf is None else self.conf + hist

    def get_iou(self):
        if self.conf is None or not bool(self.conf.any()):
            return 1.
        conf = self.conf.double().cpu()
        tp = conf.diag()
        iou = tp / (conf.sum(0) + conf.sum(1) - tp)
        return float(iou[~torch.isnan(iou)].mean()) if bool((~torch.isnan(iou)).any()) else float('nan')


class EarlyStopTracker:
    """Tracks whether a monitored value still improves (mode 'last' | 'best', criterion 'min' | 'max')."""

    def __init__(self, mode='last', criterion='min', threshold=0.0001, threshold_mode='rel'):
        if mode not in ('last', 'best'):
            raise ValueError('Unsupported type of mode. Expect either "last" or "best" but got: ' + str(mode))
        if criterion not in ('min', 'max'):
            raise ValueError('Unsupported type of mode. Expect either "min" or "max" but got: ' + str(criterion))
        self.mode, self.criterion, self.threshold, self.threshold_mode = mode, criterion, threshold, threshold_mode
        self.reset()

    def reset(self):
        self.last = None
        self.best = None
        self.last_update_success = True

    def is_better(self, old, new):
        if old is None:
            return True
        sign = -1.0 if self.criterion == 'min' else 1.0
        bar = old * (1 + sign * self.threshold) if self.threshold_mode == 'rel' else old + sign * self.threshold
        return new < bar if self.criterion == 'min' else new > bar

    def update(self, new_value):
        old = self.best if self.mode == 'best' else self.last
        ok = self.is_better(old, new_value)
        if ok and self.mode == 'best':
            self.best = new_value
        self.last = new_value
        self.last_update_success = ok
        return ok


class ImportanceFilterTracker:
    """Running mean of per-filter Taylor importances, normalised to sum 1 per layer (reference utils/util.py:191-216).
    Values may be numpy vectors (the reference's type) or tensors; average() returns {name: float32 tensor}, the table
    LayerwiseTrainer reads as `hint_filter_weight` for WeightedHintMSELoss."""

    def __init__(self, writer=None):
        self.writer = writer
        self.importance_dict = dict()
        self.counter = dict()
        self.temperature = 1
        self.scale_factor = 1e5

    def update_importance_list(self, added_gates):
        for name, gate_layer in added_gates.items():
            self.importance_dict[name] = torch.zeros(gate_layer.num_features, dtype=torch.float64)
            self.counter[name] = 0.

    def update(self, new_importance_dict):
        for name, vector in new_importance_dict.items():
            self.importance_dict[name] += torch.as_tensor(vector, dtype=torch.float64).cpu()
            self.counter[name] += 1

    def average(self):
        result = dict()
        for name, vector in self.importance_dict.items():
            mean_vector = vector / self.counter[name] * self.scale_factor
            result[name] = (mean_vector / mean_vector.sum()).float()
        return result
