"""Classification metrics resolved by name from configs (`config['metrics']`, reference models/metric.py:33-50).
Returned as 0-dim device tensors: the trainers' MetricTracker sums them on the device (no per-step host sync)."""
import torch


def accuracy(output, target):
    with torch.no_grad():
        pred = torch.argmax(output, dim=1)
        assert pred.shape[0] == len(target)
        return (pred == target).sum().float() / len(target)


def top_k_acc(output, target, k=3):
    with torch.no_grad():
        pred = torch.topk(output, k, dim=1)[1]
        assert pred.shape[0] == len(target)
        return (pred == target.unsqueeze(1)).any(dim=1).sum().float() / len(target)
