"""KD criteria with the reference's class names and constructor arguments (losses/__init__.py:1-7 of the
reference), so `config.init_obj('kd_loss' | 'hint_loss' | 'supervised_loss', losses)` resolves unchanged.
Each forward is one fused HIP pass producing the loss and its gradient (include/kdcc.h, losses section)."""
import torch
from torch import nn

from .. import ops
from .._lib import KdccError
from ..lazy import LazyLogits, deferred


class _FusedLoss(torch.autograd.Function):
    """loss, d loss / d inputs computed together in forward; backward only scales by the upstream gradient."""

    @staticmethod
    def forward(ctx, kind, inputs, targets, arg, weight):
        want = inputs.requires_grad
        s, t = inputs.detach(), targets.detach()
        if kind == "kld":
            loss, grad = ops.kldiv(s, t, arg, want_grad=want)
        elif kind == "mse":
            loss, grad = ops.hint_mse(s, t, arg, want_grad=want)
        else:
            loss, grad = ops.weighted_hint_mse(s, t, weight, want_grad=want)
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        ctx.grad = None
        if grad is None:
            return None, None, None, None, None
        # the stored gradient is this Function's own buffer: scale it in place, and not at all when the upstream factor is 1
        # (loss = sum of hint losses), which the kernel finds out on the device
        if g.numel() == 1 and grad.dtype in (torch.float32, torch.bfloat16) and grad.data_ptr() % 16 == 0:
            return None, ops.scale_by_device_scalar_(grad, g), None, None, None
        return None, grad * g.to(grad.dtype), None, None, None


def _same_device_dtype(inputs, targets):
    if targets.dtype != inputs.dtype and not (targets.dtype in (torch.float32, torch.bfloat16)):
        targets = targets.to(inputs.dtype)
    return targets


class KLDivergenceLoss(nn.Module):
    """kl_div(log_softmax(inputs/T, 1), softmax(targets/T, 1), 'mean') * T^2 * C  (losses/KLDiv.py:15-23)."""

    def __init__(self, temperature=1):
        super().__init__()
        self.temperature = temperature

    def forward(self, inputs, targets):
        if isinstance(inputs, LazyLogits) and isinstance(targets, LazyLogits) and inputs.pending and targets.pending and \
                inputs.size_hw == targets.size_hw and inputs.align_corners == targets.align_corners:
            # both sides are the classifier's half-resolution logits: interpolate in registers (kd_kldiv_up)
            try:
                return deferred(ops.kldiv_up(inputs.low, targets.low, inputs.size_hw, float(self.temperature), inputs.align_corners),
                                "kld", inputs, targets, float(self.temperature))
            except KdccError:
                pass      # resampling ratio / class count outside the kernel's range: materialise
        return _FusedLoss.apply("kld", inputs, _same_device_dtype(inputs, targets), float(self.temperature), None)


class MSELoss(nn.Module):
    """nn.MSELoss(reduction)(inputs, targets) * num_classes  (losses/MSELoss.py:9-16); reduction 'mean' (every shipped config) or 'sum'
    (the same kernel: the mean and its gradient scaled by the element count)."""

    def __init__(self, reduction='mean', num_classes=19):
        super().__init__()
        if reduction not in ('mean', 'sum'):
            raise NotImplementedError("reduction='none' returns a tensor no KD trainer consumes: 'mean' or 'sum'")
        self.reduction = reduction
        self.num_classes = num_classes

    def forward(self, inputs, targets):
        scale = float(self.num_classes) * (inputs.numel() if self.reduction == 'sum' else 1)
        return _FusedLoss.apply("mse", inputs, _same_device_dtype(inputs, targets), scale, None)


class WeightedHintMSELoss(nn.Module):
    """mean_n( sum_c w * mean_hw (s-t)^2 / sum_c w )  (losses/WeightedHintMSELoss.py:5-16); w is (C,) or (N,C)."""

    def __init__(self, reduction='mean', num_classes=19):
        super().__init__()
        self.reduction = reduction
        self.num_classes = num_classes

    def forward(self, inputs, targets, filter_weight):
        return _FusedLoss.apply("whmse", inputs, _same_device_dtype(inputs, targets), None, filter_weight)


class _CEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, ignore_index, weight=None, size_average=True):
        ctx.save_for_backward(inputs.detach(), targets)
        ctx.ignore_index, ctx.weight, ctx.size_average = ignore_index, weight, size_average
        return ops.ce2d(inputs.detach(), targets, ignore_index, weight, size_average)

    @staticmethod
    def backward(ctx, g):
        x, tgt = ctx.saved_tensors
        grad = ops.ce2d_grad(x, tgt, ctx.ignore_index, 1.0, ctx.weight, ctx.size_average)
        return grad * g.to(grad.dtype), None, None, None, None


class CrossEntropyLoss2d(nn.Module):
    """NLLLoss(ignore_index)(log_softmax(inputs, 1), targets)  (losses/CrossEntropy.py:5-14).  A logged metric in the KD
    trainers (forward-only there: `inputs` is detached when nothing asks for its gradient); differentiable for the trainers
    that back-propagate the supervised loss (trainer/taylor_prune_trainer.py:204-206)."""

    def __init__(self, weight=None, size_average=True, ignore_index=255):
        super().__init__()
        # a buffer like nn.NLLLoss(weight)'s: follows .to(device) / state_dict and is uploaded once
        self.register_buffer('weight', None if weight is None else torch.as_tensor(weight, dtype=torch.float32))
        self.size_average = bool(size_average)
        self.ignore_index = ignore_index

    def forward(self, inputs, targets):
        if self.weight is not None and self.weight.device != inputs.device:
            self.weight = self.weight.to(inputs.device)             # once (a criterion that was never moved with .to())
        if self.weight is not None or not self.size_average:       # (kd_ce2d_weighted; no shipped KD config: full-resolution logits)
            if inputs.requires_grad and torch.is_grad_enabled():
                return _CEFunction.apply(inputs, targets, self.ignore_index, self.weight, self.size_average)
            return ops.ce2d(inputs, targets, self.ignore_index, self.weight, self.size_average)
        if isinstance(inputs, LazyLogits) and inputs.pending:
            try:
                return deferred(ops.ce2d_up(inputs.low, targets, inputs.size_hw, self.ignore_index, inputs.align_corners),
                                "ce", inputs, targets, self.ignore_index)
            except KdccError:
                pass
        if inputs.requires_grad and torch.is_grad_enabled():
            return _CEFunction.apply(inputs, targets, self.ignore_index)
        return ops.ce2d(inputs, targets, self.ignore_index)
