"""nn.Conv2d / nn.BatchNorm2d whose forward and backward run on the small-shape HIP kernels (include/kdcc.h,
kd_conv2d_direct_*, kd_bn2d_*) when the tensors live on the GPU.  Same constructors, parameters, buffers and state-dict
keys as the torch classes, so checkpoints, forward hooks (hint layers) and DepthwiseStudent's module surgery are untouched.
Used by the CIFAR plumbing config (models/cifar_models): with them a ClassificationTrainer step issues no MIOpen kernel.

There is no silent fallback: a device tensor either goes through the kernels or raises, and a HOST tensor raises too unless
host plumbing mode was switched on explicitly (`allow_host_tensors(True)`: BaseTrainer does it for `n_gpu: 0` configs -- the
reference's CPU plumbing case, BASELINE config 1 -- and the CPU-only host-logic tests do it themselves).  In that mode the
modules behave exactly like their torch base classes; it is never the measured or parity-tested path.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from ._lib import KdccError

_HOST_OK = False


def allow_host_tensors(on=True):
    """Host plumbing mode (n_gpu = 0): Conv2d / BatchNorm2d given host tensors run their torch base class.  Off by default."""
    global _HOST_OK
    _HOST_OK = bool(on)


def _host(x):
    if x.is_cuda:
        return False
    if not _HOST_OK:
        raise KdccError("kdcc nn_hip modules got a host tensor: the HIP kernels need device tensors and there is no silent CPU "
                        "fallback (host plumbing runs, `n_gpu: 0`, call nn_hip.allow_host_tensors(True) -- BaseTrainer does)")
    return True


class _DirectConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil, groups):
        x, w = x.contiguous(), w.contiguous()
        ctx.save_for_backward(x, w)
        ctx.geom = (stride, pad, dil, groups, bias is not None)
        return ops.conv2d_direct(x, w, None if bias is None else bias.contiguous(), stride, pad, dil, groups)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad, dil, groups, has_bias = ctx.geom
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.conv2d_direct_dgrad(gy, w, x.shape, stride, pad, dil, groups)
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gw, gb = ops.conv2d_direct_wgrad(x, gy, w.shape, stride, pad, dil, groups, want_bias=has_bias)
        return gx, gw, gb, None, None, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """F.conv2d on the direct HIP kernels (device fp32 NCHW) / torch (host tensors)."""
    if _host(x):
        return F.conv2d(x, weight, bias, stride, padding, dilation, groups)
    if x.dtype != torch.float32:
        raise TypeError("the small-shape conv kernels are fp32 (the CIFAR path of the reference is fp32)")
    one = lambda v: v[0] if isinstance(v, (tuple, list)) else v
    return _DirectConv.apply(x, weight, bias, one(stride), one(padding), one(dilation), groups)


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
        x = x.contiguous()
        y, mean, invstd = ops.bn2d_fwd(x, gamma.contiguous(), beta.contiguous(), running_mean, running_var, training, momentum, eps, relu)
        ctx.save_for_backward(x, y if relu else x, gamma, mean, invstd)
        ctx.flags = (training, relu)
        ctx.mark_non_differentiable(mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        training, relu = ctx.flags
        dx, dg, db = ops.bn2d_bwd(gy.contiguous(), x, y, gamma.contiguous(), mean, invstd, training, relu,
                                  need_dx=ctx.needs_input_grad[0])
        return dx, dg if ctx.needs_input_grad[1] else None, db if ctx.needs_input_grad[2] else None, None, None, None, None, None, None


class Conv2d(nn.Conv2d):
    def forward(self, x):
        if _host(x):
            return super().forward(x)
        if self.padding_mode != "zeros" or isinstance(self.padding, str) or len({*self.stride}) != 1 or len({*self.padding}) != 1 \
                or len({*self.dilation}) != 1:
            raise NotImplementedError("HIP Conv2d: square zero-padded geometry only")
        return conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class BatchNorm2d(nn.BatchNorm2d):
    def forward(self, x, relu=False):
        if _host(x):
            y = super().forward(x)
            return F.relu(y) if relu else y
        if not (self.affine and self.track_running_stats) or self.momentum is None:
            raise NotImplementedError("HIP BatchNorm2d: affine, running statistics, fixed momentum")
        if x.dtype != torch.float32:
            raise TypeError("HIP BatchNorm2d is fp32")
        if self.training:
            self.num_batches_tracked.add_(1)
        return _BatchNorm.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, float(self.momentum),
                                float(self.eps), bool(relu))
