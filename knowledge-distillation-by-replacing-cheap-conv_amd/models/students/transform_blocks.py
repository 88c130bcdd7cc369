"""Replacement block: depthwise k x k (dilated) conv followed by a 1x1 conv -- the "cheap conv" of
models/students/transform_blocks/depthwise_separable_conv.py:4-13 (same constructor, same child
names `separable_conv` / `pointwise_conv`, hence the same checkpoint keys).

As an nn.Module it is a parameter container plus a standalone HIP-backed forward: the NHWC depthwise + MFMA pointwise
kernels for GEMM-sized channel counts, the direct NCHW kernels (nn_hip.conv2d, groups = C then 1x1, bias supported) for the
CIFAR configs' 16/32/64-channel 3x3 blocks; inside a DeepWV3Plus student the engine fuses it into the surrounding graph.
"""
import torch
from torch import nn

from ... import ops


def _small_shape(x, pw):
    """The MFMA pointwise kernel needs the channel count to be a multiple of its K granule (32 in fp32); fp32 blocks outside that
    (the CIFAR configs' 16-channel blocks) and biased fp32 blocks take the direct NCHW kernels."""
    return x.dtype == torch.float32 and (x.shape[1] % 32 != 0 or pw.bias is not None)


class _DwSepFunction(torch.autograd.Function):
    """y = pw(dw(x)) on NCHW-logical tensors; internally NHWC, kernels from include/kdcc.h."""

    @staticmethod
    def forward(ctx, x, w_dw, w_pw, b_dw, b_pw, k, pad, dil):
        xh = x.permute(0, 2, 3, 1).contiguous()
        mid = ops.dwconv(xh, ops.pack_dw_weight(w_dw), k, pad, dil, bias=b_dw)
        wp = ops.pack_conv_weight(w_pw, xh.dtype)
        N, H, W, _ = xh.shape
        y = torch.empty((N, H, W, w_pw.shape[0]), dtype=xh.dtype, device=xh.device)
        if b_pw is not None:      # the conv epilogue's per-channel shift (scale 1, no ReLU) is the bias
            ops.conv2d(mid, wp, out_act=y, act_shift=b_pw.detach().float().contiguous(), act_relu=False)
        else:
            ops.conv2d(mid, wp, out_raw=y)
        ctx.save_for_backward(xh, mid, w_dw, w_pw)
        ctx.geom = (k, pad, dil, b_dw is not None, b_pw is not None)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        from ..._lib import KD_PACK_DGRAD
        xh, mid, w_dw, w_pw = ctx.saved_tensors
        k, pad, dil, has_bias, has_pw_bias = ctx.geom
        g = gy.permute(0, 2, 3, 1).contiguous()
        gw_pw = torch.empty_like(w_pw, dtype=torch.float32)
        ops.pw_wgrad(mid, g, gw_pw)
        gmid = torch.empty_like(mid)
        ops.conv2d(g, ops.pack_conv_weight(w_pw, g.dtype, KD_PACK_DGRAD), out_raw=gmid)
        gw_dw = torch.empty_like(w_dw, dtype=torch.float32)
        ops.dwconv_wgrad(xh, gmid, gw_dw, k, pad, dil)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = ops.dwconv(gmid, ops.pack_dw_weight(w_dw, flip=True), k, pad, dil).permute(0, 3, 1, 2)
        gb = gmid.float().sum(dim=(0, 1, 2)) if has_bias else None
        gb_pw = g.float().sum(dim=(0, 1, 2)) if has_pw_bias else None
        return gx, gw_dw, gw_pw, gb, gb_pw, None, None, None


class DepthwiseSeparableBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, padding, dilation, groups, bias, use_cuda=True):
        super().__init__()
        if groups != in_channels:
            raise ValueError("DepthwiseSeparableBlock expects groups == in_channels")
        has_bias = bias is not None and bias is not False
        self.separable_conv = nn.Conv2d(in_channels, in_channels, kernel_size, padding=padding, dilation=dilation,
                                        groups=groups, bias=has_bias)
        self.pointwise_conv = nn.Conv2d(in_channels, out_channels, 1, bias=has_bias)
        self.in_channels, self.out_channels = in_channels, out_channels

    @property
    def geometry(self):
        c = self.separable_conv
        return c.kernel_size[0], c.padding[0], c.dilation[0]

    def forward(self, x):
        k, pad, dil = self.geometry
        if x.is_cuda and _small_shape(x, self.pointwise_conv):
            from ...nn_hip import conv2d
            mid = conv2d(x, self.separable_conv.weight, self.separable_conv.bias, 1, pad, dil, self.in_channels)
            return conv2d(mid, self.pointwise_conv.weight, self.pointwise_conv.bias)
        return _DwSepFunction.apply(x, self.separable_conv.weight, self.pointwise_conv.weight, self.separable_conv.bias,
                                    self.pointwise_conv.bias, k, pad, dil)


class GateLayer(nn.Module):
    """Per-channel multiplier initialised to 1 (models/students/transform_blocks/gate.py:5-12): what TaylorPruneStudent puts
    behind a block to read its filters' Taylor importance, (gate * d loss / d gate)^2.  A trainable parameter, as in the
    reference -- its optimizer steps it.  Inside the fused student graph the engine folds it into the neighbouring kernel
    (a conv's packed weights or a BN+ReLU epilogue's scale / shift) and produces its gradient during backward."""

    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.weight = nn.Parameter(torch.ones(num_features))

    def forward(self, input):
        return input * self.weight.view(1, -1, 1, 1)
