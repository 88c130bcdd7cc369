"""Lazily materialised full-resolution logits.

`model(data) -> (output_st, output_tc)` (models/students/depthwise_student.py:168-177 of the reference) hands the trainer two
(N,19,H,W) fp32 tensors: the classifier's half-resolution output, bilinearly up-sampled (models/deeplabv3/deeplabv3.py:160-162).
In the KD trainers their only readers are the logged criteria -- CrossEntropyLoss2d x2, KLDivergenceLoss
(trainer/layerwise_trainer.py:222-227) -- and this library's criteria interpolate a pixel's logits in registers (kd_ce2d_up /
kd_kldiv_up): 2 x 1.27 GB per 8 images are neither written nor read back.  `LazyLogits` keeps the drop-in signature: it IS a
tensor of the full-resolution shape; the criteria recognise it and read `low`; ANY other use (a torch op, the mIoU kernel,
`.cpu()`, indexing ...) materialises the up-sampled tensor once through kd_upsample_bilinear and proceeds on it.  Only tensors
nothing differentiates through are wrapped (the engine returns a plain tensor when a logit loss is back-propagated)."""
import torch

from . import ops


class LazyLogits(torch.Tensor):
    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def __new__(cls, low, size, align_corners=True):
        N, h, w, C = low.shape
        H, W = size
        # NCHW-logical, channels-last strides: what the engine's materialised logits look like
        t = torch.Tensor._make_wrapper_subclass(cls, (N, C, H, W), strides=(H * W * C, 1, W * C, C), dtype=torch.float32,
                                                device=low.device, requires_grad=False)
        t.low, t.size_hw, t.align_corners, t._real = low, (int(H), int(W)), bool(align_corners), None
        t.anchor = None     # set by the engine: a differentiable 0-dim output of the student's autograd node (see deferred)
        t.pending_grad = None
        return t

    def materialize(self):
        """The real (N,C,H,W) channels-last tensor, computed once."""
        if self._real is None:
            full = ops.upsample_bilinear_ac(self.low, self.size_hw, out_dtype=torch.float32, align_corners=self.align_corners)
            self._real = full.permute(0, 3, 1, 2)
        return self._real

    @property
    def pending(self):
        """True while nothing has asked for the full-resolution tensor."""
        return self._real is None

    def __repr__(self):
        return f"LazyLogits(low={tuple(self.low.shape)}, size={self.size_hw}, materialised={self._real is not None})"

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def real(a):
            if isinstance(a, LazyLogits):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(x) for x in a)
            return a
        return func(*real(args), **{k: real(v) for k, v in (kwargs or {}).items()})


def unwrap(t):
    """A plain tensor for code that takes raw pointers (ops.*): materialises a LazyLogits."""
    return t.materialize() if isinstance(t, LazyLogits) else t


class _DeferredLogitLoss(torch.autograd.Function):
    """A loss value computed from LazyLogits is normally a logged number.  It stays differentiable all the same (the reference's
    logits always are): the value is tied to the student's autograd node through `anchor`, and IF something back-propagates it,
    the full-resolution logits are materialised here, the ordinary fused kernel produces d loss / d logits, and the student's
    backward picks that tensor up (`pending_grad`) as the gradient of its logits output.  Trainers that always back-propagate a
    logit loss say so up front (DepthwiseStudent.logits_need_grad) and never come this way."""

    @staticmethod
    def forward(ctx, value, anchor, kind, lazy_s, other, arg):
        ctx.kind, ctx.lazy_s, ctx.other, ctx.arg = kind, lazy_s, other, arg
        return value.detach().clone()

    @staticmethod
    def backward(ctx, g):
        s = ctx.lazy_s.materialize()
        if ctx.kind == "kld":
            t = ctx.other.materialize() if isinstance(ctx.other, LazyLogits) else ctx.other
            _, grad = ops.kldiv(s, t, ctx.arg, want_grad=True)
        else:
            grad = ops.ce2d_grad(s, ctx.other, ctx.arg)
        grad = grad * g.to(grad.dtype)
        ls = ctx.lazy_s
        ls.pending_grad = grad if ls.pending_grad is None else ls.pending_grad + grad
        ctx.lazy_s = ctx.other = None
        return None, torch.zeros_like(g), None, None, None, None


def deferred(value, kind, lazy_s, other, arg):
    """`value` (a 0-dim loss computed from the half-resolution logits of `lazy_s`) as the criteria return it."""
    a = getattr(lazy_s, "anchor", None)
    if a is not None and a.requires_grad and torch.is_grad_enabled():
        return _DeferredLogitLoss.apply(value, a, kind, lazy_s, other, arg)
    return value
