"""Tensor-level wrappers over the C-ABI (include/kdcc.h).

Activations are torch tensors of logical shape (N, H, W, C) whose last
dimension is dense and whose pixel stride `ld` may exceed C (a channel slice of
a wider NHWC buffer).  Every function validates shapes on the host before a
kernel is launched and enqueues on torch's current HIP stream.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (KD_BF16, KD_F32, KD_PACK_DGRAD, KD_PACK_FWD, ConvDesc, ConvEpilogue, DConvDesc, DwDesc, DwEpilogue, View3,
                   check)


# Optional live profiler (bench.py): a list that receives (family, algorithmic work, start event, end event, label, device
# kernel the dispatcher picked) for every launch of the conv / depthwise / weight-gradient / loss entry points, the events
# recorded on the stream the kernel is launched on.  Work is algorithmic FLOPs for the MFMA-bound families ("conv_igemm",
# "conv_wgrad", "pw_wgrad") and algorithmic HBM bytes -- every operand read once, every result written once -- for the
# HBM-bound ones ("depthwise", "loss").
PROFILER = None


def _prof_start():
    if PROFILER is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _prof_stop(e0, family, work, label, kernel=None):
    if e0 is None or PROFILER is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILER.append((family, float(work), e0, e1, label, kernel if kernel is not None else _lib.last_kernel()))


def _nbytes(*ts):
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


def dt_of(t):
    if t.dtype == torch.bfloat16:
        return KD_BF16
    if t.dtype == torch.float32:
        return KD_F32
    raise TypeError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    if type(t) is not torch.Tensor and hasattr(t, "materialize"):   # lazy.LazyLogits has no storage of its own
        t = t.materialize()
    return C.c_void_p(t.data_ptr())


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.KdccError("kdcc kernels need device tensors (there is no CPU fallback)")


def nhwc_ld(t):
    """Pixel stride of an (N,H,W,C) activation view; validates the layout."""
    if t.dim() != 4 or t.stride(3) != 1:
        raise ValueError(f"expected an (N,H,W,C) tensor with dense channels, got shape {tuple(t.shape)} stride {t.stride()}")
    N, H, W, _ = t.shape
    ld = t.stride(2)
    if (W > 1 and ld < t.shape[3]) or (H > 1 and t.stride(1) != W * ld) or (N > 1 and t.stride(0) != H * W * ld):
        raise ValueError(f"not a pixel-strided NHWC view: shape {tuple(t.shape)} stride {t.stride()}")
    return ld


def new_nhwc(N, H, W, Cc, dtype, device, zero=False):
    f = torch.zeros if zero else torch.empty
    return f((N, H, W, Cc), dtype=dtype, device=device)


def as_nchw(t):
    """(N,H,W,C) activation -> logical NCHW view (channels_last memory), no copy."""
    return t.permute(0, 3, 1, 2)


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


# ------------------------------------------------------------------------------ dense conv
def pack_conv_weight(w, dtype, mode=KD_PACK_FWD, cin_pad=None):
    """nn.Conv2d.weight (Cout,Cin,kh,kw) fp32 -> packed operand for conv2d()."""
    _need_cuda(w)
    w = w.detach().contiguous().float()
    Cout, Cin, kh, kw = w.shape
    cin_pad = Cin if cin_pad is None else cin_pad
    if mode == KD_PACK_FWD:
        out = torch.empty((Cout, kh, kw, cin_pad), dtype=dtype, device=w.device)
    else:
        out = torch.empty((Cin, kh, kw, Cout), dtype=dtype, device=w.device)
    check(_lib.lib().kd_pack_conv_weight(_ptr(w), _ptr(out), dt_of(out), mode, Cout, Cin, kh, kw, cin_pad, stream_ptr()),
          "kd_pack_conv_weight")
    return out


class DualUnsupported(_lib.KdccError):
    """conv2d(..., x2=...) on a shape the K-concatenated 1x1 kernel does not take: run the two convs instead."""


class ClsUnsupported(_lib.KdccError):
    """conv2d(..., cls_w=, cls_out=) on a problem whose kernel does not carry the classifier epilogue: store the activation and run the 1x1."""


def conv2d(x, w_packed, stride=1, pad=0, dil=1, *, res_pre=None, mask=None, mask_scale=None, res_post=None,
           out_raw=None, out_act=None, act_scale=None, act_shift=None, act_relu=False, algo_cin=None, algo_cout=None, bn_sums=None,
           x2=None, out_sums=None, cls_w=None, cls_out=None):
    """Implicit-GEMM conv; w_packed is (Cout,kh,kw,Cin). Outputs are caller-provided NHWC views.
    cls_w / cls_out: a (32,1,1,Cout) packed 1x1 classifier (rows >= ncls zero: pack_conv_weight of the weight padded to 32 outputs) and a
    (N,Ho,Wo,ncls) fp32 tensor: the classifier is applied to the activation act(scale * v + shift) in the conv's epilogue INSTEAD of storing
    it (no out_raw / out_act); raises ClsUnsupported where conv_cls_ok() is false.
    x2: a second (N,H,W,Cin2) source of a K-concatenated 1x1 conv -- w_packed is then (Cout,1,1,Cin + Cin2), the result
    [x | x2] . w^T in one accumulator chain (kd_conv1x1_dual_fwd); raises DualUnsupported where the kernel does not apply.
    bn_sums: an empty list (backward, with `mask`): when the kernel this problem selects can take the eval-BN parameter sums in
    its epilogue, (s1, s2) = per-channel sums of the masked gradient and of it times `mask` are appended -- what
    channel_sums(out_raw, sub=res_post, a=mask) would return from another pass over the tensors; otherwise it stays empty.
    out_sums: an empty list (forward, no epilogue operand, out_raw alone): when the selected kernel can sum its output per channel
    over blocks of 128 pixels in its epilogue, the [M/128][2][Cout] partial rows are appended (aspp_image_pool(..., sums=) takes
    them: the global average pool of the tensor without another pass over it); otherwise it stays empty."""
    _need_cuda(x, w_packed)
    N, H, W, Cin = x.shape
    Cout, kh, kw, Cin_w = w_packed.shape
    Cin2 = 0
    if x2 is not None:
        _need_cuda(x2)
        if tuple(x2.shape[:3]) != (N, H, W) or x2.dtype != x.dtype or (kh, kw, stride, pad) != (1, 1, 1, 0):
            raise ValueError("conv2d: x2 needs a 1x1 / stride-1 conv and a source of the same pixels and dtype")
        Cin2 = x2.shape[3]
    if Cin_w != Cin + Cin2:
        raise ValueError(f"conv2d: input has {Cin + Cin2} channels, packed weight expects {Cin_w}")
    if w_packed.dtype != x.dtype or not w_packed.is_contiguous():
        raise ValueError("conv2d: packed weight must be contiguous and of the activation dtype")
    Ho, Wo = conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil)
    d = ConvDesc(dt_of(x), N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, nhwc_ld(x))
    ep = ConvEpilogue()

    def chk(t, name, f32_ok=False):
        if t is None:
            return 0
        if tuple(t.shape) != (N, Ho, Wo, Cout):
            raise ValueError(f"conv2d: {name} has shape {tuple(t.shape)}, expected {(N, Ho, Wo, Cout)}")
        if t.dtype != x.dtype and not (f32_ok and t.dtype == torch.float32):
            raise ValueError(f"conv2d: {name} dtype {t.dtype} != {x.dtype}")
        _need_cuda(t)
        return nhwc_ld(t)

    def chk_vec(v, name):
        if v is not None and (v.dtype != torch.float32 or v.numel() != Cout or not v.is_contiguous()):
            raise ValueError(f"conv2d: {name} must be a contiguous fp32 vector of {Cout}")
        _need_cuda(v)

    ep.res_pre, ep.ld_res_pre = _ptr(res_pre), chk(res_pre, "res_pre")
    ep.mask, ep.ld_mask = _ptr(mask), chk(mask, "mask")
    chk_vec(mask_scale, "mask_scale"); ep.mask_scale = _ptr(mask_scale)
    ep.res_post, ep.ld_res_post = _ptr(res_post), chk(res_post, "res_post")
    ep.out_raw, ep.ld_raw = _ptr(out_raw), chk(out_raw, "out_raw", f32_ok=True)
    ep.raw_f32 = int(out_raw is not None and out_raw.dtype == torch.float32 and x.dtype != torch.float32)
    ep.out_act, ep.ld_act = _ptr(out_act), chk(out_act, "out_act")
    chk_vec(act_scale, "act_scale"); chk_vec(act_shift, "act_shift")
    ep.act_scale, ep.act_shift, ep.act_relu = _ptr(act_scale), _ptr(act_shift), int(act_relu)
    ep.bn_sums = None
    if (cls_w is None) != (cls_out is None):
        raise ValueError("conv2d: cls_w and cls_out come together")
    if cls_w is not None:
        _need_cuda(cls_w, cls_out)
        if tuple(cls_w.shape) != (32, 1, 1, Cout) or cls_w.dtype != x.dtype or not cls_w.is_contiguous():
            raise ValueError(f"conv2d: cls_w must be a contiguous (32,1,1,{Cout}) packed weight of the activation dtype")
        if cls_out.dtype != torch.float32 or tuple(cls_out.shape[:3]) != (N, Ho, Wo) or cls_out.shape[3] > 32 or cls_out.stride(3) != 1:
            raise ValueError("conv2d: cls_out must be (N,Ho,Wo,ncls <= 32) fp32")
        ep.cls_w, ep.cls_out, ep.ld_cls, ep.ncls = _ptr(cls_w), _ptr(cls_out), nhwc_ld(cls_out), cls_out.shape[3]
        if x2 is not None or not _lib.lib().kd_conv2d_cls_supported(C.byref(d), C.byref(ep)):
            raise ClsUnsupported(f"conv2d: no classifier epilogue for {kh}x{kw} {H}x{W} {Cin}->{Cout}")
    sums_rows = 0
    if bn_sums is not None and mask is not None:
        sums_rows = int(_lib.lib().kd_conv2d_bn_sums_rows(C.byref(d), C.byref(ep)))
        if sums_rows > 0:
            part = torch.empty((sums_rows, 2, Cout), dtype=torch.float32, device=x.device)
            ep.bn_sums = _ptr(part)
    out_rows = 0
    if out_sums is not None and mask is None and res_pre is None and res_post is None and out_act is None and out_raw is not None:
        out_rows = int(_lib.lib().kd_conv2d_bn_sums_rows(C.byref(d), C.byref(ep)))
        if out_rows > 0:
            opart = torch.empty((out_rows, 2, Cout), dtype=torch.float32, device=x.device)
            ep.bn_sums = _ptr(opart)
            out_sums.append(opart)
    prof = PROFILER
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    if x2 is not None:
        ld2 = nhwc_ld(x2)
        if not _lib.lib().kd_conv1x1_dual_supported(C.byref(d), Cin2, ld2, C.byref(ep)):
            raise DualUnsupported(f"conv2d: no K-concatenated kernel for {H}x{W} {Cin}+{Cin2}->{Cout}")
        check(_lib.lib().kd_conv1x1_dual_fwd(C.byref(d), _ptr(x), _ptr(x2), Cin2, ld2, _ptr(w_packed), C.byref(ep), stream_ptr()), "kd_conv1x1_dual_fwd")
    else:
        check(_lib.lib().kd_conv2d_fwd(C.byref(d), _ptr(x), _ptr(w_packed), C.byref(ep), stream_ptr()), "kd_conv2d_fwd")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        epi = "".join(c for c, t in (("p", res_pre), ("m", mask), ("q", res_post), ("r", out_raw), ("a", out_act), ("c", cls_out)) if t is not None)
        prof.append(("conv_igemm", 2.0 * N * Ho * Wo * (algo_cout or Cout) * kh * kw * (algo_cin or (Cin + Cin2)), e0, e1,
                     f"{kh}x{kw} s{stride} d{dil} {H}x{W} {Cin}{'+%d' % Cin2 if Cin2 else ''}->{Cout} [{epi}{'s' if sums_rows or out_rows else ''}]", _lib.last_kernel()))
    if sums_rows > 0:
        s12 = torch.empty((2, Cout), dtype=torch.float32, device=x.device)
        need = _lib.lib().kd_bn_sums_finish_workspace(sums_rows, Cout)
        ws = _ws(need, x.device) if need else None
        check(_lib.lib().kd_bn_sums_finish(_ptr(part), sums_rows, Cout, _ptr(s12[0]), _ptr(s12[1]), _ptr(ws), need, stream_ptr()),
              "kd_bn_sums_finish")
        bn_sums.append((s12[0], s12[1]))
    return out_raw, out_act


def conv_cls_ok(x, Cout, k=3, dil=1):
    """Does the kernel kd_conv2d_fwd selects for this 'same' conv of x carry the classifier epilogue (conv2d(..., cls_w=, cls_out=))?"""
    if not x.is_cuda or x.dtype != torch.bfloat16 or os.environ.get("KDCC_FUSE_CLS", "1") == "0":
        return False
    N, H, W, Cin = x.shape
    d = ConvDesc(dt_of(x), N, H, W, Cin, H, W, Cout, k, k, 1, dil * (k // 2), dil, nhwc_ld(x))
    ep = ConvEpilogue()
    ep.cls_w, ep.cls_out, ep.ld_cls, ep.ncls = C.c_void_p(256), C.c_void_p(256), 32, 19     # (the selection reads alignment and sizes only)
    return bool(_lib.lib().kd_conv2d_cls_supported(C.byref(d), C.byref(ep)))


def conv1x1_dual_ok(x, x2, Cout, operands=0):
    """Does the K-concatenated 1x1 kernel take [x | x2] -> Cout with `operands` epilogue operands (dense, freshly allocated outputs)?"""
    if x2 is None or not x.is_cuda or x.dtype != torch.bfloat16 or x2.dtype != x.dtype or tuple(x2.shape[:3]) != tuple(x.shape[:3]):
        return False
    N, H, W, Cin = x.shape
    d = ConvDesc(dt_of(x), N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, nhwc_ld(x))
    ep = ConvEpilogue()
    some = C.c_void_p(x.data_ptr())          # (alignment and strides are all the selection reads of an operand)
    if operands >= 1:
        ep.mask, ep.ld_mask = some, Cout
    if operands >= 2:
        ep.res_post, ep.ld_res_post = some, Cout
    if operands >= 3:
        ep.res_pre, ep.ld_res_pre = some, Cout
    ep.out_raw, ep.ld_raw = some, Cout
    return bool(_lib.lib().kd_conv1x1_dual_supported(C.byref(d), x2.shape[3], nhwc_ld(x2), C.byref(ep)))


def conv1x1_dual_ok_dims(N, H, W, Cin, Cin2, Cout, dtype, operands=0):
    """conv1x1_dual_ok for dense tensors of these sizes (before they exist)."""
    if dtype != torch.bfloat16:
        return False
    d = ConvDesc(KD_BF16, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, Cin)
    ep = ConvEpilogue()
    if operands >= 1:
        ep.mask, ep.ld_mask = C.c_void_p(256), Cout       # (the selection reads only alignment and strides of an operand)
    if operands >= 2:
        ep.res_post, ep.ld_res_post = C.c_void_p(256), Cout
    if operands >= 3:
        ep.res_pre, ep.ld_res_pre = C.c_void_p(256), Cout
    ep.out_raw, ep.ld_raw = C.c_void_p(256), Cout
    return bool(_lib.lib().kd_conv1x1_dual_supported(C.byref(d), Cin2, Cin2, C.byref(ep)))


def pw_wgrad(a, dy, dw, accumulate=False, workspace=None):
    """dw (Cout,Cin,1,1) fp32 = sum_pixels dy (N,H,W,Cout) x a (N,H,W,Cin)."""
    _need_cuda(a, dy, dw)
    N, H, W, Cin = a.shape
    Cout = dy.shape[3]
    if tuple(dy.shape[:3]) != (N, H, W) or a.dtype != dy.dtype:
        raise ValueError("pw_wgrad: a / dy mismatch")
    if dw.dtype != torch.float32 or dw.numel() != Cout * Cin or not dw.is_contiguous():
        raise ValueError("pw_wgrad: dw must be contiguous fp32 (Cout,Cin,1,1)")
    M = N * H * W
    need = _lib.lib().kd_pw_wgrad_workspace(M, Cin, Cout)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=a.device)
    e0 = _prof_start()
    check(_lib.lib().kd_pw_wgrad(dt_of(a), M, Cin, Cout, _ptr(a), nhwc_ld(a), _ptr(dy), nhwc_ld(dy), _ptr(dw),
                                 int(accumulate), _ptr(workspace), workspace.numel() * workspace.element_size(),
                                 stream_ptr()), "kd_pw_wgrad")
    _prof_stop(e0, "pw_wgrad", 2.0 * M * Cin * Cout, f"pw wgrad {H}x{W} {Cin}->{Cout}")
    return dw


def conv2d_wgrad(x, dy, dw, stride=1, pad=0, dil=1, accumulate=False, workspace=None):
    """dw (Cout,Cin,kh,kw) fp32 = weight gradient of conv2d(x; stride, pad, dil) given dy (N,Ho,Wo,Cout)."""
    _need_cuda(x, dy, dw)
    N, H, W, Cin = x.shape
    Cout, Cin_w, kh, kw = dw.shape
    Ho, Wo = conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil)
    if Cin_w != Cin or tuple(dy.shape) != (N, Ho, Wo, Cout) or dy.dtype != x.dtype:
        raise ValueError(f"conv2d_wgrad: x {tuple(x.shape)} / dy {tuple(dy.shape)} / dw {tuple(dw.shape)} mismatch")
    if dw.dtype != torch.float32 or not dw.is_contiguous():
        raise ValueError("conv2d_wgrad: dw must be contiguous fp32 (Cout,Cin,kh,kw)")
    d = ConvDesc(dt_of(x), N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, nhwc_ld(x))
    need = _lib.lib().kd_conv2d_wgrad_workspace(C.byref(d))
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
    prof = PROFILER
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().kd_conv2d_wgrad(C.byref(d), _ptr(x), _ptr(dy), nhwc_ld(dy), _ptr(dw), int(accumulate), _ptr(workspace),
                                     workspace.numel() * workspace.element_size(), stream_ptr()), "kd_conv2d_wgrad")
    if prof is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        prof.append(("conv_wgrad", 2.0 * N * Ho * Wo * Cout * kh * kw * Cin, e0, e1,
                     f"wgrad {kh}x{kw} s{stride} d{dil} {H}x{W} {Cin}->{Cout}", _lib.last_kernel()))
    return dw


# --------------------------------------------------------------------------- depthwise conv
def pack_dw_weight(w, flip=False):
    """(C,1,k,k) fp32 -> tap-major [k*k][C] fp32 (flip=True: the dgrad operand)."""
    _need_cuda(w)
    w = w.detach().contiguous().float()
    Cc, one, k, k2 = w.shape
    if one != 1 or k != k2:
        raise ValueError("pack_dw_weight: expected (C,1,k,k)")
    out = torch.empty((k * k, Cc), dtype=torch.float32, device=w.device)
    check(_lib.lib().kd_pack_dw_weight(_ptr(w), _ptr(out), Cc, k, int(flip), stream_ptr()), "kd_pack_dw_weight")
    return out


def _dw_desc(x, k, pad, dil, y=None):
    N, H, W, Cc = x.shape
    return DwDesc(dt_of(x), N, H, W, Cc, k, pad, dil, nhwc_ld(x), nhwc_ld(y) if y is not None else Cc)


def dwconv(x, w_taps, k, pad, dil, bias=None, out=None, res_pre=None, mask=None, mask_scale=None, res_post=None):
    _need_cuda(x, w_taps, bias, out, res_pre, mask, mask_scale, res_post)
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        bias = bias.detach().float().contiguous()      # the kernel reads fp32 (C) (a bf16 module's parameter is not)
    N, H, W, Cc = x.shape
    if tuple(w_taps.shape) != (k * k, Cc) or w_taps.dtype != torch.float32 or not w_taps.is_contiguous():
        raise ValueError("dwconv: w_taps must be contiguous fp32 [k*k][C]")
    if out is None:
        out = torch.empty((N, H, W, Cc), dtype=x.dtype, device=x.device)
    if tuple(out.shape) != (N, H, W, Cc) or out.dtype != x.dtype:
        raise ValueError("dwconv: bad output view")
    d = _dw_desc(x, k, pad, dil, out)
    ep = None
    if res_pre is not None or mask is not None or res_post is not None:
        ep = DwEpilogue()
        for name, t in (("res_pre", res_pre), ("mask", mask), ("res_post", res_post)):
            if t is not None and (tuple(t.shape) != (N, H, W, Cc) or t.dtype != x.dtype):
                raise ValueError(f"dwconv: {name} must match the output shape/dtype")
            setattr(ep, name, _ptr(t))
            setattr(ep, "ld_" + name, nhwc_ld(t) if t is not None else 0)
        if mask_scale is not None and (mask_scale.dtype != torch.float32 or mask_scale.numel() != Cc):
            raise ValueError("dwconv: mask_scale must be fp32 (C,)")
        ep.mask_scale = _ptr(mask_scale)
    e0 = _prof_start()
    check(_lib.lib().kd_dwconv_fwd(C.byref(d), _ptr(x), _ptr(w_taps), _ptr(bias), C.byref(ep) if ep is not None else None,
                                   _ptr(out), stream_ptr()), "kd_dwconv_fwd")
    _prof_stop(e0, "depthwise", _nbytes(x, out, res_pre, mask, res_post), f"dw {k}x{k} d{dil} {H}x{W} C{Cc}" + (" +epi" if ep is not None else ""))
    return out


def scale_by_device_scalar_(x, scale):
    """x *= scale (0-dim / 1-element device tensor) in place; when the scalar is exactly 1 -- decided on the device, no host
    sync -- nothing is read or written.  The upstream-gradient factor of the fused loss Functions (layerwise_trainer.py:229-235)."""
    _need_cuda(x, scale)
    if scale.numel() != 1:
        raise ValueError("scale_by_device_scalar_: scale must hold one element")
    dense = x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))
    if not dense:
        x.mul_(scale.to(x.dtype))     # a strided view: plain torch
        return x
    scale = scale.detach().reshape(1)
    if scale.dtype != torch.float32:
        scale = scale.float()
    check(_lib.lib().kd_scale_by_device_scalar(_ptr(x), dt_of(x), x.numel(), _ptr(scale), stream_ptr()), "kd_scale_by_device_scalar")
    return x


def dwconv_sum(xs, w_taps, k, pad, dil, out=None):
    """out = sum_i dwconv(xs[i], w_taps[i]): the input gradient of a tensor that feeds several depthwise convs of one geometry
    (deeplabv3.py:64-75, the ASPP input under its replaced branches), summed inside one launch where the shape allows."""
    xs, w_taps = list(xs), list(w_taps)
    if not xs or len(xs) != len(w_taps):
        raise ValueError("dwconv_sum: need as many tap tables as inputs")
    _need_cuda(*xs, *w_taps, out)
    N, H, W, Cc = xs[0].shape
    ld = nhwc_ld(xs[0])
    for x, w in zip(xs, w_taps):
        if tuple(x.shape) != (N, H, W, Cc) or x.dtype != xs[0].dtype or nhwc_ld(x) != ld:
            raise ValueError("dwconv_sum: inputs must share shape, dtype and pixel stride")
        if tuple(w.shape) != (k * k, Cc) or w.dtype != torch.float32 or not w.is_contiguous():
            raise ValueError("dwconv_sum: w_taps must be contiguous fp32 [k*k][C]")
    if out is None:
        out = torch.empty((N, H, W, Cc), dtype=xs[0].dtype, device=xs[0].device)
    if tuple(out.shape) != (N, H, W, Cc) or out.dtype != xs[0].dtype:
        raise ValueError("dwconv_sum: bad output view")
    d = _dw_desc(xs[0], k, pad, dil, out)
    n = len(xs)
    xp = (C.c_void_p * n)(*[_ptr(x) for x in xs])
    wp = (C.c_void_p * n)(*[_ptr(w) for w in w_taps])
    e0 = _prof_start()
    check(_lib.lib().kd_dwconv_fwd_sum(C.byref(d), n, xp, wp, _ptr(out), stream_ptr()), "kd_dwconv_fwd_sum")
    _prof_stop(e0, "depthwise", _nbytes(out, *xs), f"dw sum of {n} {k}x{k} d{dil} {H}x{W} C{Cc}")
    return out


def dwconv_fanout(x, w_taps, k, pad, dil, outs=None):
    """[dwconv(x, w) for w in w_taps]: several depthwise convs of one geometry reading the same tensor (deeplabv3.py:71-75,
    the ASPP branches), one pass over x where the shape allows."""
    w_taps = list(w_taps)
    if not w_taps:
        raise ValueError("dwconv_fanout: no tap tables")
    _need_cuda(x, *w_taps, *(outs or []))
    N, H, W, Cc = x.shape
    for w in w_taps:
        if tuple(w.shape) != (k * k, Cc) or w.dtype != torch.float32 or not w.is_contiguous():
            raise ValueError("dwconv_fanout: w_taps must be contiguous fp32 [k*k][C]")
    if outs is None:
        outs = [torch.empty((N, H, W, Cc), dtype=x.dtype, device=x.device) for _ in w_taps]
    outs = list(outs)
    if len(outs) != len(w_taps):
        raise ValueError("dwconv_fanout: need one output per tap table")
    ld = nhwc_ld(outs[0])
    for o in outs:
        if tuple(o.shape) != (N, H, W, Cc) or o.dtype != x.dtype or nhwc_ld(o) != ld:
            raise ValueError("dwconv_fanout: outputs must share shape, dtype and pixel stride")
    d = _dw_desc(x, k, pad, dil, outs[0])
    n = len(outs)
    wp = (C.c_void_p * n)(*[_ptr(w) for w in w_taps])
    yp = (C.c_void_p * n)(*[_ptr(o) for o in outs])
    e0 = _prof_start()
    check(_lib.lib().kd_dwconv_fwd_fanout(C.byref(d), n, _ptr(x), wp, yp, stream_ptr()), "kd_dwconv_fwd_fanout")
    _prof_stop(e0, "depthwise", _nbytes(x, *outs), f"dw fan-out of {n} {k}x{k} d{dil} {H}x{W} C{Cc}")
    return outs


def dwconv_wgrad(x, dy, dw, k, pad, dil, accumulate=False, workspace=None):
    _need_cuda(x, dy, dw)
    N, H, W, Cc = x.shape
    if tuple(dy.shape) != (N, H, W, Cc) or dy.dtype != x.dtype:
        raise ValueError("dwconv_wgrad: x / dy mismatch")
    if dw.dtype != torch.float32 or dw.numel() != Cc * k * k or not dw.is_contiguous():
        raise ValueError("dwconv_wgrad: dw must be contiguous fp32 (C,1,k,k)")
    d = _dw_desc(x, k, pad, dil)
    need = _lib.lib().kd_dwconv_wgrad_workspace(C.byref(d))
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
    e0 = _prof_start()
    check(_lib.lib().kd_dwconv_wgrad(C.byref(d), _ptr(x), _ptr(dy), nhwc_ld(dy), _ptr(dw), int(accumulate), _ptr(workspace),
                                     workspace.numel() * workspace.element_size(), stream_ptr()), "kd_dwconv_wgrad")
    _prof_stop(e0, "depthwise", _nbytes(x, dy), f"dw wgrad {k}x{k} d{dil} {H}x{W} C{Cc}")
    return dw


def dwconv_wgrad_multi(x, dys, dws, k, pad, dil, accumulate=False):
    """dws[i] = weight gradient of dwconv(x, .) given dys[i]: several depthwise convs of one geometry reading the same tensor
    (deeplabv3.py:71-75, the replaced ASPP branches); one launch for two or three branches where the shape allows."""
    dys, dws = list(dys), list(dws)
    if not dys or len(dys) != len(dws):
        raise ValueError("dwconv_wgrad_multi: need one gradient buffer per dy")
    _need_cuda(x, *dys, *dws)
    N, H, W, Cc = x.shape
    ld = nhwc_ld(dys[0])
    for dy, dw in zip(dys, dws):
        if tuple(dy.shape) != (N, H, W, Cc) or dy.dtype != x.dtype or nhwc_ld(dy) != ld:
            raise ValueError("dwconv_wgrad_multi: every dy must match x's shape / dtype and share one pixel stride")
        if dw.dtype != torch.float32 or dw.numel() != Cc * k * k or not dw.is_contiguous():
            raise ValueError("dwconv_wgrad_multi: dw must be contiguous fp32 (C,1,k,k)")
    d = _dw_desc(x, k, pad, dil)
    n = len(dys)
    need = _lib.lib().kd_dwconv_wgrad_multi_workspace(C.byref(d), n)
    workspace = _ws(need, x.device)
    yp = (C.c_void_p * n)(*[_ptr(t) for t in dys])
    wp = (C.c_void_p * n)(*[_ptr(t) for t in dws])
    e0 = _prof_start()
    check(_lib.lib().kd_dwconv_wgrad_multi(C.byref(d), n, _ptr(x), yp, ld, wp, int(accumulate), _ptr(workspace), need, stream_ptr()),
          "kd_dwconv_wgrad_multi")
    _prof_stop(e0, "depthwise", _nbytes(x, *dys), f"dw wgrad of {n} {k}x{k} d{dil} {H}x{W} C{Cc}")
    return dws


# ------------------------------------------------------------------ lattice-planar intermediates (include/kdcc.h)
class Lattice:
    """An (N,H,W,C) bf16 tensor in the lattice-planar layout of dilation `dil`: `t` is (C/16, rows, 16) -- the depthwise outputs of
    the replaced ASPP branches and their gradients, which only the depthwise kernels and the 1x1 convs next to them read
    (depthwise_separable_conv.py:11-13 under deeplabv3.py:64-75).  Not an image any more: `to_nhwc()` is for tests."""

    def __init__(self, N, H, W, Cc, dil, dtype=torch.bfloat16, device="cuda", t=None):
        self.N, self.H, self.W, self.C, self.dil = N, H, W, Cc, dil
        self.rows = lattice_rows(N, H, W, dil)
        if Cc % 16:
            raise ValueError("Lattice: channels must be a multiple of 16")
        self.t = t if t is not None else torch.empty((Cc // 16, self.rows, 16), dtype=dtype, device=device)
        if tuple(self.t.shape) != (Cc // 16, self.rows, 16) or not self.t.is_contiguous():
            raise ValueError("Lattice: bad storage")
        used = N * dil * dil * (-(-H // dil)) * (-(-W // dil))
        if t is None and used < self.rows:
            self.t[:, used:].zero_()          # the tail rows of every plane hold zeros (nobody else writes them)

    dtype = property(lambda self: self.t.dtype)
    device = property(lambda self: self.t.device)
    shape = property(lambda self: (self.N, self.H, self.W, self.C))
    plane = property(lambda self: self.rows * 16)

    def same_geometry(self, o):
        return (self.N, self.H, self.W, self.C, self.dil, self.dtype) == (o.N, o.H, o.W, o.C, o.dil, o.dtype)

    def to_nhwc(self):
        """Image-order copy (tests / debugging): plain torch indexing, not a kernel."""
        rows = self.t.permute(1, 0, 2).reshape(self.rows, self.C)
        out = torch.empty((self.N, self.H, self.W, self.C), dtype=self.dtype, device=self.device)
        return lattice_to_image(rows, out, self.dil)


def lattice_rows(N, H, W, dil):
    return int(_lib.lib().kd_lattice_rows(N, H, W, dil))


def dwconv_lattice_ok(x, n, k, pad, dil):
    """Can n (2, 3) depthwise convs of this geometry reading / summing into x-shaped tensors keep lattice-planar intermediates?"""
    if not x.is_cuda or x.dtype != torch.bfloat16:
        return False
    return bool(_lib.lib().kd_dwconv_lattice_ok(C.byref(_dw_desc(x, k, pad, dil)), n))


def image_to_lattice(img, dil, out=None):
    """(N,H,W,C) view -> dense [rows][C] in lattice row order (zero rows where a lattice cell has no pixel)."""
    _need_cuda(img, out)
    N, H, W, Cc = img.shape
    rows = lattice_rows(N, H, W, dil)
    if out is None:
        out = torch.empty((rows, Cc), dtype=img.dtype, device=img.device)
    if tuple(out.shape) != (rows, Cc) or out.dtype != img.dtype or out.stride(1) != 1:
        raise ValueError("image_to_lattice: bad output")
    e0 = _prof_start()
    check(_lib.lib().kd_lattice_rows_move(dt_of(img), N, H, W, Cc, dil, _ptr(img), nhwc_ld(img), _ptr(out), out.stride(0), 1, stream_ptr()),
          "kd_lattice_rows_move")
    _prof_stop(e0, "plumbing", _nbytes(img, out), f"rows to lattice order {H}x{W} C{Cc}")
    return out


def lattice_to_image(rows, img, dil):
    """Dense [rows][C] in lattice row order -> the (N,H,W,C) view `img`."""
    _need_cuda(img, rows)
    N, H, W, Cc = img.shape
    if tuple(rows.shape) != (lattice_rows(N, H, W, dil), Cc) or rows.dtype != img.dtype or rows.stride(1) != 1:
        raise ValueError("lattice_to_image: bad input")
    e0 = _prof_start()
    check(_lib.lib().kd_lattice_rows_move(dt_of(img), N, H, W, Cc, dil, _ptr(img), nhwc_ld(img), _ptr(rows), rows.stride(0), 0, stream_ptr()),
          "kd_lattice_rows_move")
    _prof_stop(e0, "plumbing", _nbytes(img, rows), f"rows to image order {H}x{W} C{Cc}")
    return img


def dwconv_fanout_lattice(x, w_taps, k, pad, dil, outs=None):
    """dwconv_fanout with lattice-planar outputs (n = 2, 3; dwconv_lattice_ok)."""
    w_taps = list(w_taps)
    _need_cuda(x, *w_taps)
    N, H, W, Cc = x.shape
    for w in w_taps:
        if tuple(w.shape) != (k * k, Cc) or w.dtype != torch.float32 or not w.is_contiguous():
            raise ValueError("dwconv_fanout_lattice: w_taps must be contiguous fp32 [k*k][C]")
    if outs is None:
        outs = [Lattice(N, H, W, Cc, dil, x.dtype, x.device) for _ in w_taps]
    outs = list(outs)
    if len(outs) != len(w_taps) or any(o.shape != (N, H, W, Cc) or o.dil != dil or o.dtype != x.dtype for o in outs):
        raise ValueError("dwconv_fanout_lattice: need one matching Lattice per tap table")
    d = _dw_desc(x, k, pad, dil)
    n = len(outs)
    wp = (C.c_void_p * n)(*[_ptr(w) for w in w_taps])
    yp = (C.c_void_p * n)(*[_ptr(o.t) for o in outs])
    e0 = _prof_start()
    check(_lib.lib().kd_dwconv_fwd_fanout_lattice(C.byref(d), n, _ptr(x), wp, yp, stream_ptr()), "kd_dwconv_fwd_fanout_lattice")
    _prof_stop(e0, "depthwise", x.numel() * x.element_size() * (1 + n), f"dw fan-out of {n} {k}x{k} d{dil} {H}x{W} C{Cc} [lattice]")
    return outs


def dwconv_sum_lattice(xs, w_taps, k, pad, dil, out=None):
    """dwconv_sum over lattice-planar inputs (n = 2, 3); the sum is an NHWC tensor."""
    xs, w_taps = list(xs), list(w_taps)
    if len(xs) != len(w_taps) or not xs or any(not xs[0].same_geometry(x) for x in xs) or xs[0].dil != dil:
        raise ValueError("dwconv_sum_lattice: need matching Lattice inputs, one tap table each")
    N, H, W, Cc = xs[0].shape
    _need_cuda(*[x.t for x in xs], *w_taps, out)
    for w in w_taps:
        if tuple(w.shape) != (k * k, Cc) or w.dtype != torch.float32 or not w.is_contiguous():
            raise ValueError("dwconv_sum_lattice: w_taps must be contiguous fp32 [k*k][C]")
    if out is None:
        out = torch.empty((N, H, W, Cc), dtype=xs[0].dtype, device=xs[0].device)
    if tuple(out.shape) != (N, H, W, Cc) or out.dtype != xs[0].dtype:
        raise ValueError("dwconv_sum_lattice: bad output view")
    d = DwDesc(dt_of(out), N, H, W, Cc, k, pad, dil, Cc, nhwc_ld(out))
    n = len(xs)
    xp = (C.c_void_p * n)(*[_ptr(x.t) for x in xs])
    wp = (C.c_void_p * n)(*[_ptr(w) for w in w_taps])
    e0 = _prof_start()
    check(_lib.lib().kd_dwconv_fwd_sum_lattice(C.byref(d), n, xp, wp, _ptr(out), stream_ptr()), "kd_dwconv_fwd_sum_lattice")
    _prof_stop(e0, "depthwise", out.numel() * out.element_size() * (1 + n), f"dw sum of {n} {k}x{k} d{dil} {H}x{W} C{Cc} [lattice]")
    return out


def dwconv_wgrad_multi_lattice(x, dys, dws, k, pad, dil, accumulate=False):
    """dwconv_wgrad_multi with lattice-planar gradients dys (n = 2, 3)."""
    dys, dws = list(dys), list(dws)
    N, H, W, Cc = x.shape
    if len(dys) != len(dws) or not dys or any(dy.shape != (N, H, W, Cc) or dy.dil != dil or dy.dtype != x.dtype for dy in dys):
        raise ValueError("dwconv_wgrad_multi_lattice: need one matching Lattice gradient per weight gradient")
    _need_cuda(x, *[dy.t for dy in dys], *dws)
    for dw in dws:
        if dw.dtype != torch.float32 or dw.numel() != Cc * k * k or not dw.is_contiguous():
            raise ValueError("dwconv_wgrad_multi_lattice: dw must be contiguous fp32 (C,1,k,k)")
    d = _dw_desc(x, k, pad, dil)
    n = len(dys)
    need = _lib.lib().kd_dwconv_wgrad_multi_workspace(C.byref(d), n)
    workspace = _ws(need, x.device)
    yp = (C.c_void_p * n)(*[_ptr(t.t) for t in dys])
    wp = (C.c_void_p * n)(*[_ptr(t) for t in dws])
    e0 = _prof_start()
    check(_lib.lib().kd_dwconv_wgrad_multi_lattice(C.byref(d), n, _ptr(x), yp, wp, int(accumulate), _ptr(workspace), need, stream_ptr()),
          "kd_dwconv_wgrad_multi_lattice")
    _prof_stop(e0, "depthwise", x.numel() * x.element_size() * (1 + n), f"dw wgrad of {n} {k}x{k} d{dil} {H}x{W} C{Cc} [lattice]")
    return dws


# ------------------------------------------------------------------------------ trunk plumbing
def stem_conv(x_nchw, w, dtype):
    """(N,3,H,W) fp32 NCHW batch + (64,3,3,3) fp32 weight -> (N,H,W,64) NHWC."""
    _need_cuda(x_nchw, w)
    if x_nchw.dtype != torch.float32 or not x_nchw.is_contiguous() or x_nchw.shape[1] != 3:
        raise ValueError("stem_conv: expects a contiguous fp32 (N,3,H,W) batch")
    if tuple(w.shape) != (64, 3, 3, 3) or w.dtype != torch.float32 or not w.is_contiguous():
        raise ValueError("stem_conv: expects a contiguous fp32 (64,3,3,3) weight")
    N, _, H, W = x_nchw.shape
    y = torch.empty((N, H, W, 64), dtype=dtype, device=x_nchw.device)
    check(_lib.lib().kd_stem_conv(dt_of(y), _ptr(x_nchw), _ptr(w), _ptr(y), N, H, W, stream_ptr()), "kd_stem_conv")
    return y


def stem_conv_pool(x_nchw, w, scale=None, shift=None, want_raw=True):
    """stem conv -> pool2 (-> BN/ReLU of the consumer) in one pass, bf16, the full-resolution tensor never stored
    (wider_resnet.py:307-309, 353-356).  Returns (raw, act) like maxpool3x3s2; bit-identical to stem_conv + maxpool3x3s2."""
    _need_cuda(x_nchw, w, scale, shift)
    if x_nchw.dtype != torch.float32 or not x_nchw.is_contiguous() or x_nchw.shape[1] != 3:
        raise ValueError("stem_conv_pool: expects a contiguous fp32 (N,3,H,W) batch")
    if tuple(w.shape) != (64, 3, 3, 3) or w.dtype != torch.float32 or not w.is_contiguous():
        raise ValueError("stem_conv_pool: expects a contiguous fp32 (64,3,3,3) weight")
    N, _, H, W = x_nchw.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y_raw = torch.empty((N, Ho, Wo, 64), dtype=torch.bfloat16, device=x_nchw.device) if want_raw else None
    y_act = torch.empty((N, Ho, Wo, 64), dtype=torch.bfloat16, device=x_nchw.device) if scale is not None else None
    check(_lib.lib().kd_stem_conv_pool(_ptr(x_nchw), _ptr(w), _ptr(y_raw), _ptr(y_act), _ptr(scale), _ptr(shift), N, H, W,
                                       stream_ptr()), "kd_stem_conv_pool")
    return y_raw, y_act


def maxpool3x3s2(x, scale=None, shift=None, want_raw=True):
    _need_cuda(x, scale, shift)
    N, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y_raw = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device) if want_raw else None
    y_act = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device) if scale is not None else None
    check(_lib.lib().kd_maxpool3x3s2(dt_of(x), _ptr(x), nhwc_ld(x), _ptr(y_raw), Cc, _ptr(y_act), Cc, _ptr(scale), _ptr(shift),
                                     N, H, W, Cc, stream_ptr()), "kd_maxpool3x3s2")
    return y_raw, y_act


def upsample_bilinear_ac(x, size, out=None, out_dtype=None, align_corners=True):
    _need_cuda(x, out)
    N, H, W, Cc = x.shape
    Ho, Wo = size
    if out is None:
        out = torch.empty((N, Ho, Wo, Cc), dtype=out_dtype or x.dtype, device=x.device)
    if tuple(out.shape) != (N, Ho, Wo, Cc):
        raise ValueError("upsample: bad output view")
    check(_lib.lib().kd_upsample_bilinear(_ptr(x), dt_of(x), nhwc_ld(x), _ptr(out), dt_of(out), nhwc_ld(out), N, H, W, Cc,
                                          Ho, Wo, int(bool(align_corners)), stream_ptr()), "kd_upsample_bilinear")
    return out


def aspp_image_pool(x, w, scale, shift, out, sums=None):
    """x (N,H,W,Cin); w (Cout,Cin[,1,1]) fp32; writes the broadcast branch into `out` (N,H,W,Cout) view.
    sums: the [N*H*W/128][2][Cin] partial rows the conv that produced x took in its epilogue (conv2d(out_sums=)): x is not read."""
    _need_cuda(x, w, scale, shift, out)
    N, H, W, Cin = x.shape
    Cout = out.shape[3]
    w = w.reshape(Cout, Cin)
    if w.dtype != torch.float32 or not w.is_contiguous() or tuple(out.shape[:3]) != (N, H, W) or out.dtype != x.dtype:
        raise ValueError("aspp_image_pool: bad operands")
    need = _lib.lib().kd_aspp_image_pool_workspace(N, Cin, Cout)
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    if sums is not None:
        _need_cuda(sums)
        if tuple(sums.shape) != (N * H * W // 128, 2, Cin) or (H * W) % 128 or sums.dtype != torch.float32 or not sums.is_contiguous():
            raise ValueError("aspp_image_pool: sums must be the contiguous fp32 [N*H*W/128][2][Cin] partial rows of x")
        check(_lib.lib().kd_aspp_image_pool_sums(dt_of(x), _ptr(sums), _ptr(w), _ptr(scale), _ptr(shift), _ptr(out), nhwc_ld(out),
                                                 N, H, W, Cin, Cout, _ptr(ws), need, stream_ptr()), "kd_aspp_image_pool_sums")
        return out
    check(_lib.lib().kd_aspp_image_pool(dt_of(x), _ptr(x), nhwc_ld(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(out),
                                        nhwc_ld(out), N, H, W, Cin, Cout, _ptr(ws), need, stream_ptr()), "kd_aspp_image_pool")
    return out


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def stem_wgrad(x_nchw, dy, dw, accumulate=False):
    """dw (64,3,3,3) fp32 from the NCHW fp32 batch and dy (N,H,W,64)."""
    _need_cuda(x_nchw, dy, dw)
    N, _, H, W = x_nchw.shape
    if x_nchw.dtype != torch.float32 or not x_nchw.is_contiguous() or x_nchw.shape[1] != 3 or tuple(dy.shape) != (N, H, W, 64):
        raise ValueError("stem_wgrad: expects a contiguous fp32 (N,3,H,W) batch and dy (N,H,W,64)")
    if tuple(dw.shape) != (64, 3, 3, 3) or dw.dtype != torch.float32 or not dw.is_contiguous():
        raise ValueError("stem_wgrad: dw must be contiguous fp32 (64,3,3,3)")
    need = _lib.lib().kd_stem_wgrad_workspace(N, H, W)
    ws = _ws(need, dy.device)
    check(_lib.lib().kd_stem_wgrad(dt_of(dy), _ptr(x_nchw), _ptr(dy), nhwc_ld(dy), _ptr(dw), N, H, W, int(accumulate), _ptr(ws), need,
                                   stream_ptr()), "kd_stem_wgrad")
    return dw


def maxpool3x3s2_bwd(x, gy, out=None):
    _need_cuda(x, gy, out)
    N, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if tuple(gy.shape) != (N, Ho, Wo, Cc) or gy.dtype != x.dtype:
        raise ValueError("maxpool3x3s2_bwd: gy must be (N,Ho,Wo,C) of the input dtype")
    if out is None:
        out = torch.empty((N, H, W, Cc), dtype=x.dtype, device=x.device)
    need = _lib.lib().kd_maxpool3x3s2_bwd_workspace(N, H, W, Cc)
    ws = _ws(need, x.device)
    check(_lib.lib().kd_maxpool3x3s2_bwd(dt_of(x), _ptr(x), nhwc_ld(x), _ptr(gy), nhwc_ld(gy), _ptr(out), nhwc_ld(out), N, H, W, Cc,
                                         _ptr(ws), need, stream_ptr()), "kd_maxpool3x3s2_bwd")
    return out


def upsample_bilinear_ac_bwd(gy, size, out=None, out_dtype=None, align_corners=True):
    """gy (N,Ho,Wo,C) -> gradient w.r.t. the (N,H,W,C) input of upsample_bilinear_ac (same corner convention), size = (H, W)."""
    _need_cuda(gy, out)
    N, Ho, Wo, Cc = gy.shape
    H, W = size
    if out is None:
        out = torch.empty((N, H, W, Cc), dtype=out_dtype or gy.dtype, device=gy.device)
    if tuple(out.shape) != (N, H, W, Cc):
        raise ValueError("upsample_bilinear_ac_bwd: bad output view")
    need = _lib.lib().kd_upsample_bilinear_ac_bwd_workspace(N, H, W, Cc, Ho, Wo)
    ws = _ws(need, gy.device)
    check(_lib.lib().kd_upsample_bilinear_bwd(_ptr(gy), dt_of(gy), nhwc_ld(gy), _ptr(out), dt_of(out), nhwc_ld(out), N, H, W, Cc,
                                              Ho, Wo, int(bool(align_corners)), _ptr(ws), need, stream_ptr()), "kd_upsample_bilinear_bwd")
    return out


def zero_insert(x, stride, size):
    _need_cuda(x)
    N, H, W, Cc = x.shape
    Hy, Wy = size
    y = torch.empty((N, Hy, Wy, Cc), dtype=x.dtype, device=x.device)
    check(_lib.lib().kd_zero_insert(dt_of(x), _ptr(x), nhwc_ld(x), _ptr(y), Cc, N, H, W, Cc, stride, Hy, Wy, stream_ptr()),
          "kd_zero_insert")
    return y


def relu_bn_bwd(g, mask, scale, res=None, out=None):
    """(mask > 0 ? g * scale[c] : 0) + res on (N,H,W,C) views."""
    _need_cuda(g, mask, scale, res, out)
    N, H, W, Cc = g.shape
    if tuple(mask.shape) != (N, H, W, Cc) or mask.dtype != g.dtype or scale.dtype != torch.float32 or scale.numel() != Cc:
        raise ValueError("relu_bn_bwd: operand mismatch")
    if out is None:
        out = torch.empty((N, H, W, Cc), dtype=g.dtype, device=g.device)
    ldg, ldm, ldo = nhwc_ld(g), nhwc_ld(mask), nhwc_ld(out)
    check(_lib.lib().kd_relu_bn_bwd(dt_of(g), _ptr(g), ldg, _ptr(mask), ldm, _ptr(scale), _ptr(res), nhwc_ld(res) if res is not None else 0,
                                    _ptr(out), ldo, N * H * W, Cc, stream_ptr()), "kd_relu_bn_bwd")
    return out


def channel_sums(g, sub=None, a=None, per_image=False):
    """Per-channel sums over the pixels of g (N,H,W,C) [minus sub], optionally also of (g - sub) * a.
    Returns (s1, s2 | None) fp32 of shape (C,) or, with per_image, (N, C)."""
    _need_cuda(g, sub, a)
    N, H, W, Cc = g.shape
    groups, rows = (N, H * W) if per_image else (1, N * H * W)
    s1 = torch.empty((groups, Cc), dtype=torch.float32, device=g.device)
    s2 = torch.empty_like(s1) if a is not None else None
    need = _lib.lib().kd_channel_sums_workspace(groups, rows, Cc)
    ws = _ws(need, g.device)
    ld = lambda t: nhwc_ld(t) if t is not None else 0
    for t in (sub, a):
        if t is not None and (tuple(t.shape) != (N, H, W, Cc) or t.dtype != g.dtype):
            raise ValueError("channel_sums: operand mismatch")
    e0 = _prof_start()
    check(_lib.lib().kd_channel_sums(dt_of(g), _ptr(g), ld(g), _ptr(sub), ld(sub), _ptr(a), ld(a), groups, rows, Cc, _ptr(s1), _ptr(s2),
                                     _ptr(ws), need, stream_ptr()), "kd_channel_sums")
    _prof_stop(e0, "channel_sums", float(_nbytes(g, sub, a)), f"channel sums {H}x{W} C{Cc}{' -sub' if sub is not None else ''}{' *a' if a is not None else ''}",
               "channel_sums_partial_kernel")
    if not per_image:
        s1 = s1[0]
        s2 = s2[0] if s2 is not None else None
    return s1, s2


def bn_eval_param_grads(s1, s2, scale, gamma, beta, dgamma, dbeta, accumulate=False):
    _need_cuda(s1, s2, scale, gamma, beta, dgamma, dbeta)
    Cc = s1.numel()
    for t in (s1, s2, scale, gamma, beta, dgamma, dbeta):
        if t.dtype != torch.float32 or t.numel() != Cc or not t.is_contiguous():
            raise ValueError("bn_eval_param_grads: contiguous fp32 (C,) vectors required")
    check(_lib.lib().kd_bn_eval_param_grads(_ptr(s1), _ptr(s2), _ptr(scale), _ptr(gamma), _ptr(beta), _ptr(dgamma), _ptr(dbeta), Cc,
                                            int(accumulate), stream_ptr()), "kd_bn_eval_param_grads")


def broadcast_add(v, y, alpha=1.0, accumulate=True):
    """y[n,h,w,c] (+)= alpha * v[n,c]."""
    _need_cuda(v, y)
    N, H, W, Cc = y.shape
    if tuple(v.shape) != (N, Cc) or v.dtype != torch.float32 or not v.is_contiguous():
        raise ValueError("broadcast_add: v must be contiguous fp32 (N,C)")
    check(_lib.lib().kd_broadcast_add(dt_of(y), _ptr(v), _ptr(y), nhwc_ld(y), N, H * W, Cc, C.c_float(alpha), int(accumulate),
                                      stream_ptr()), "kd_broadcast_add")
    return y


def bn_fold(bn):
    """Eval-mode nn.BatchNorm2d -> (scale, shift) fp32 device vectors."""
    g, b, m, v = bn.weight.detach().float(), bn.bias.detach().float(), bn.running_mean.float(), bn.running_var.float()
    _need_cuda(g, b, m, v)
    scale, shift = torch.empty_like(g), torch.empty_like(g)
    check(_lib.lib().kd_bn_fold(_ptr(g.contiguous()), _ptr(b.contiguous()), _ptr(m.contiguous()), _ptr(v.contiguous()),
                                C.c_float(bn.eps), _ptr(scale), _ptr(shift), g.numel(), stream_ptr()), "kd_bn_fold")
    return scale, shift


# ------------------------------------------------------------------------- small-shape path (NCHW fp32)
def _nchw32(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
            raise ValueError("small-shape kernels take contiguous fp32 device tensors (NCHW)")


def _dconv_desc(x_shape, w_shape, stride, pad, dil, groups):
    N, Cc, H, W = x_shape
    K, cg, kh, kw = w_shape
    if cg * groups != Cc:
        raise ValueError(f"direct conv: weight {tuple(w_shape)} does not match {Cc} input channels / {groups} groups")
    return DConvDesc(N, Cc, H, W, K, kh, kw, stride, pad, dil, groups), (N, K, conv_out_size(H, kh, stride, pad, dil),
                                                                         conv_out_size(W, kw, stride, pad, dil))


def conv2d_direct(x, w, bias=None, stride=1, pad=0, dil=1, groups=1):
    _nchw32(x, w, bias)
    d, oshape = _dconv_desc(x.shape, w.shape, stride, pad, dil, groups)
    y = torch.empty(oshape, dtype=torch.float32, device=x.device)
    check(_lib.lib().kd_conv2d_direct_fwd(C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), stream_ptr()), "kd_conv2d_direct_fwd")
    return y


def conv2d_direct_dgrad(dy, w, x_shape, stride=1, pad=0, dil=1, groups=1):
    _nchw32(dy, w)
    d, oshape = _dconv_desc(x_shape, w.shape, stride, pad, dil, groups)
    if tuple(dy.shape) != oshape:
        raise ValueError(f"direct dgrad: dy {tuple(dy.shape)} != {oshape}")
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    check(_lib.lib().kd_conv2d_direct_dgrad(C.byref(d), _ptr(dy), _ptr(w), _ptr(dx), stream_ptr()), "kd_conv2d_direct_dgrad")
    return dx


def conv2d_direct_wgrad(x, dy, w_shape, stride=1, pad=0, dil=1, groups=1, want_bias=False):
    _nchw32(x, dy)
    d, oshape = _dconv_desc(x.shape, w_shape, stride, pad, dil, groups)
    if tuple(dy.shape) != oshape:
        raise ValueError(f"direct wgrad: dy {tuple(dy.shape)} != {oshape}")
    dw = torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
    db = torch.empty(w_shape[0], dtype=torch.float32, device=x.device) if want_bias else None
    check(_lib.lib().kd_conv2d_direct_wgrad(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(db), 0, stream_ptr()), "kd_conv2d_direct_wgrad")
    return dw, db


def bn2d_fwd(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu=False):
    """Returns (y, save_mean, save_invstd); updates the running statistics in place when training."""
    _nchw32(x, gamma, beta, running_mean, running_var)
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    y = torch.empty_like(x)
    mean, invstd = torch.empty(Cc, device=x.device), torch.empty(Cc, device=x.device)
    check(_lib.lib().kd_bn2d_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(running_mean),
                                 _ptr(running_var), C.c_float(momentum), C.c_float(eps), int(training), int(relu), N, Cc, HW,
                                 stream_ptr()), "kd_bn2d_fwd")
    return y, mean, invstd


def bn2d_bwd(dy, x, y, gamma, mean, invstd, training, relu=False, need_dx=True):
    """Returns (dx | None, dgamma, dbeta)."""
    _nchw32(dy, x, y, gamma, mean, invstd)
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    dx = torch.empty_like(x) if need_dx else None
    dg, db = torch.empty(Cc, device=x.device), torch.empty(Cc, device=x.device)
    check(_lib.lib().kd_bn2d_bwd(_ptr(dy), _ptr(x), _ptr(y), _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(dx), _ptr(dg), _ptr(db),
                                 int(training), int(relu), 0, N, Cc, HW, stream_ptr()), "kd_bn2d_bwd")
    return dx, dg, db


# ------------------------------------------------------------------------- Gated-SCNN shape stream
def pointwise_small(x, w, bias=None, out=None):
    """1x1 conv + bias, bf16, (Cin, Cout) in {(64, 32), (32, 16), (16, 8)}: x (N,H,W,Cin) view, w fp32 (Cout,Cin[,1,1]) (gscnn.py:232-235)."""
    _need_cuda(x, w, bias, out)
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    w = w.detach().reshape(Cout, -1)
    if x.dtype != torch.bfloat16 or w.shape[1] != Cin or w.dtype != torch.float32 or not w.is_contiguous():
        raise ValueError("pointwise_small: bf16 (N,H,W,Cin) input and a contiguous fp32 (Cout,Cin) weight required")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != Cout or not bias.is_contiguous()):
        raise ValueError("pointwise_small: bias must be contiguous fp32 (Cout,)")
    if out is None:
        out = torch.empty((N, H, W, Cout), dtype=x.dtype, device=x.device)
    if tuple(out.shape) != (N, H, W, Cout) or out.dtype != x.dtype:
        raise ValueError("pointwise_small: bad output view")
    check(_lib.lib().kd_pointwise_small(_ptr(x), nhwc_ld(x), _ptr(w), _ptr(bias), _ptr(out), nhwc_ld(out), N * H * W, Cin, Cout,
                                        stream_ptr()), "kd_pointwise_small")
    return out


def conv3x3_small(x, w_packed, bias=None, res=None, relu=True, out=None):
    """3x3 / stride 1 / pad 1 conv on 16, 32 or 64 channels (bf16): x (N,H,W,C) view, w_packed (C,3,3,C) bf16, bias fp32 (C),
    res (N,H,W,C) view added before the ReLU (Resnet.py:64-99 BasicBlock)."""
    _need_cuda(x, w_packed, bias, res, out)
    N, H, W, Cc = x.shape
    if x.dtype != torch.bfloat16 or tuple(w_packed.shape) != (Cc, 3, 3, Cc) or w_packed.dtype != torch.bfloat16 or not w_packed.is_contiguous():
        raise ValueError("conv3x3_small: bf16 (N,H,W,C) input and a contiguous bf16 (C,3,3,C) packed weight required")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != Cc or not bias.is_contiguous()):
        raise ValueError("conv3x3_small: bias must be contiguous fp32 (C,)")
    if res is not None and (tuple(res.shape) != (N, H, W, Cc) or res.dtype != x.dtype):
        raise ValueError("conv3x3_small: res must match the output shape/dtype")
    if out is None:
        out = torch.empty((N, H, W, Cc), dtype=x.dtype, device=x.device)
    if tuple(out.shape) != (N, H, W, Cc) or out.dtype != x.dtype:
        raise ValueError("conv3x3_small: bad output view")
    check(_lib.lib().kd_conv3x3_small(_ptr(x), nhwc_ld(x), _ptr(w_packed), _ptr(bias), _ptr(res), nhwc_ld(res) if res is not None else 0,
                                      _ptr(out), nhwc_ld(out), N, H, W, Cc, int(bool(relu)), stream_ptr()), "kd_conv3x3_small")
    return out


def gated_conv(feat, gate, params, C_, out=None):
    """feat (N,H,W,>=C) view whose first C channels are the features; gate (N,H,W,1); params: packed fp32 vector
    (W1, b1, w2, b2, Wg with the BNs folded); out: (N,H,W,C) view."""
    _need_cuda(feat, gate, params, out)
    N, H, W, _ = feat.shape
    f = feat[..., :C_]
    if out is None:
        out = torch.empty((N, H, W, C_), dtype=feat.dtype, device=feat.device)
    need = (C_ + 1) * (C_ + 1) + 2 * (C_ + 1) + 1 + C_ * C_
    if params.dtype != torch.float32 or params.numel() != need or not params.is_contiguous() or gate.dtype != feat.dtype:
        raise ValueError("gated_conv: bad parameter vector / gate dtype")
    check(_lib.lib().kd_gated_conv(dt_of(feat), _ptr(f), nhwc_ld(f), _ptr(gate), gate.stride(2), _ptr(params), _ptr(out), nhwc_ld(out),
                                   N * H * W, C_, stream_ptr()), "kd_gated_conv")
    return out


def edge_attention(cs, canny, weights):
    """cs (N,H,W,>=8) view, canny (N,H,W) fp32 0/255, weights fp32 (10,) = fuse[8] + cw[2] -> acts (N,H,W) fp32."""
    _need_cuda(cs, canny, weights)
    N, H, W, _ = cs.shape
    c8 = cs[..., :8]
    acts = torch.empty((N, H, W), dtype=torch.float32, device=cs.device)
    if canny.dtype != torch.float32 or not canny.is_contiguous() or tuple(canny.shape) != (N, H, W) or weights.numel() != 10:
        raise ValueError("edge_attention: bad operands")
    check(_lib.lib().kd_edge_attention(dt_of(cs), _ptr(c8), nhwc_ld(c8), _ptr(canny), _ptr(weights), _ptr(acts), N * H * W, stream_ptr()),
          "kd_edge_attention")
    return acts


def edge_aspp(acts, w, scale, shift, out):
    """acts (N,H,W) fp32 -> out (N,Ho,Wo,C) slice: relu(bilinear(acts) * w[c] * scale[c] + shift[c])."""
    _need_cuda(acts, w, scale, shift, out)
    N, H, W = acts.shape
    _, Ho, Wo, Cc = out.shape
    check(_lib.lib().kd_edge_aspp(dt_of(out), _ptr(acts), H, W, _ptr(w), _ptr(scale), _ptr(shift), _ptr(out), nhwc_ld(out), N, Ho, Wo, Cc,
                                  stream_ptr()), "kd_edge_aspp")
    return out


def canny(x_nchw, low=10, high=100, sweeps=8, max_rounds=64):
    """Device Canny of the uint8-cast batch (N,3,H,W) -> (N,H,W) fp32 0/255.  Hysteresis runs `sweeps` sweeps per round and
    reads one device int between rounds (the reference round-trips the whole image through the host here)."""
    _need_cuda(x_nchw)
    if x_nchw.dtype != torch.float32 or not x_nchw.is_contiguous() or x_nchw.shape[1] != 3:
        raise ValueError("canny: expects a contiguous fp32 (N,3,H,W) batch")
    N, _, H, W = x_nchw.shape
    out = torch.empty((N, H, W), dtype=torch.float32, device=x_nchw.device)
    changed = torch.zeros(1, dtype=torch.int32, device=x_nchw.device)
    need = _lib.lib().kd_canny_workspace(N, H, W)
    ws = _ws(need, x_nchw.device)
    check(_lib.lib().kd_canny(_ptr(x_nchw), N, H, W, low, high, sweeps, _ptr(out), _ptr(changed), _ptr(ws), need, stream_ptr()), "kd_canny")
    for _ in range(max_rounds):
        if int(changed.item()) == 0:
            break
        check(_lib.lib().kd_canny_continue(N, H, W, sweeps, _ptr(out), _ptr(changed), _ptr(ws), need, stream_ptr()), "kd_canny_continue")
    return out


# ------------------------------------------------------------------------- Gated-SCNN shape stream, backward pieces
def _rows(t):
    """(..., C) tensor with dense channels and uniformly strided pixels -> (pixel stride, pixels, channels)."""
    if t.stride(-1) != 1:
        raise ValueError("expected dense channels")
    Cc = t.shape[-1]
    npix = t.numel() // Cc
    ld = t.stride(-2) if t.dim() > 1 else Cc
    lead = t.shape[:-1]
    exp = ld
    for n, st in zip(reversed(lead), reversed(t.stride()[:-1])):   # every leading dimension must continue the pixel stride
        if n > 1 and st != exp:
            raise ValueError(f"not a pixel-strided view: shape {tuple(t.shape)} stride {t.stride()}")
        exp *= n
    return ld, npix, Cc


def small_linear(x, w, bias=None, out=None, out_dtype=None, accumulate=False, relu=False, mask=None):
    """out[p][co] (+)= bias[co] + sum_ci w[co][ci] x[p][ci]; x (..., Cin), w fp32 (Cout, Cin), <= 72 channels on either side.
    mask: fp32 (..., Cout), the result is zeroed where mask <= 0 (backward through the ReLU that produced `mask`)."""
    _need_cuda(x, w, bias, out, mask)
    ldx, npix, Cin = _rows(x)
    w = w.detach().float().contiguous()
    Cout = w.shape[0]
    if w.dim() != 2 or w.shape[1] != Cin:
        raise ValueError(f"small_linear: w {tuple(w.shape)} does not match {Cin} input channels")
    if bias is not None:
        bias = bias.detach().float().contiguous()
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (Cout,), dtype=out_dtype or x.dtype, device=x.device)
    ldy, npo, Co = _rows(out)
    if npo != npix or Co != Cout:
        raise ValueError("small_linear: bad output view")
    ldm = 0
    if mask is not None:
        ldm, npm, Cm = _rows(mask)
        if mask.dtype != torch.float32 or npm != npix or Cm != Cout:
            raise ValueError("small_linear: mask must be fp32 and match the output")
    check(_lib.lib().kd_small_linear(_ptr(x), dt_of(x), ldx, Cin, _ptr(w), _ptr(bias), _ptr(out), dt_of(out), ldy, Cout, npix, int(accumulate),
                                     int(relu), _ptr(mask), ldm, stream_ptr()), "kd_small_linear")
    return out


def small_wgrad(a, b, want_bias=False, dw=None, db=None, accumulate=False):
    """dw[cb][ca] = sum_p b[p][cb] a[p][ca] (fp32 (Cb, Ca)), db[cb] = sum_p b[p][cb]; a (..., Ca), b (..., Cb), <= 72 channels."""
    _need_cuda(a, b, dw, db)
    lda, npix, Ca = _rows(a)
    ldb, npb, Cb = _rows(b)
    if npb != npix:
        raise ValueError("small_wgrad: a and b must cover the same pixels")
    if dw is None:
        dw = torch.empty((Cb, Ca), dtype=torch.float32, device=a.device)
    if want_bias and db is None:
        db = torch.empty((Cb,), dtype=torch.float32, device=a.device)
    need = _lib.lib().kd_small_wgrad_workspace(Ca, Cb, npix)
    ws = _ws(need, a.device)
    check(_lib.lib().kd_small_wgrad(_ptr(a), dt_of(a), lda, Ca, _ptr(b), dt_of(b), ldb, Cb, npix, _ptr(dw), _ptr(db), int(accumulate), _ptr(ws),
                                    need, stream_ptr()), "kd_small_wgrad")
    return dw, db


def gate_mix_bwd(feat, a, gv=None, want_v=False):
    """GatedSpatialConv2d's tail: returns (gfeat, ga, v) -- gfeat = gv (sigmoid(a) + 1), ga = (sum_c gv feat) sigmoid'(a) (both None
    without gv), v = feat (sigmoid(a) + 1) (if want_v); fp32 outputs."""
    _need_cuda(feat, a, gv)
    ldf, npix, Cc = _rows(feat)
    if a.dtype != torch.float32 or a.numel() != npix or not a.is_contiguous():
        raise ValueError("gate_mix_bwd: a must be a contiguous fp32 tensor with one value per pixel")
    gfeat = ga = v = None
    ldgv = 0
    if gv is not None:
        if gv.dtype != torch.float32:
            raise ValueError("gate_mix_bwd: gv must be fp32")
        ldgv, npg, Cg = _rows(gv)
        if npg != npix or Cg != Cc:
            raise ValueError("gate_mix_bwd: gv must match feat")
        gfeat = torch.empty(tuple(feat.shape), dtype=torch.float32, device=feat.device)
        ga = torch.empty(tuple(feat.shape[:-1]), dtype=torch.float32, device=feat.device)
    if want_v:
        v = torch.empty(tuple(feat.shape), dtype=torch.float32, device=feat.device)
    check(_lib.lib().kd_gate_mix_bwd(_ptr(feat), dt_of(feat), ldf, _ptr(a), _ptr(gv), ldgv, _ptr(gfeat), Cc, _ptr(ga), _ptr(v), Cc, Cc, npix,
                                     stream_ptr()), "kd_gate_mix_bwd")
    return gfeat, ga, v


def edge_attention_bwd(cs, canny, weights, g_acts):
    """Backward of edge_attention: returns (g_t, g_s, eo_canny) -- gradients w.r.t. the cw and fuse pre-activations (N,H,W) fp32 and the
    cw conv's input [sigmoid(fuse . cs), canny] (N,H,W,2) fp32."""
    _need_cuda(cs, canny, weights, g_acts)
    c8 = cs[..., :8]
    ldc, npix, _ = _rows(c8)
    for t in (canny, g_acts):
        if t.dtype != torch.float32 or t.numel() != npix or not t.is_contiguous():
            raise ValueError("edge_attention_bwd: canny / g_acts must be contiguous fp32 (N,H,W)")
    g_t, g_s = torch.empty_like(g_acts), torch.empty_like(g_acts)
    eo_canny = torch.empty(tuple(g_acts.shape) + (2,), dtype=torch.float32, device=cs.device)
    check(_lib.lib().kd_edge_attention_bwd(dt_of(cs), _ptr(c8), ldc, _ptr(canny), _ptr(weights), _ptr(g_acts), _ptr(g_t), _ptr(g_s), _ptr(eo_canny),
                                           npix, stream_ptr()), "kd_edge_attention_bwd")
    return g_t, g_s, eo_canny


def rank1_add(y, g, w, accumulate=True):
    """y[p][c] (+)= g[p] * w[c]; y (..., C) view, g fp32 one value per pixel, w fp32 (C,)."""
    _need_cuda(y, g, w)
    ldy, npix, Cc = _rows(y)
    w = w.detach().float().contiguous()
    if g.dtype != torch.float32 or g.numel() != npix or not g.is_contiguous() or w.numel() != Cc:
        raise ValueError("rank1_add: g must be contiguous fp32 with one value per pixel, w (C,)")
    check(_lib.lib().kd_rank1_add(dt_of(y), _ptr(y), ldy, _ptr(g), _ptr(w), Cc, npix, int(accumulate), stream_ptr()), "kd_rank1_add")
    return y


# ------------------------------------------------------------------------------------- losses
def view3(t):
    """(N,C,*spatial) logical tensor (any of NCHW / channels_last / (N,C)) -> kd_view3 + (N,C,P)."""
    mat = getattr(t, "materialize", None)     # lazy.LazyLogits: a raw pointer is wanted, so the tensor has to exist
    if mat is not None:
        t = mat()
    if t.dim() == 2:
        N, Cc = t.shape
        return View3(t.data_ptr(), dt_of(t), t.stride(0), t.stride(1), 0), (N, Cc, 1)
    if t.dim() != 4:
        raise ValueError("loss operands must be (N,C) or (N,C,H,W)")
    N, Cc, H, W = t.shape
    sN, sC, sH, sW = t.stride()
    if H > 1 and sH != W * sW:
        raise ValueError(f"loss operand rows are not uniformly strided: stride {t.stride()}")
    return View3(t.data_ptr(), dt_of(t), sN, sC, sW), (N, Cc, H * W)


_ws_cache = {}


def loss_workspace(N, Cc, P, device):
    need = _lib.lib().kd_loss_workspace(N, Cc, P)
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws, need


def _loss_common(s, t):
    _need_cuda(s, t)
    if s.shape != t.shape:
        raise ValueError(f"loss: inputs {tuple(s.shape)} vs targets {tuple(t.shape)}")
    vs, dims = view3(s)
    vt, _ = view3(t)
    return vs, vt, dims


def kldiv(s, t, temperature=1.0, want_grad=True, grad_scale=1.0):
    vs, vt, (N, Cc, P) = _loss_common(s, t)
    loss = torch.empty((), dtype=torch.float32, device=s.device)
    grad = torch.empty_like(s) if want_grad else None
    vg = view3(grad)[0] if want_grad else None
    ws, need = loss_workspace(N, Cc, P, s.device)
    e0 = _prof_start()
    check(_lib.lib().kd_kldiv(C.byref(vs), C.byref(vt), C.c_float(temperature), N, Cc, P, _ptr(loss),
                              C.byref(vg) if vg is not None else None, C.c_float(grad_scale), _ptr(ws), need, stream_ptr()),
          "kd_kldiv")
    _prof_stop(e0, "loss", _nbytes(s, t, grad), f"kldiv {N}x{Cc}x{P}", "kldiv_kernel")
    return loss, grad


def hint_mse(s, t, num_classes=19, want_grad=True, grad_scale=1.0):
    vs, vt, (N, Cc, P) = _loss_common(s, t)
    loss = torch.empty((), dtype=torch.float32, device=s.device)
    grad = torch.empty_like(s) if want_grad else None
    vg = view3(grad)[0] if want_grad else None
    ws, need = loss_workspace(N, Cc, P, s.device)
    e0 = _prof_start()
    check(_lib.lib().kd_hint_mse(C.byref(vs), C.byref(vt), C.c_float(num_classes), N, Cc, P, _ptr(loss),
                                 C.byref(vg) if vg is not None else None, C.c_float(grad_scale), _ptr(ws), need, stream_ptr()),
          "kd_hint_mse")
    _prof_stop(e0, "loss", _nbytes(s, t, grad), f"hint mse {N}x{Cc}x{P}", "mse_vec_kernel")
    return loss, grad


def weighted_hint_mse(s, t, w, want_grad=True, grad_scale=1.0):
    vs, vt, (N, Cc, P) = _loss_common(s, t)
    _need_cuda(w)
    w = w.detach().float().contiguous()
    if tuple(w.shape) not in ((Cc,), (N, Cc)):
        raise ValueError(f"weighted_hint_mse: filter_weight must be (C,) or (N,C), got {tuple(w.shape)}")
    loss = torch.empty((), dtype=torch.float32, device=s.device)
    grad = torch.empty_like(s) if want_grad else None
    vg = view3(grad)[0] if want_grad else None
    ws, need = loss_workspace(N, Cc, P, s.device)
    e0 = _prof_start()
    check(_lib.lib().kd_weighted_hint_mse(C.byref(vs), C.byref(vt), _ptr(w), int(w.dim() == 2), N, Cc, P, _ptr(loss),
                                          C.byref(vg) if vg is not None else None, C.c_float(grad_scale), _ptr(ws), need,
                                          stream_ptr()), "kd_weighted_hint_mse")
    _prof_stop(e0, "loss", _nbytes(s, t, grad), f"weighted hint mse {N}x{Cc}x{P}", "whmse_kernel")
    return loss, grad


def _class_weight(weight, Cc, device):
    w = weight.detach().to(device=device, dtype=torch.float32).contiguous()
    if w.numel() != Cc:
        raise ValueError("class weights: one per class")
    return w


def ce2d(x, target, ignore_index=255, weight=None, size_average=True):
    _need_cuda(x, target)
    vx, (N, Cc, P) = view3(x)
    tgt = target.contiguous()
    if tgt.dtype != torch.int64 or tgt.numel() != N * P:
        raise ValueError("ce2d: target must be int64 (N,H,W)")
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    ws, need = loss_workspace(N, Cc, P, x.device)
    e0 = _prof_start()
    if weight is not None or not size_average:
        w = _class_weight(weight, Cc, x.device) if weight is not None else None
        check(_lib.lib().kd_ce2d_weighted(C.byref(vx), _ptr(tgt), _ptr(w) if w is not None else None, int(not size_average), ignore_index, N, Cc, P,
                                          _ptr(loss), _ptr(ws), need, stream_ptr()), "kd_ce2d_weighted")
    else:
        check(_lib.lib().kd_ce2d(C.byref(vx), _ptr(tgt), ignore_index, N, Cc, P, _ptr(loss), _ptr(ws), need, stream_ptr()), "kd_ce2d")
    _prof_stop(e0, "loss", _nbytes(x, tgt), f"ce2d {N}x{Cc}x{P}", "ce2d_kernel")
    return loss


def _lowres_ok(*lows):
    for t in lows:
        if t.dim() != 4 or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("low-resolution logits must be dense fp32 (N,h,w,C)")


def ce2d_up(x_lo, target, size, ignore_index=255, align_corners=True):
    """ce2d(upsample_bilinear(x_lo, size), target) without the full-resolution tensor (kd_ce2d_up)."""
    _need_cuda(x_lo, target)
    _lowres_ok(x_lo)
    N, h, w, Cc = x_lo.shape
    H, W = size
    tgt = target.contiguous()
    if tgt.dtype != torch.int64 or tgt.numel() != N * H * W:
        raise ValueError("ce2d_up: target must be int64 (N,H,W)")
    loss = torch.empty((), dtype=torch.float32, device=x_lo.device)
    ws, need = loss_workspace(N, Cc, H * W, x_lo.device)
    e0 = _prof_start()
    check(_lib.lib().kd_ce2d_up(_ptr(x_lo), _ptr(tgt), ignore_index, N, h, w, Cc, H, W, int(bool(align_corners)), _ptr(loss), _ptr(ws),
                                need, stream_ptr()), "kd_ce2d_up")
    _prof_stop(e0, "loss", _nbytes(x_lo, tgt), f"ce2d from {h}x{w} logits at {H}x{W}", "ce2d_up_kernel")
    return loss


def kldiv_up(s_lo, t_lo, size, temperature=1.0, align_corners=True):
    """kldiv(upsample_bilinear(s_lo, size), upsample_bilinear(t_lo, size)) forward, without the full-resolution tensors."""
    _need_cuda(s_lo, t_lo)
    _lowres_ok(s_lo, t_lo)
    if s_lo.shape != t_lo.shape:
        raise ValueError("kldiv_up: shape mismatch")
    N, h, w, Cc = s_lo.shape
    H, W = size
    loss = torch.empty((), dtype=torch.float32, device=s_lo.device)
    ws, need = loss_workspace(N, Cc, H * W, s_lo.device)
    e0 = _prof_start()
    check(_lib.lib().kd_kldiv_up(_ptr(s_lo), _ptr(t_lo), C.c_float(temperature), N, h, w, Cc, H, W, int(bool(align_corners)), _ptr(loss),
                                 _ptr(ws), need, stream_ptr()), "kd_kldiv_up")
    _prof_stop(e0, "loss", _nbytes(s_lo, t_lo), f"kldiv from {h}x{w} logits at {H}x{W}", "kldiv_up_kernel")
    return loss


def ce2d_grad(x, target, ignore_index=255, grad_scale=1.0, weight=None, size_average=True):
    """d ce2d / d x, same layout as x."""
    _need_cuda(x, target)
    vx, (N, Cc, P) = view3(x)
    tgt = target.contiguous()
    if tgt.dtype != torch.int64 or tgt.numel() != N * P:
        raise ValueError("ce2d_grad: target must be int64 with one label per pixel")
    grad = torch.empty_like(x)
    vg, _ = view3(grad)
    ws, need = loss_workspace(N, Cc, P, x.device)
    if weight is not None or not size_average:
        w = _class_weight(weight, Cc, x.device) if weight is not None else None
        check(_lib.lib().kd_ce2d_weighted_grad(C.byref(vx), _ptr(tgt), _ptr(w) if w is not None else None, int(not size_average), ignore_index, N, Cc, P,
                                               C.byref(vg), C.c_float(grad_scale), _ptr(ws), need, stream_ptr()), "kd_ce2d_weighted_grad")
        return grad
    check(_lib.lib().kd_ce2d_grad(C.byref(vx), _ptr(tgt), ignore_index, N, Cc, P, C.byref(vg), C.c_float(grad_scale), _ptr(ws), need,
                                  stream_ptr()), "kd_ce2d_grad")
    return grad


def confusion(x, target, conf=None, accumulate=False):
    """conf (C,C) int64 [label][prediction] of argmax_c x vs target; pixels whose label is outside [0, C) are skipped."""
    _need_cuda(x, target, conf)
    vx, (N, Cc, P) = view3(x)
    tgt = target.contiguous()
    if tgt.dtype != torch.int64 or tgt.numel() != N * P:
        raise ValueError("confusion: target must be int64 with one label per pixel")
    if conf is None:
        conf, accumulate = torch.empty((Cc, Cc), dtype=torch.int64, device=x.device), False
    if conf.dtype != torch.int64 or tuple(conf.shape) != (Cc, Cc) or not conf.is_contiguous():
        raise ValueError("confusion: conf must be a contiguous int64 (C, C) tensor")
    check(_lib.lib().kd_confusion(C.byref(vx), _ptr(tgt), N, Cc, P, _ptr(conf), int(accumulate), stream_ptr()), "kd_confusion")
    return conf


def radam_step_multi(items):
    """items: [(p, g, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay), ...] -- kd_radam_step for all of them in one
    launch per 48 tensors; same arithmetic per element as radam_step."""
    if not items:
        return
    arr = (_lib.RadamTensor * len(items))()
    for i, (p, g, m, v, step, lr, beta1, beta2, eps, wd) in enumerate(items):
        _need_cuda(p, g, m, v)
        for t in (p, g, m, v):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
                raise ValueError("radam_step_multi: fp32 contiguous tensors of equal size required")
        a = arr[i]
        a.p, a.g, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
        a.n, a.step = p.numel(), int(step)
        a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = lr, beta1, beta2, eps, wd
    check(_lib.lib().kd_radam_step_multi(arr, len(items), stream_ptr()), "kd_radam_step_multi")
    # the kernel wrote through raw pointers: tell autograd / version-keyed caches (engine weight packs) about it
    for (p, _, m, v, *_rest) in items:
        for t in (p, m, v):
            torch.autograd.graph.increment_version(t)


def radam_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    _need_cuda(p, g, m, v)
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError("radam_step: fp32 contiguous tensors of equal size required")
    check(_lib.lib().kd_radam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), int(step), C.c_float(lr), C.c_float(beta1),
                                   C.c_float(beta2), C.c_float(eps), C.c_float(weight_decay), stream_ptr()), "kd_radam_step")
    # the kernel wrote through raw pointers: tell autograd / version-keyed caches (engine weight packs) about it
    for t in (p, m, v):
        torch.autograd.graph.increment_version(t)
