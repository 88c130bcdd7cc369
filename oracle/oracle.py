"""ctypes/numpy front-end of oracle/oracle.c.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.  All arrays are
float32 NCHW like the reference (PyTorch) tensors the restatement mirrors.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i64 = ctypes.POINTER(ctypes.c_int64)


def build():
    """Compile oracle.c with gcc (idempotent)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_kldiv.restype = ctypes.c_double
        _lib.orc_mse.restype = ctypes.c_double
        _lib.orc_whmse.restype = ctypes.c_double
        _lib.orc_ce2d.restype = ctypes.c_double
        _lib.orc_mse_sum.restype = ctypes.c_double
        _lib.orc_ce2d_w.restype = ctypes.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_f)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def conv_out(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv2d_fwd(x, w, bias=None, stride=1, pad=0, dil=1, groups=1):
    x, w = _c(x), _c(w)
    N, C, H, W = x.shape
    K, _, kh, kw = w.shape
    y = np.empty((N, K, conv_out(H, kh, stride, pad, dil), conv_out(W, kw, stride, pad, dil)), np.float32)
    b = None if bias is None else _c(bias)
    lib().orc_conv2d_fwd(_p(x), _p(w), _p(b), _p(y), N, C, H, W, K, kh, kw, stride, pad, dil, groups)
    return y


def conv2d_dgrad(dy, w, x_shape, stride=1, pad=0, dil=1, groups=1):
    dy, w = _c(dy), _c(w)
    N, C, H, W = x_shape
    K, _, kh, kw = w.shape
    dx = np.empty(x_shape, np.float32)
    lib().orc_conv2d_dgrad(_p(dy), _p(w), _p(dx), N, C, H, W, K, kh, kw, stride, pad, dil, groups)
    return dx


def conv2d_wgrad(x, dy, w_shape, stride=1, pad=0, dil=1, groups=1):
    x, dy = _c(x), _c(dy)
    N, C, H, W = x.shape
    K, _, kh, kw = w_shape
    dw = np.empty(w_shape, np.float32)
    lib().orc_conv2d_wgrad(_p(x), _p(dy), _p(dw), N, C, H, W, K, kh, kw, stride, pad, dil, groups)
    return dw


def bn_eval(x, gamma, beta, mean, var, eps=1e-5, relu=False):
    x = _c(x)
    N, C = x.shape[:2]
    y = np.empty_like(x)
    lib().orc_bn_eval(_p(x), _p(y), _p(_c(gamma)), _p(_c(beta)), _p(_c(mean)), _p(_c(var)),
                      ctypes.c_float(eps), int(relu), N, C, int(np.prod(x.shape[2:])))
    return y


def bn_eval_bwd(dy, y, gamma, var, eps=1e-5, relu=False):
    dy, y = _c(dy), _c(y)
    N, C = y.shape[:2]
    dx = np.empty_like(dy)
    lib().orc_bn_eval_bwd(_p(dy), _p(y), _p(dx), _p(_c(gamma)), _p(_c(var)), ctypes.c_float(eps), int(relu),
                          N, C, int(np.prod(y.shape[2:])))
    return dx


def bn_train_fwd(x, gamma, beta, eps=1e-5, relu=False):
    x = _c(x)
    N, C = x.shape[:2]
    y = np.empty_like(x)
    mean = np.empty(C, np.float32)
    invstd = np.empty(C, np.float32)
    lib().orc_bn_train_fwd(_p(x), _p(y), _p(_c(gamma)), _p(_c(beta)), _p(mean), _p(invstd),
                           ctypes.c_float(eps), int(relu), N, C, int(np.prod(x.shape[2:])))
    return y, mean, invstd


def bn_train_bwd(dy, x, y, gamma, mean, invstd, relu=False):
    dy, x, y = _c(dy), _c(x), _c(y)
    N, C = x.shape[:2]
    dx = np.empty_like(x)
    dgamma = np.empty(C, np.float32)
    dbeta = np.empty(C, np.float32)
    lib().orc_bn_train_bwd(_p(dy), _p(x), _p(y), _p(dx), _p(dgamma), _p(dbeta), _p(_c(gamma)), _p(_c(mean)),
                           _p(_c(invstd)), int(relu), N, C, int(np.prod(x.shape[2:])))
    return dx, dgamma, dbeta


def maxpool3x3s2(x):
    x = _c(x)
    N, C, H, W = x.shape
    y = np.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), np.float32)
    lib().orc_maxpool3x3s2(_p(x), _p(y), N, C, H, W)
    return y


def upsample_bilinear_ac(x, size):
    x = _c(x)
    N, C, H, W = x.shape
    y = np.empty((N, C, size[0], size[1]), np.float32)
    lib().orc_upsample_bilinear_ac(_p(x), _p(y), N, C, H, W, size[0], size[1])
    return y


def gap(x):
    x = _c(x)
    N, C = x.shape[:2]
    y = np.empty((N, C), np.float32)
    lib().orc_gap(_p(x), _p(y), N, C, int(np.prod(x.shape[2:])))
    return y


def _nchw3(a):
    a = _c(a)
    N, C = a.shape[:2]
    return a, N, C, int(np.prod(a.shape[2:])) if a.ndim > 2 else 1


def kldiv(s, t, T=1.0, want_grad=True):
    s, N, C, HW = _nchw3(s)
    t = _c(t)
    g = np.empty_like(s) if want_grad else None
    loss = lib().orc_kldiv(_p(s), _p(t), _p(g), ctypes.c_double(T), N, C, HW)
    return loss, g


def mse(s, t, num_classes=19, want_grad=True):
    s, t = _c(s), _c(t)
    g = np.empty_like(s) if want_grad else None
    loss = lib().orc_mse(_p(s), _p(t), _p(g), ctypes.c_double(num_classes), ctypes.c_size_t(s.size))
    return loss, g


def mse_sum(s, t, num_classes=19, want_grad=True):
    """MSELoss(reduction='sum', num_classes)."""
    s, t = _c(s), _c(t)
    g = np.empty_like(s) if want_grad else None
    loss = lib().orc_mse_sum(_p(s), _p(t), _p(g), ctypes.c_double(num_classes), ctypes.c_size_t(s.size))
    return loss, g


def whmse(s, t, w, want_grad=True):
    s, N, C, HW = _nchw3(s)
    t, w = _c(t), _c(w)
    g = np.empty_like(s) if want_grad else None
    loss = lib().orc_whmse(_p(s), _p(t), _p(w), int(w.ndim == 2), _p(g), N, C, HW)
    return loss, g


def ce2d(x, target, ignore_index=255):
    x, N, C, HW = _nchw3(x)
    tgt = np.ascontiguousarray(target, dtype=np.int64)
    return lib().orc_ce2d(_p(x), tgt.ctypes.data_as(_i64), ignore_index, N, C, HW)


def ce2d_weighted(x, target, weight=None, size_average=True, ignore_index=255, want_grad=True):
    """CrossEntropyLoss2d(weight, size_average, ignore_index): (loss, grad_x)."""
    x, N, C, HW = _nchw3(x)
    tgt = np.ascontiguousarray(target, dtype=np.int64)
    w = None if weight is None else _c(weight)
    g = np.empty_like(x) if want_grad else None
    loss = lib().orc_ce2d_w(_p(x), tgt.ctypes.data_as(_i64), _p(w), int(bool(size_average)), ignore_index, N, C, HW, _p(g))
    return loss, g


def confusion(x, target, ignore_index=255, conf=None):
    """(C, C) int64 confusion matrix [label][argmax prediction]; accumulates into `conf` when given."""
    x = _c(x)
    N, C = x.shape[:2]
    t = np.ascontiguousarray(target, dtype=np.int64)
    if conf is None:
        conf = np.zeros((C, C), np.int64)
    lib().orc_confusion(_p(x), t.ctypes.data_as(_i64), ctypes.c_int64(ignore_index), N, C, int(np.prod(x.shape[2:])),
                        conf.ctypes.data_as(_i64))
    return conf


def miou(conf):
    """CityscapesMetricTracker.get_iou (utils/util.py:113-118)."""
    conf = np.asarray(conf, np.float64)
    if not conf.any():
        return 1.0
    tp = np.diag(conf)
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.nanmean(tp / (conf.sum(0) + conf.sum(1) - tp)))


def canny_ref(img_u8, low=10, high=100):
    """Canny edge map of an (H, W, 3) uint8 image as cv2.Canny(img, low, high) documents it (aperture 3, L1 gradient):
    Sobel per channel with replicated borders, the channel of largest |gx| + |gy| wins, non-maximum suppression on the
    quantised direction (tan 22.5 / 67.5 degrees in 2^15 fixed point), double threshold, 8-connected hysteresis.
    PARITY UNPINNED: opencv-python is not in this image; this restates the published algorithm and is what
    the device kernel kd_canny is checked against.  Returns (H, W) uint8 with values 0 / 255."""
    img = np.asarray(img_u8, dtype=np.int64)
    H, W, _ = img.shape
    p = np.pad(img, ((1, 1), (1, 1), (0, 0)), mode="edge")
    gx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    gy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    mag3 = np.abs(gx) + np.abs(gy)
    best = np.argmax(mag3, axis=2)                     # first channel of the largest magnitude
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    xs, ys, mag = gx[ii, jj, best], gy[ii, jj, best], mag3[ii, jj, best]
    mp = np.pad(mag, 1)                                # magnitudes outside the image count as 0
    M = lambda dy, dx: mp[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    ax, ay = np.abs(xs), np.abs(ys) << 15
    tg22 = ax * 13573
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = (~horiz) & (ay > tg67)
    s = np.where((xs ^ ys) < 0, -1, 1)
    up = np.where(s == 1, M(-1, -1), M(-1, 1))
    dn = np.where(s == 1, M(1, 1), M(1, -1))
    keep = np.where(horiz, (mag > M(0, -1)) & (mag >= M(0, 1)),
                    np.where(vert, (mag > M(-1, 0)) & (mag >= M(1, 0)), (mag > up) & (mag > dn)))
    keep &= mag > low
    strong = keep & (mag > high)
    weak = keep & ~strong
    while True:
        sp = np.pad(strong, 1)
        near = np.zeros_like(strong)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dy or dx:
                    near |= sp[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
        grow = weak & near
        if not grow.any():
            break
        strong |= grow
        weak &= ~grow
    return (strong * 255).astype(np.uint8)


def radam_step(p, g, m, v, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """In-place on p, m, v (float32 contiguous)."""
    lib().orc_radam_step(_p(p), _p(_c(g)), _p(m), _p(v), ctypes.c_size_t(p.size), int(step),
                         ctypes.c_double(lr), ctypes.c_double(betas[0]), ctypes.c_double(betas[1]),
                         ctypes.c_double(eps), ctypes.c_double(weight_decay))
