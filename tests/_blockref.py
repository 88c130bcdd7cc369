"""Block-level CPU restatements built from oracle/oracle.c ops (test infrastructure): one IdentityResidualBlock
(reference models/encoders/wider_resnet.py:119-182, eval mode) forward + input gradient, and the ASPP module
(models/deeplabv3/deeplabv3.py:21-75).  Weights come from tests/_seeded.py by checkpoint key, the same fill
tools/make_golden.py applied to the reference's modules, so only inputs/outputs live in tests/golden/."""
import numpy as np
import torch

from _seeded import seeded_value
from oracle import oracle as orc

RESBLOCK_CASES = {  # tag -> (cin, channels, stride, dilation); mirrors tools/make_golden.py:g_resblock
    "id2_d1": (16, (16, 16), 1, 1),
    "proj2_s2": (8, (16, 16), 2, 1),
    "proj2_d2": (16, (8, 32), 1, 2),
    "bott_d4": (16, (8, 16, 32), 1, 4),
}


def _sv(key, shape):
    return seeded_value(key, torch.empty(shape)).numpy()


def resblock_params(tag):
    """{state-dict name: array} of the reference block `tag` (names as in IdentityResidualBlock.state_dict())."""
    cin, ch, stride, dil = RESBLOCK_CASES[tag]
    pre = f"resblock.{tag}."
    P = {}

    def bn(name, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            P[f"{name}.0.{s}"] = _sv(pre + f"{name}.0.{s}", (c,))
    bn("bn1", cin)
    ks = [3, 3] if len(ch) == 2 else [1, 3, 1]
    cins = [cin] + list(ch[:-1])
    for i, (ci, co, k) in enumerate(zip(cins, ch, ks)):
        if i > 0:
            bn(f"convs.bn{i + 1}", ci)
        P[f"convs.conv{i + 1}.weight"] = _sv(pre + f"convs.conv{i + 1}.weight", (co, ci, k, k))
    if stride != 1 or cin != ch[-1]:
        P["proj_conv.weight"] = _sv(pre + "proj_conv.weight", (ch[-1], cin, 1, 1))
    return P


def _bn(P, name, x, relu=True):
    return orc.bn_eval(x, P[name + ".weight"], P[name + ".bias"], P[name + ".running_mean"], P[name + ".running_var"],
                       relu=relu)


def resblock_fwd_bwd(tag, x, gy):
    """Returns (y, gx): the block output and d<y, gy>/dx, eval-mode BN (wider_resnet.py:169-182)."""
    cin, ch, stride, dil = RESBLOCK_CASES[tag]
    P = resblock_params(tag)
    ks = [3, 3] if len(ch) == 2 else [1, 3, 1]
    a = _bn(P, "bn1.0", x)
    proj = "proj_conv.weight" in P
    shortcut = orc.conv2d_fwd(a, P["proj_conv.weight"], stride=stride) if proj else x
    acts, t = [a], a
    geoms = []
    for i, k in enumerate(ks):
        if i > 0:
            t = _bn(P, f"convs.bn{i + 1}.0", t)
            acts.append(t)
        s = stride if i == 0 else 1
        pad, d = (dil, dil) if k == 3 else (0, 1)
        geoms.append((s, pad, d))
        t = orc.conv2d_fwd(t, P[f"convs.conv{i + 1}.weight"], stride=s, pad=pad, dil=d)
    y = t + shortcut
    # backward
    g = gy
    for i in range(len(ks) - 1, -1, -1):
        s, pad, d = geoms[i]
        g = orc.conv2d_dgrad(g, P[f"convs.conv{i + 1}.weight"], acts[i].shape, stride=s, pad=pad, dil=d)
        if i > 0:
            nm = f"convs.bn{i + 1}.0"
            g = orc.bn_eval_bwd(g, acts[i], P[nm + ".weight"], P[nm + ".running_var"], relu=True)
    if proj:
        g = g + orc.conv2d_dgrad(gy, P["proj_conv.weight"], a.shape, stride=stride)
    gx = orc.bn_eval_bwd(g, a, P["bn1.0.weight"], P["bn1.0.running_var"], relu=True)
    if not proj:
        gx = gx + gy
    return y, gx


def aspp_params(in_dim=32, red=16):
    P = {}
    for i, k in enumerate([1, 3, 3, 3]):
        P[f"features.{i}.0.weight"] = _sv(f"aspp.features.{i}.0.weight", (red, in_dim, k, k))
        for s in ("weight", "bias", "running_mean", "running_var"):
            P[f"features.{i}.1.{s}"] = _sv(f"aspp.features.{i}.1.{s}", (red,))
    P["img_conv.0.weight"] = _sv("aspp.img_conv.0.weight", (red, in_dim, 1, 1))
    for s in ("weight", "bias", "running_mean", "running_var"):
        P[f"img_conv.1.{s}"] = _sv(f"aspp.img_conv.1.{s}", (red,))
    return P


def aspp_fwd(x, in_dim=32, red=16, rates=(12, 24, 36)):
    """cat[upsample(img_conv(avgpool(x))), f0(x), .., f3(x)] (deeplabv3.py:64-75), output_stride 8 -> rates doubled already."""
    P = aspp_params(in_dim, red)
    N, _, H, W = x.shape
    pooled = orc.gap(x).reshape(N, in_dim, 1, 1)
    img = _bn(P, "img_conv.1", orc.conv2d_fwd(pooled, P["img_conv.0.weight"]))
    outs = [np.broadcast_to(img, (N, red, H, W))]   # bilinear upsample of a 1x1 map = broadcast
    for i, r in enumerate((None,) + tuple(rates)):
        y = orc.conv2d_fwd(x, P[f"features.{i}.0.weight"], pad=0 if r is None else r, dil=1 if r is None else r)
        outs.append(_bn(P, f"features.{i}.1", y))
    return np.concatenate(outs, 1)
