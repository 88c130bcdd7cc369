"""Pins oracle/oracle.c against golden vectors produced by the reference itself
(tools/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag,T", [("kld_T1", 1.0), ("kld_T5", 5.0), ("kld2d_T5", 5.0)])
def test_kldiv(golden, tag, T):
    g = golden("losses")
    loss, grad = orc.kldiv(g[f"{tag}.s"], g[f"{tag}.t"], T)
    close(loss, g[f"{tag}.loss"], rtol=1e-5)
    close(grad, g[f"{tag}.grad"], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("tag,nc", [("mse_1000", 1000), ("mse_1", 1)])
def test_mse(golden, tag, nc):
    g = golden("losses")
    loss, grad = orc.mse(g[f"{tag}.s"], g[f"{tag}.t"], nc)
    close(loss, g[f"{tag}.loss"], rtol=1e-5)
    close(grad, g[f"{tag}.grad"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("tag", ["whmse_c", "whmse_nc"])
def test_whmse(golden, tag):
    g = golden("losses")
    loss, grad = orc.whmse(g[f"{tag}.s"], g[f"{tag}.t"], g[f"{tag}.w"])
    close(loss, g[f"{tag}.loss"], rtol=1e-5)
    close(grad, g[f"{tag}.grad"], rtol=1e-4, atol=1e-9)


def test_ce_and_anchors(golden):
    g = golden("losses")
    close(orc.ce2d(g["ce.x"], g["ce.target"]), g["ce.loss"], rtol=1e-5)
    # SURVEY 8c anchors measured on the reference
    close(g["anchor.kld_T1"], 0.915876, rtol=1e-5)
    close(g["anchor.kld_T5"], 0.956884, rtol=1e-5)
    close(g["anchor.mse_1000"], 2022.908, rtol=1e-5)
    close(orc.kldiv(g["anchor.x"], g["anchor.t"], 1.0)[0], 0.915876, rtol=1e-5)
    close(orc.kldiv(g["anchor.x"], g["anchor.t"], 5.0)[0], 0.956884, rtol=1e-5)
    close(orc.mse(g["anchor.x"], g["anchor.t"], 1000)[0], 2022.908, rtol=1e-5)


def test_loss_constructor_variants(golden):
    """MSELoss(reduction='sum'), CrossEntropyLoss2d(weight, size_average): the oracle against the reference's own values and gradients."""
    g = golden("variants")
    loss, grad = orc.mse_sum(g["mse_sum.s"], g["mse_sum.t"], 19)
    close(loss, g["mse_sum.loss"], rtol=1e-5)
    close(grad, g["mse_sum.grad"], rtol=1e-5, atol=1e-7)
    for tag, w, avg in (("ce_w_mean", g["ce.w"], True), ("ce_w_sum", g["ce.w"], False), ("ce_sum", None, False)):
        loss, grad = orc.ce2d_weighted(g["ce.x"], g["ce.target"], w, avg)
        close(loss, g[f"{tag}.loss"], rtol=1e-5)
        close(grad, g[f"{tag}.grad"], rtol=1e-4, atol=1e-7)
    close(orc.ce2d_weighted(g["ce.x"], g["ce.target"], None, True, want_grad=False)[0], orc.ce2d(g["ce.x"], g["ce.target"]), rtol=1e-12)


def dwsep_bias_ref(g):
    """DepthwiseSeparableBlock(bias=True) restated from oracle ops: forward and all six gradients."""
    C, Co, k, p, d, H, W = [int(v) for v in g["dwsep_bias.cfg"]]
    x, wdw, bdw, wpw, bpw, gy = (g[f"dwsep_bias.{n}"] for n in ("x", "w_dw", "b_dw", "w_pw", "b_pw", "gy"))
    mid = orc.conv2d_fwd(x, wdw, bias=bdw, pad=p, dil=d, groups=C)
    y = orc.conv2d_fwd(mid, wpw, bias=bpw)
    gmid = orc.conv2d_dgrad(gy, wpw, mid.shape)
    return {"y": y, "gx": orc.conv2d_dgrad(gmid, wdw, x.shape, pad=p, dil=d, groups=C),
            "gw_dw": orc.conv2d_wgrad(x, gmid, wdw.shape, pad=p, dil=d, groups=C), "gb_dw": gmid.sum(axis=(0, 2, 3)),
            "gw_pw": orc.conv2d_wgrad(mid, gy, wpw.shape), "gb_pw": gy.sum(axis=(0, 2, 3))}


def test_dwsep_block_with_bias(golden):
    g = golden("variants")
    for n, v in dwsep_bias_ref(g).items():
        close(v, g[f"dwsep_bias.{n}"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag", ["k9d5", "k3d1"])
def test_dwsep_block(golden, tag):
    g = golden("dwsep")
    C, Co, k, p, d, H, W = [int(v) for v in g[f"{tag}.cfg"]]
    x, wdw, wpw = g[f"{tag}.x"], g[f"{tag}.w_dw"], g[f"{tag}.w_pw"]
    mid = orc.conv2d_fwd(x, wdw, pad=p, dil=d, groups=C)
    y = orc.conv2d_fwd(mid, wpw)
    close(y, g[f"{tag}.y"], rtol=1e-4, atol=1e-5)
    gy = g[f"{tag}.gy"]
    gw_pw = orc.conv2d_wgrad(mid, gy, wpw.shape)
    gmid = orc.conv2d_dgrad(gy, wpw, mid.shape)
    gw_dw = orc.conv2d_wgrad(x, gmid, wdw.shape, pad=p, dil=d, groups=C)
    gx = orc.conv2d_dgrad(gmid, wdw, x.shape, pad=p, dil=d, groups=C)
    close(gw_pw, g[f"{tag}.gw_pw"], rtol=1e-4, atol=1e-4)
    close(gw_dw, g[f"{tag}.gw_dw"], rtol=1e-4, atol=1e-4)
    close(gx, g[f"{tag}.gx"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag,s,p,d", [("s1d1", 1, 1, 1), ("s2d1", 2, 1, 1), ("s1d4", 1, 4, 4)])
def test_conv_variants(golden, tag, s, p, d):
    g = golden("ops")
    x, w = g["x"], g["w"]
    close(orc.conv2d_fwd(x, w, stride=s, pad=p, dil=d), g[f"conv_{tag}.y"], rtol=1e-4, atol=1e-5)
    gy = g[f"conv_{tag}.gy"]
    close(orc.conv2d_dgrad(gy, w, x.shape, stride=s, pad=p, dil=d), g[f"conv_{tag}.gx"], rtol=1e-4, atol=1e-5)
    close(orc.conv2d_wgrad(x, gy, w.shape, stride=s, pad=p, dil=d), g[f"conv_{tag}.gw"], rtol=1e-4, atol=1e-4)


def test_pool_upsample_gap(golden):
    g = golden("ops")
    x = g["x"]
    close(orc.maxpool3x3s2(x), g["maxpool"], rtol=0, atol=0)
    close(orc.upsample_bilinear_ac(x, (29, 40)), g["up"], rtol=1e-5, atol=5e-6)
    close(orc.upsample_bilinear_ac(x, (22, 28)), g["up2"], rtol=1e-5, atol=5e-6)
    close(orc.gap(x), g["gap"].reshape(2, 6), rtol=1e-5, atol=1e-7)


def test_bn_train(golden):
    g = golden("ops")
    y, mean, invstd = orc.bn_train_fwd(g["x"], g["bn_gamma"], g["bn_beta"], relu=True)
    close(y, g["bn_y"], rtol=1e-4, atol=1e-6)
    gx, gg, gb = orc.bn_train_bwd(g["bn_gy"], g["x"], y, g["bn_gamma"], mean, invstd, relu=True)
    close(gx, g["bn_gx"], rtol=1e-3, atol=1e-6)
    close(gg, g["bn_ggamma"], rtol=1e-4, atol=1e-5)
    close(gb, g["bn_gbeta"], rtol=1e-4, atol=1e-5)


def test_radam(golden):
    g = golden("radam")
    p = g["p"][0].copy()
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for i in range(8):
        orc.radam_step(p, g["g"][i], m, v, step=i + 1, lr=float(g["lr"]))
        close(p, g["p"][i + 1], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("tag", ["id2_d1", "proj2_s2", "proj2_d2", "bott_d4"])
def test_resblock_golden(golden, tag):
    """IdentityResidualBlock (eval BN, identity / projection shortcut, stride 2, dilation 2 / 4, bottleneck): the block
    restated from oracle.c ops (tests/_blockref.py) reproduces the reference's output and input gradient."""
    from _blockref import RESBLOCK_CASES, resblock_fwd_bwd
    g = golden("resblock")
    cin, ch, stride, dil = RESBLOCK_CASES[tag]
    assert [int(v) for v in g[f"{tag}.cfg"]] == [cin, stride, dil, len(ch)] + list(ch)
    y, gx = resblock_fwd_bwd(tag, g[f"{tag}.x"], g[f"{tag}.gy"])
    close(y, g[f"{tag}.y"], rtol=1e-4, atol=1e-5)
    close(gx, g[f"{tag}.gx"], rtol=1e-4, atol=1e-5)


def test_aspp_golden(golden):
    """ASPP module (image pooling + 1x1 + rates 12/24/36, BN-eval + ReLU, concat) vs the reference's output."""
    from _blockref import aspp_fwd
    g = golden("aspp")
    close(aspp_fwd(g["x"]), g["y"], rtol=1e-4, atol=1e-5)


def test_confusion_golden(golden):
    """Integer work: the 19x19 confusion matrix accumulated over two CityscapesMetricTracker.update() calls, bit-exact
    (ignore band, exact logit ties -> first maximum, a never-predicted and a never-labelled class), and get_iou()."""
    g = golden("confusion")
    conf = None
    for x, t in zip(g["x"], g["target"]):
        conf = orc.confusion(x, t.astype(np.int64), 255, conf)
    assert conf.dtype == np.int64 and np.array_equal(conf, g["conf"])
    assert conf[:, 17].sum() == 0 and conf[16].sum() == 0
    close(orc.miou(conf), g["miou"], rtol=1e-12)
