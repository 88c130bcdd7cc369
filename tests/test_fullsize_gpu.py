"""Full-size (BASELINE config: 1024x2048 input, trunk maps 128x256) checks through size-independent properties, where an
element-wise CPU oracle would take minutes: adjoint identities tying forward / dgrad / wgrad kernels together, exact
power-of-two linearity, loss identities, student == teacher when nothing is replaced, batch independence.  bf16 path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def K():
    import kdcc_amd
    from kdcc_amd import ops
    return ops


def dot(a, b):
    return float((a.double() * b.double()).sum())


def rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-30)


def test_conv_fwd_dgrad_adjoint_and_linearity(K):
    """<conv(x), g> == <x, dgrad(g)> for the mod5-shaped 3x3 dil-2 512->1024 conv at 128x256; conv(2x) == 2 conv(x) bitwise."""
    from kdcc_amd._lib import KD_PACK_DGRAD
    g0 = torch.Generator(device="cuda").manual_seed(1)
    N, H, W, Cin, Cout, d = 1, 128, 256, 512, 1024, 2
    x = torch.randn((N, H, W, Cin), device="cuda", generator=g0).to(BF)
    g = torch.randn((N, H, W, Cout), device="cuda", generator=g0).to(BF)
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g0) * 0.02
    wf = K.pack_conv_weight(w, BF)
    wd = K.pack_conv_weight(w, BF, KD_PACK_DGRAD)
    y = torch.empty((N, H, W, Cout), device="cuda", dtype=BF)
    K.conv2d(x, wf, 1, d, d, out_raw=y)
    g = y.clone()                                 # cotangent = the output itself: <y, g> = |y|^2, no cancellation
    gx = torch.empty((N, H, W, Cin), device="cuda", dtype=BF)
    K.conv2d(g, wd, 1, d, d, out_raw=gx)
    assert rel(dot(y, g), dot(x, gx)) < 2e-3      # the two sides carry independent bf16 output rounding
    y2 = torch.empty_like(y)
    K.conv2d(x * 2, wf, 1, d, d, out_raw=y2)
    assert torch.equal(y2, y * 2)                 # scaling by 2 commutes with every rounding step


def test_pointwise_adjoints(K):
    """1x1 conv 1024->2048 at 128x256: <pw(x), g> == <x, dgrad(g)> == <w, wgrad(x, g)>."""
    from kdcc_amd._lib import KD_PACK_DGRAD
    g0 = torch.Generator(device="cuda").manual_seed(2)
    N, H, W, Cin, Cout = 1, 128, 256, 1024, 2048
    x = torch.randn((N, H, W, Cin), device="cuda", generator=g0).to(BF)
    g = torch.randn((N, H, W, Cout), device="cuda", generator=g0).to(BF)
    w = (torch.randn((Cout, Cin, 1, 1), device="cuda", generator=g0) * 0.03).to(BF).float()   # bf16-exact weights
    y = torch.empty((N, H, W, Cout), device="cuda", dtype=BF)
    K.conv2d(x, K.pack_conv_weight(w, BF), out_raw=y)
    g = y.clone()
    gx = torch.empty((N, H, W, Cin), device="cuda", dtype=BF)
    K.conv2d(g, K.pack_conv_weight(w, BF, KD_PACK_DGRAD), out_raw=gx)
    gw = torch.empty_like(w)
    K.pw_wgrad(x, g, gw)
    a, b, c = dot(y, g), dot(x, gx), dot(w, gw)
    assert rel(a, b) < 2e-3 and rel(a, c) < 2e-3, (a, b, c)


def test_depthwise_adjoints_4096ch(K):
    """ASPP-branch depthwise 9x9/d5 on 4096 channels at 128x256: <dw(x), g> == <x, dw^T(g)> == <w, wgrad(x, g)>."""
    g0 = torch.Generator(device="cuda").manual_seed(3)
    N, H, W, Cc, k, p, d = 1, 128, 256, 4096, 9, 20, 5
    x = torch.randn((N, H, W, Cc), device="cuda", generator=g0).to(BF)
    g = torch.randn((N, H, W, Cc), device="cuda", generator=g0).to(BF)
    w = torch.randn((Cc, 1, k, k), device="cuda", generator=g0) / k
    y = K.dwconv(x, K.pack_dw_weight(w), k, p, d)
    g = y.clone()
    gx = K.dwconv(g, K.pack_dw_weight(w, flip=True), k, p, d)
    gw = torch.empty_like(w)
    K.dwconv_wgrad(x, g, gw, k, p, d)
    a, b, c = dot(y, g), dot(x, gx), dot(w, gw)
    assert rel(a, b) < 2e-3 and rel(a, c) < 2e-3, (a, b, c)
    y2 = K.dwconv(x * 2, K.pack_dw_weight(w), k, p, d)
    assert torch.equal(y2, y * 2)


def test_loss_identities_fullsize(K):
    """(1,19,1024,2048) logits, NHWC fp32 like the engine's: KL(s,s) = 0 with zero gradient; the KL gradient sums to zero
    over classes at every pixel; MSE / CE agree with independent torch evaluations."""
    g0 = torch.Generator(device="cuda").manual_seed(4)
    s = (torch.randn((1, 1024, 2048, 19), device="cuda", generator=g0) * 2).permute(0, 3, 1, 2)
    t = (torch.randn((1, 1024, 2048, 19), device="cuda", generator=g0) * 2).permute(0, 3, 1, 2)
    loss, grad = K.kldiv(s, s, 1.0)
    assert abs(loss.item()) < 1e-6 and float(grad.abs().max()) < 1e-9
    loss, grad = K.kldiv(s, t, 2.0)
    assert float(grad.sum(dim=1).abs().max()) < 1e-9
    ref = torch.nn.functional.kl_div(torch.log_softmax(s / 2, 1), torch.softmax(t / 2, 1), reduction="sum") * 4 / (1024 * 2048)
    assert rel(loss.item(), ref.item()) < 1e-4
    loss, grad = K.hint_mse(s, t, 1000)
    assert rel(loss.item(), ((s - t).double() ** 2).mean().item() * 1000) < 1e-5
    tgt = torch.randint(0, 19, (1, 1024, 2048), device="cuda", generator=g0)
    tgt[:, :32] = 255
    ce = K.ce2d(s, tgt)
    assert rel(ce.item(), torch.nn.functional.cross_entropy(s, tgt, ignore_index=255).item()) < 1e-4


def test_student_equals_teacher_when_nothing_is_replaced_fullsize():
    """Empty plan at 1024x2048: the student is a frozen copy of the teacher, so both engine passes must agree bit for bit
    (logits and hints), every hint loss and the KD loss are exactly zero, and a batch of two identical images gives two
    identical halves."""
    import kdcc_amd
    from kdcc_amd import losses
    from kdcc_amd.models import DeepWV3Plus
    from kdcc_amd.models.students import DepthwiseStudent
    torch.manual_seed(123)
    teacher = DeepWV3Plus(19).eval()
    model = DepthwiseStudent(teacher, None, dtype=BF).cuda()
    hints = ["mod4.block2.convs.conv2", "mod5.block1.convs.conv1", "mod7.block1.convs.conv2", "aspp.features.2.0"]
    model.register_hint_layers(hints)
    x1 = torch.randn((1, 3, 1024, 2048), generator=torch.Generator().manual_seed(5))
    x = torch.cat([x1, x1]).cuda()
    with torch.no_grad():
        out_st, out_tc = model(x)
    assert tuple(out_st.shape) == (2, 19, 1024, 2048) and out_st.dtype == torch.float32
    assert torch.equal(out_st, out_tc)
    assert torch.equal(out_st[0], out_st[1])
    assert len(model.student_hidden_outputs) == 4
    for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        assert torch.equal(s, t) and torch.isfinite(s.float()).all()
        assert losses.MSELoss(num_classes=1000)(s, t).item() == 0.0
    assert losses.KLDivergenceLoss(1)(out_st, out_tc).item() == 0.0
    # not degenerate: logits vary across pixels and classes
    assert float(out_st.std()) > 1e-3


def test_stem_wgrad_mfma_equals_the_valu_kernel_fullsize(K):
    """mod1.conv1's weight gradient at 2 x 1024 x 2048: the fp32-MFMA kernel (bf16 dy) against the VALU kernel fed the same values
    as fp32 -- same products, different summation order -- and the adjoint identity <conv(x; w), dy> == <w, dW> with the stem
    forward (wider_resnet.py:307-309)."""
    from kdcc_amd import _lib
    g0 = torch.Generator(device="cuda").manual_seed(5)
    N, H, W = 2, 1024, 2048
    x = torch.randn((N, 3, H, W), device="cuda", generator=g0)
    dy = torch.randn((N, H, W, 64), device="cuda", generator=g0).to(BF)
    dw, dw32 = torch.empty((64, 3, 3, 3), device="cuda"), torch.empty((64, 3, 3, 3), device="cuda")
    K.stem_wgrad(x, dy, dw)
    assert _lib.last_kernel() == "stem_wgrad_mfma_kernel"
    K.stem_wgrad(x, dy.float(), dw32)
    err = float((dw.double() - dw32.double()).norm() / dw32.double().norm())
    assert err < 1e-5, err
    dw2 = torch.empty_like(dw)
    K.stem_wgrad(x, dy, dw2)
    assert torch.equal(dw, dw2)                                   # run to run
    # adjoint with the stem forward: the forward rounds x and w to bf16 (MFMA operands), the gradient keeps x in fp32
    w = torch.randn((64, 3, 3, 3), device="cuda", generator=g0) * 0.2
    y = K.stem_conv(x, w, BF)
    dwy = torch.empty_like(dw)
    K.stem_wgrad(x, y, dwy)
    assert rel(dot(y, y), dot(w, dwy)) < 5e-3


def test_bn_sums_in_the_dgrad_epilogue_fullsize(K):
    """The eval-BN parameter sums of a mod5-shaped input gradient (3x3 dil 2, 1024 -> 512 at 4 x 128 x 256, mask + shortcut
    gradient: the bench's [mq] launches) taken in the epilogue against kd_channel_sums of the stored tensor, and
    S1 == column sums of the stored gradient minus the shortcut's."""
    from kdcc_amd._lib import KD_PACK_DGRAD
    g0 = torch.Generator(device="cuda").manual_seed(6)
    N, H, W, Cin, Cout, d = 4, 128, 256, 1024, 512, 2
    g = torch.randn((N, H, W, Cin), device="cuda", generator=g0).to(BF)
    w = torch.randn((Cin, Cout, 3, 3), device="cuda", generator=g0) * 0.02       # the forward conv's weight (Cout_fwd = Cin here)
    wd = K.pack_conv_weight(w, BF, KD_PACK_DGRAD)
    act = torch.relu(torch.randn((N, H, W, Cout), device="cuda", generator=g0)).to(BF)
    sub = torch.randn((N, H, W, Cout), device="cuda", generator=g0).to(BF)
    scale = torch.rand(Cout, device="cuda", generator=g0) + 0.5
    out = torch.empty((N, H, W, Cout), device="cuda", dtype=BF)
    got = []
    K.conv2d(g, wd, 1, d, d, mask=act, mask_scale=scale, res_post=sub, out_raw=out, bn_sums=got)
    assert len(got) == 1, "the persistent row kernel carries the sums at this shape"
    s1, s2 = got[0]
    c1, c2 = K.channel_sums(out, sub=sub, a=act)
    for a, b in ((s1, c1), (s2, c2)):
        assert float((a.double() - b.double()).abs().max()) / float(b.double().abs().max()) < 5e-3
    ref1 = (out.double() - sub.double()).sum((0, 1, 2))
    assert float((s1.double() - ref1).abs().max()) / float(ref1.abs().max()) < 5e-3
