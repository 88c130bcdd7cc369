"""The logged logit losses from the half-resolution logits (kd_ce2d_up / kd_kldiv_up) and the lazily materialised
full-resolution tensor behind `model(data) -> (output_st, output_tc)` (lazy.LazyLogits).  Reference semantics:
models/deeplabv3/deeplabv3.py:160-162 (bilinear up-sampling, align_corners=True; the Gated-SCNN uses the default False,
models/gscnn/gscnn.py:323) followed by losses/CrossEntropy.py:10-14 and losses/KLDiv.py:19-23."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from _seeded import seeded_fill_, seeded_input  # noqa: E402


def _ref(lo_nhwc, size, align):
    """fp32 torch reference of the full-resolution logits (N,C,H,W) on the CPU"""
    x = torch.from_numpy(lo_nhwc).permute(0, 3, 1, 2).contiguous()
    return F.interpolate(x, size=size, mode="bilinear", align_corners=align)


@pytest.mark.parametrize("h,w,H,W,align", [(12, 20, 24, 40, True), (23, 39, 46, 78, True), (16, 24, 31, 47, True), (12, 20, 24, 40, False),
                                           (9, 300, 18, 600, True), (14, 22, 25, 40, False)])
def test_losses_from_low_resolution_logits_match_the_upsampled_reference(h, w, H, W, align):
    from kdcc_amd import ops
    from oracle import oracle as orc
    rng = np.random.default_rng(h * 1000 + w)
    N, C = 2, 19
    s_lo = (rng.standard_normal((N, h, w, C)) * 3).astype(np.float32)
    t_lo = (rng.standard_normal((N, h, w, C)) * 3).astype(np.float32)
    tgt = rng.integers(0, C, (N, H, W)).astype(np.int64)
    tgt[:, : max(1, H // 8)] = 255                           # an ignore band
    tgt[0, -1, -1] = 255
    s_full, t_full = _ref(s_lo, (H, W), align).numpy(), _ref(t_lo, (H, W), align).numpy()
    if align:   # the plain-C oracle restates align_corners=True up-sampling: pin the torch reference to it
        np.testing.assert_allclose(orc.upsample_bilinear_ac(np.ascontiguousarray(s_lo.transpose(0, 3, 1, 2)), (H, W)), s_full, rtol=1e-3, atol=1e-3)
    ce_ref = orc.ce2d(s_full, tgt, 255)
    cu = lambda a: torch.from_numpy(a).cuda()
    ce = ops.ce2d_up(cu(s_lo), cu(tgt), (H, W), 255, align)
    np.testing.assert_allclose(ce.item(), ce_ref, rtol=1e-5)
    for T in (1.0, 4.0):
        kl_ref, _ = orc.kldiv(s_full, t_full, T, want_grad=False)
        kl = ops.kldiv_up(cu(s_lo), cu(t_lo), (H, W), T, align)
        np.testing.assert_allclose(kl.item(), kl_ref, rtol=2e-5, atol=1e-7)
    # ... and the materialised path of this library gives the same numbers
    full = ops.upsample_bilinear_ac(cu(s_lo), (H, W), out_dtype=torch.float32, align_corners=align).permute(0, 3, 1, 2)
    np.testing.assert_allclose(ops.ce2d(full, cu(tgt), 255).item(), ce.item(), rtol=1e-6)


def test_unsupported_ratio_is_refused_loudly():
    from kdcc_amd import _lib, ops
    x = torch.zeros(1, 4, 800, 19, device="cuda")
    with pytest.raises(_lib.KdccError):       # 4x down-sampling: 256 output pixels span > 160 source columns
        ops.ce2d_up(x, torch.zeros(1, 2, 200, dtype=torch.int64, device="cuda"), (2, 200))


def _model(arch="deeplab"):
    import kdcc_amd
    from kdcc_amd.models import GSCNN, DeepWV3Plus
    from kdcc_amd.models.students import DepthwiseStudent
    plan = ["mod4.block2.convs.conv2", "aspp.features.2.0"]
    teacher = GSCNN(num_classes=19) if arch == "gscnn" else DeepWV3Plus(num_classes=19)
    seeded_fill_(teacher, "gscnn." if arch == "gscnn" else "teacher.")
    teacher.eval()
    model = DepthwiseStudent(teacher, None, dtype=torch.bfloat16)
    model.replace([{"name": n, "epoch": 1} for n in plan], kernel_size=9, padding=20, dilation=5)
    model.register_hint_layers(plan)
    model.unfreeze(plan)
    for n in plan:
        seeded_fill_(model.get_block(n, model.student), f"student.{n}.")
    return model.cuda()


@pytest.mark.parametrize("arch", ["deeplab", "gscnn"])
def test_lazy_logits_behind_the_model_signature(arch, monkeypatch):
    """model(data) returns LazyLogits for both networks when no logit loss is back-propagated; the criteria take the
    half-resolution fast path and give the numbers of the materialised tensors; any torch op materialises (once) exactly what the
    eager up-sampling produces; the hint backward is unaffected."""
    from kdcc_amd import _lib, losses
    from kdcc_amd.lazy import LazyLogits
    x = seeded_input("lazy.x", (2, 3, 64, 128), scale=30.0 if arch == "gscnn" else 1.0).cuda()
    tgt = torch.randint(0, 19, (2, 64, 128), generator=torch.Generator().manual_seed(3))
    tgt[:, :8] = 255
    tgt = tgt.cuda()
    ce, kl, mse = losses.CrossEntropyLoss2d(ignore_index=255), losses.KLDivergenceLoss(1), losses.MSELoss(num_classes=1000)

    def step(lazy):
        model = _model(arch)
        monkeypatch.setenv("KDCC_LAZY_LOGITS", "1" if lazy else "0")
        model._engine = None
        model._teacher_engine = None
        with _lib.kernel_log() as log:
            out_st, out_tc = model(x)
            vals = [ce(out_st, tgt), kl(out_st, out_tc), ce(out_tc, tgt)]
            hint = 0
            for s_, t_ in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
                hint = hint + mse(s_, t_)
            hint.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.clone() for n, p in model.student.named_parameters() if p.requires_grad}
        return out_st, out_tc, [float(v) for v in vals] + [float(hint)], grads

    st1, tc1, v1, g1 = step(True)
    assert isinstance(st1, LazyLogits) and isinstance(tc1, LazyLogits) and st1.pending and tc1.pending      # nothing up-sampled
    st0, tc0, v0, g0 = step(False)
    assert type(st0) is torch.Tensor and not isinstance(tc0, LazyLogits)
    np.testing.assert_allclose(v1, v0, rtol=2e-5)      # (the KL of two nearly equal networks is 1e-4: fp32 summation order shows)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    # any other reader gets the real tensor, bit for bit the eager one
    assert tuple(st1.shape) == tuple(st0.shape) and st1.stride() == st0.stride() and st1.dtype == st0.dtype
    pred = st1.argmax(1)
    assert not st1.pending and torch.equal(pred, st0.argmax(1))
    assert torch.equal(st1.materialize(), st0) and torch.equal(tc1.cpu(), tc0.cpu())
    assert np.isfinite(ce(st1, tgt).item())                     # a materialised LazyLogits goes down the ordinary path
    if arch == "gscnn":
        return          # (its shape stream keeps nothing for a backward unless logits_need_grad is set: refused, test_gscnn_gpu.py)
    # a logit loss that IS back-propagated, on a model nobody told (logits_need_grad False): the deferred path materialises the
    # logits in backward and gives the gradients of the eager model
    def kd_step(lazy):
        monkeypatch.setenv("KDCC_LAZY_LOGITS", "1" if lazy else "0")
        model = _model(arch)
        out_st, out_tc = model(x)
        assert isinstance(out_st, LazyLogits) == lazy
        loss = kl(out_st, out_tc) + 0.5 * ce(out_st, tgt)
        for s_, t_ in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
            loss = loss + mse(s_, t_)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {n: p.grad.clone() for n, p in model.student.named_parameters() if p.requires_grad}
    l1, k1 = kd_step(True)
    l0, k0 = kd_step(False)
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    for n in k0:
        assert torch.equal(k0[n], k1[n]), n


def test_trainer_metrics_read_lazy_logits():
    """The device confusion matrix (logged mIoU) takes raw pointers: a LazyLogits is materialised for it."""
    from kdcc_amd import ops
    from kdcc_amd.lazy import LazyLogits
    g = torch.Generator(device="cuda").manual_seed(5)
    low = torch.randn(2, 16, 32, 19, device="cuda", generator=g)
    tgt = torch.randint(0, 19, (2, 32, 64), device="cuda", generator=g)
    lazy = LazyLogits(low, (32, 64))
    full = ops.upsample_bilinear_ac(low, (32, 64), out_dtype=torch.float32).permute(0, 3, 1, 2)
    assert torch.equal(ops.confusion(lazy, tgt), ops.confusion(full, tgt)) and not lazy.pending
