"""The MEASURED path (bf16, the kernels bench.py times) under the network-level oracle at shapes where the dispatchers pick
the shipped kernels -- the one-wave-per-SIMD row kernel (hand-scheduled loop), the persistent 1x1 ping-pong kernel, the 512x128 mod2 kernel, the matrix-core depthwise forward /
fan-out / summed input gradient / weight gradient, the fused stem + pool -- with the selection asserted through the kernel log
(kd_debug_kernel_log_*).  The per-kernel bf16 oracle tests live in test_ops_gpu.py; this file checks the ENGINE WIRING of those
kernels (trainer/layerwise_trainer.py:220-239 of the reference: forward, criteria, backward) against oracle/net_ref.py.

bf16 cannot meet the fp32 bar through 38 layers: the bars here are relative L2 on logits / hints, the
loss within a few percent, and for every gradient tensor cosine similarity + norm ratio against the fp32 oracle.

Selecting the wide persistent kernels needs >= 224 tiles of 256 pixels x 256 channels per layer and trunk rows that are
multiples of 256 pixels: the trunk runs at 1/8 resolution, so the input must be 2048 wide; 2 x 512 x 2048 gives every
>= 512-channel layer 256+ tiles.  The CPU oracle takes about a minute on that (16 host threads)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _netutil import canny_stub_map, seeded_cheap_weights, seeded_gscnn_sd, seeded_teacher_sd  # noqa: E402
from _seeded import seeded_fill_, seeded_input  # noqa: E402

P92 = ["mod4.block2.convs.conv2", "mod4.block3.convs.conv1", "mod7.block1.convs.conv2",
       "aspp.features.1.0", "aspp.features.2.0", "aspp.features.3.0"]
BF = torch.bfloat16


def _threads():
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))


def _build(plan, dtype, arch="deeplab"):
    import kdcc_amd
    from kdcc_amd.models import GSCNN, DeepWV3Plus
    from kdcc_amd.models.students import DepthwiseStudent
    teacher = GSCNN(num_classes=19) if arch == "gscnn" else DeepWV3Plus(num_classes=19)
    seeded_fill_(teacher, "gscnn." if arch == "gscnn" else "teacher.")
    teacher.eval()
    model = DepthwiseStudent(teacher, None, dtype=dtype)
    model.replace([{"name": n, "epoch": 1} for n in plan], kernel_size=9, padding=20, dilation=5)
    model.register_hint_layers(plan)
    model.unfreeze(plan)
    for n in plan:
        seeded_fill_(model.get_block(n, model.student), f"student.{n}.")
    return model.cuda()


def _rel_l2(got, ref):
    got, ref = got.detach().float().cpu().double(), ref.detach().float().cpu().double()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def _worst_tile(got, ref, px=256, ch=256):
    """Worst relative L2 error over the conv kernels' output tiles of an NCHW-logical tensor: blocks of `px` consecutive pixels of one
    image row x `ch` channels (256 x 256 = the workgroup tile of the persistent kernels; a narrower tensor is one channel block).  A
    global norm averages one wrong tile away (1 of ~2000 at these sizes moves the global relative L2 by 2 %); here it is the maximum.
    The denominator is the tile's own reference norm, floored at a quarter of the mean tile norm (near-empty tiles)."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    N, Cc, H, W = ref.shape
    px, ch = min(px, W), min(ch, Cc)
    Wt, Ct = W // px * px, Cc // ch * ch
    e = ((got - ref)[:, :Ct, :, :Wt].double() ** 2).reshape(N, Ct // ch, ch, H, Wt // px, px).sum(dim=(2, 5))
    r = (ref[:, :Ct, :, :Wt].double() ** 2).reshape(N, Ct // ch, ch, H, Wt // px, px).sum(dim=(2, 5))
    floor = r.mean() / 16.0
    return float((e / torch.maximum(r, floor)).max().sqrt())


def _grad_report(model, ref_grads, names=None):
    """[(name, cosine, norm ratio)] of every trainable tensor's gradient against the fp32 oracle's."""
    rows = []
    for n, p in model.student.named_parameters():
        if not p.requires_grad or (names is not None and n not in names):
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        g, r = p.grad.detach().cpu().double().reshape(-1), ref_grads[n].double().reshape(-1)
        cos = float((g @ r) / (g.norm() * r.norm()).clamp_min(1e-300))
        rows.append((n, cos, float(g.norm() / r.norm().clamp_min(1e-300))))
    return rows


def _step(model, backprop="hint"):
    from kdcc_amd import losses
    out_st, out_tc = model(model._x)
    crit = losses.MSELoss(num_classes=1000)
    hint = 0
    for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        hint = hint + crit(s, t)
    kd = losses.KLDivergenceLoss(1)(out_st, out_tc)
    loss = hint if backprop == "hint" else kd + hint
    loss.backward()
    torch.cuda.synchronize()
    return out_st, out_tc, hint, kd, loss


SHIPPED_MODE_A = ["conv_row_lw_kernel", "conv_igemm_persist_kernel<pp>", "conv_row_tall_kernel",
                  "dw_mfma_fwd_kernel<1,false>", "dw_lw_fan3_kernel", "dw_mfma_fwd_kernel<3,false>", "out_sums_epilogue",
                  "dw_mfma_wgrad_kernel", "dw_mfma_wgrad_multi_kernel<3>", "stem_pool_kernel", "conv_wgrad_pw_lw_kernel", "cls_epilogue"]   # (the pointwise weight gradients take the 256x256 wgrad tile, one wave per SIMD)


def test_bf16_p92_step_on_the_shipped_kernels_vs_network_oracle():
    """BASELINE config 2's step (plan P92, loss = hint MSE, bf16) at 2 x 512 x 2048: every kernel class the 1024x2048 bench
    step runs is selected (asserted), logits / hints / loss / all 12 gradients against oracle/net_ref.py (fp32 CPU)."""
    from kdcc_amd import _lib
    from oracle import net_ref
    model = _build(P92, BF)
    x = seeded_input("bf16.p92.x", (2, 3, 512, 2048))
    model._x = x.cuda()
    with _lib.kernel_log() as log:
        out_st, out_tc, hint, kd, _ = _step(model)
    missing = [k for k in SHIPPED_MODE_A if log.counts.get(k, 0) == 0]
    assert not missing, f"kernels the bench step runs but this step did not select: {missing}; selected: {log.counts}"
    # the 3x3 / 1x1 layers of mod3..mod7 and the decoder must be on the persistent kernels, not on the one-tile fallbacks
    persistent = log.counts["conv_row_lw_kernel"] + log.counts["conv_igemm_persist_kernel<pp>"] + log.counts["conv_row_tall_kernel"]
    total_conv = sum(v for k, v in log.counts.items() if k.startswith(("conv_row_", "conv_igemm_")))
    assert persistent >= 0.75 * total_conv, log.counts

    tsd = seeded_teacher_sd()
    ssd = net_ref.make_student_sd(tsd, P92, seeded_cheap_weights(tsd, P92))
    _threads()
    r = net_ref.kd_step(tsd, ssd, x, None, P92)
    assert model.student_hint_names == r["hint_names"]
    errs = {"student logits": _rel_l2(out_st, r["student_logits"]), "teacher logits": _rel_l2(out_tc, r["teacher_logits"])}
    for i, (s, t) in enumerate(zip(model.student_hidden_outputs, model.teacher_hidden_outputs)):
        errs[f"student hint {i}"] = _rel_l2(s, r["student_hints"][i])
        errs[f"teacher hint {i}"] = _rel_l2(t, r["teacher_hints"][i])
    tiles = {"student logits": _worst_tile(out_st, r["student_logits"], px=512), "teacher logits": _worst_tile(out_tc, r["teacher_logits"], px=512)}
    for i, (s, t) in enumerate(zip(model.student_hidden_outputs, model.teacher_hidden_outputs)):
        tiles[f"student hint {i}"] = _worst_tile(s, r["student_hints"][i])
        tiles[f"teacher hint {i}"] = _worst_tile(t, r["teacher_hints"][i])
    rows = _grad_report(model, r["grads"])
    print("bf16 P92 2x512x2048 vs net_ref:", {k: f"{v:.2e}" for k, v in errs.items()}, "worst tile", {k: f"{v:.2e}" for k, v in tiles.items()},
          "hint", hint.item(), r["hint_loss"].item(), [(n, f"{c:.5f}", f"{q:.4f}") for n, c, q in rows])
    # ... and no single 256-pixel x 256-channel tile may stand out: a tile that lost one 64-channel k-step of K = 4608 is off by 12 %,
    # a stale or unwritten one by 100 %; measured: the worst tile is within 1.2x of the global figure (logits 5e-3, hints 1.2e-2); bars 1.5x the global ones
    assert tiles["student logits"] < 1.5e-2 and tiles["teacher logits"] < 1.5e-2 and max(tiles.values()) < 3e-2, tiles
    # measured (round 3): logits 4e-3, hints 7e-3 .. 1e-2 relative L2, hint loss 8e-4, gradient cosines >= 0.99997, norms within
    # 0.13 % -- the bars leave 2-3x of that
    assert errs["student logits"] < 1e-2 and errs["teacher logits"] < 1e-2 and max(errs.values()) < 2e-2, errs
    assert abs(hint.item() - r["hint_loss"].item()) <= 5e-3 * abs(r["hint_loss"].item())
    assert abs(kd.item() - r["kd_loss"].item()) <= 5e-2 * abs(r["kd_loss"].item()) + 1e-6
    assert len(rows) == 12
    bad = [(n, c, q) for n, c, q in rows if c < 0.9995 or abs(q - 1) > 0.01]
    assert not bad, bad


P79 = ["mod4.block2.convs.conv2", "mod4.block3.convs.conv1", "mod4.block3.convs.conv2", "mod4.block4.convs.conv2",
       "mod4.block5.convs.conv2", "mod4.block6.convs.conv2", "mod5.block2.convs.conv2", "mod7.block1.convs.conv2",
       "aspp.features.1.0", "aspp.features.2.0", "aspp.features.3.0"]      # cfg/cityscapes/58M_deeplab_all.json:123-168


def _check_step(model, r, hint, kd, out_st, out_tc, n_grads, what):
    errs = {"student logits": _rel_l2(out_st, r["student_logits"]), "teacher logits": _rel_l2(out_tc, r["teacher_logits"])}
    for i, (s, t) in enumerate(zip(model.student_hidden_outputs, model.teacher_hidden_outputs)):
        errs[f"student hint {i}"] = _rel_l2(s, r["student_hints"][i])
        errs[f"teacher hint {i}"] = _rel_l2(t, r["teacher_hints"][i])
    tiles = {"student logits": _worst_tile(out_st, r["student_logits"], px=512), "teacher logits": _worst_tile(out_tc, r["teacher_logits"], px=512)}
    for i, (s, t) in enumerate(zip(model.student_hidden_outputs, model.teacher_hidden_outputs)):
        tiles[f"student hint {i}"] = _worst_tile(s, r["student_hints"][i])
        tiles[f"teacher hint {i}"] = _worst_tile(t, r["teacher_hints"][i])
    rows = _grad_report(model, r["grads"])
    print(what, {k: f"{v:.2e}" for k, v in errs.items()}, "worst tile", {k: f"{v:.2e}" for k, v in tiles.items()}, "hint", hint.item(),
          r["hint_loss"].item(), "worst gradients", sorted(rows, key=lambda t: t[1])[:3])
    assert tiles["student logits"] < 1.5e-2 and tiles["teacher logits"] < 1.5e-2 and max(tiles.values()) < 3e-2, tiles   # (see the P92 test)
    # the bars of the P92 test: logits 1e-2, hints 2e-2 relative L2, hint loss 5e-3, gradient cosine 0.9995, norm within 1 %
    assert errs["student logits"] < 1e-2 and errs["teacher logits"] < 1e-2 and max(errs.values()) < 2e-2, errs
    assert abs(hint.item() - r["hint_loss"].item()) <= 5e-3 * abs(r["hint_loss"].item())
    assert abs(kd.item() - r["kd_loss"].item()) <= 5e-2 * abs(r["kd_loss"].item()) + 1e-6
    assert len(rows) == n_grads
    bad = [(n, c, q) for n, c, q in rows if c < 0.9995 or abs(q - 1) > 0.01]
    assert not bad, bad


def test_bf16_weighted_hint_step_vs_network_oracle():
    """BASELINE config 4 on the measured path: WeightedHintMSELoss (losses/WeightedHintMSELoss.py:12-16 of the reference) with
    filter_weight = rand(C), generator seed 7 + hint index (what trainer.hint_filter_weight = 'rand:7' gives), plan P92, bf16,
    1 x 256 x 2048 (the persistent kernels are selected), against oracle/net_ref.kd_step(hint_weights=...)."""
    from kdcc_amd import _lib, losses
    from oracle import net_ref
    model = _build(P92, BF)
    x = seeded_input("bf16.whint.x", (1, 3, 256, 2048))
    with _lib.kernel_log() as log:
        out_st, out_tc = model(x.cuda())
        crit = losses.WeightedHintMSELoss()
        ws = [torch.rand(s.shape[1], generator=torch.Generator().manual_seed(7 + i)) for i, s in enumerate(model.student_hidden_outputs)]
        hint = 0
        for i, (s, t) in enumerate(zip(model.student_hidden_outputs, model.teacher_hidden_outputs)):
            hint = hint + crit(s, t, ws[i].cuda())
        kd = losses.KLDivergenceLoss(1)(out_st, out_tc)
        hint.backward()
        torch.cuda.synchronize()
    for k in ("conv_row_lw_kernel", "conv_igemm_persist_kernel<pp>", "dw_lw_fan3_kernel", "dw_mfma_wgrad_multi_kernel<3>"):
        assert log.counts.get(k, 0) > 0, (k, log.counts)
    tsd = seeded_teacher_sd()
    ssd = net_ref.make_student_sd(tsd, P92, seeded_cheap_weights(tsd, P92))
    _threads()
    r = net_ref.kd_step(tsd, ssd, x, None, P92, hint_weights=ws)
    _check_step(model, r, hint, kd, out_st, out_tc, 12, "bf16 weighted hints 1x256x2048 vs net_ref:")


def test_bf16_p79_step_vs_network_oracle():
    """The shipped 58M plan (cfg/cityscapes/58M_deeplab_all.json: eleven cheap-conv blocks incl. mod5.block2's dw-512 -> pw-1024
    and five more mod4 blocks) in bf16 at 1 x 256 x 2048, hint MSE, against oracle/net_ref.py: all 22 gradients."""
    from kdcc_amd import _lib
    from oracle import net_ref
    model = _build(P79, BF)
    x = seeded_input("bf16.p79.x", (1, 3, 256, 2048))
    model._x = x.cuda()
    with _lib.kernel_log() as log:
        out_st, out_tc, hint, kd, _ = _step(model)
    for k in ("conv_row_lw_kernel", "conv_igemm_persist_kernel<pp>", "dw_mfma_fwd_kernel<1,false>", "dw_mfma_wgrad_kernel"):
        assert log.counts.get(k, 0) > 0, (k, log.counts)
    assert log.counts["dw_mfma_fwd_kernel<1,false>"] >= 8          # the eight trunk blocks' depthwise forwards
    tsd = seeded_teacher_sd()
    ssd = net_ref.make_student_sd(tsd, P79, seeded_cheap_weights(tsd, P79))
    _threads()
    r = net_ref.kd_step(tsd, ssd, x, None, P79)
    assert model.student_hint_names == r["hint_names"]
    _check_step(model, r, hint, kd, out_st, out_tc, 22, "bf16 P79 1x256x2048 vs net_ref:")


def test_bf16_mode_b_step_on_the_shipped_kernels_vs_network_oracle():
    """Mode B (loss = KLDiv + hints, every student parameter trainable) in bf16 at 1 x 256 x 2048: the row-buffer / wide dense
    weight-gradient kernels, the strided dgrad, pools / upsamples / stem backward as wired by the engine, vs net_ref."""
    from kdcc_amd import _lib
    from oracle import net_ref
    model = _build(P92, BF)
    for p in model.student.parameters():
        p.requires_grad = True
    x = seeded_input("bf16.modeb.x", (1, 3, 256, 2048))
    model._x = x.cuda()
    with _lib.kernel_log() as log:
        out_st, out_tc, hint, kd, loss = _step(model, "kd+hint")
    for k in ("conv_wgrad_lw_kernel", "conv_wgrad_pw_lw_kernel", "conv_wgrad_wide_kernel", "conv_row_lw_kernel", "conv_igemm_persist_kernel<pp>",
              "dw_mfma_wgrad_kernel", "dw_mfma_fwd_kernel<3,false>", "bn_sums_epilogue"):
        assert log.counts.get(k, 0) > 0, (k, log.counts)
    # the eval-BN parameter sums ride in the input-gradient epilogues: only the sites those kernels do not cover still read the
    # gradient back (depthwise sites, three-operand epilogues, narrow tiles)
    print("bf16 mode B: fused BN-sums launches", log.counts["bn_sums_epilogue"])
    tsd = seeded_teacher_sd()
    ssd = net_ref.make_student_sd(tsd, P92, seeded_cheap_weights(tsd, P92), trainable="all")
    _threads()
    r = net_ref.kd_step(tsd, ssd, x, None, P92, backprop="kd+hint")
    assert _rel_l2(out_st, r["student_logits"]) < 1e-2
    assert abs(loss.item() - r["loss"].item()) <= 5e-3 * abs(r["loss"].item())
    rows = _grad_report(model, r["grads"])
    assert len(rows) == len(list(model.student.parameters()))
    # measured (round 3): worst cosine 0.99990, worst norm ratio 0.9942 over all 145 tensors, loss within 8e-4
    bad = [(n, c, q) for n, c, q in rows if (c < 0.999 or abs(q - 1) > 0.02)]
    print("bf16 mode B 1x256x2048 vs net_ref: worst", sorted(rows, key=lambda t: t[1])[:5], "loss", loss.item(), r["loss"].item())
    assert not bad, bad


def test_bf16_mode_b_bn_gradients_fused_sums_equal_the_read_back_path(monkeypatch):
    """The eval-BN weight / bias gradients with their sums taken in the input-gradient epilogues (kd_conv_epilogue.bn_sums)
    against the same step with kd_channel_sums reading every gradient tensor back (KDCC_FUSE_BN_SUMS=0), at a size where the
    persistent kernels carry the sums: every other gradient bit-identical (the stored tensors do not change), the BN ones equal
    to the rounding of the stored bf16 gradient."""
    from kdcc_amd import _lib, engine
    x = seeded_input("bf16.modeb.x", (1, 3, 256, 2048)).cuda()
    grads = {}
    for fused in (True, False):
        monkeypatch.setattr(engine, "_FUSE_BN_SUMS", fused)
        model = _build(P92, BF)
        for p in model.student.parameters():
            p.requires_grad = True
        model._x = x
        with _lib.kernel_log() as log:
            _step(model, "kd+hint")
        assert (log.counts.get("bn_sums_epilogue", 0) > 0) == fused, log.counts
        grads[fused] = {n: p.grad.detach().clone() for n, p in model.student.named_parameters()}
    worst = 0.0
    for n, g in grads[True].items():
        r = grads[False][n]
        if g.dim() == 1:      # BN weights / biases (and the classifier bias, which does not depend on the sums)
            err = float((g.double() - r.double()).norm() / r.double().norm().clamp_min(1e-30))
            worst = max(worst, err)
            assert err < 5e-3, (n, err)
        else:
            assert torch.equal(g, r), n
    print("fused vs read-back BN parameter gradients: worst relative L2", worst)


def test_bf16_gscnn_step_vs_network_oracle():
    """Gated-SCNN student (BASELINE config 5; cheap convs in mod4 / mod7 / ASPP, hint MSE) in bf16 at 1 x 256 x 2048: the
    shape stream's small-channel kernels (conv3x3_small, pointwise_small, matrix-core gated conv) and the 512x128 kernel at
    full resolution, vs oracle/net_ref.py gscnn_forward with the same seeded Canny map on both sides."""
    from kdcc_amd import _lib
    from oracle import net_ref
    plan = ["mod4.block2.convs.conv2", "mod4.block3.convs.conv1", "mod7.block1.convs.conv2", "aspp.features.1.0", "aspp.features.3.0"]
    model = _build(plan, BF, arch="gscnn")
    maps = canny_stub_map((1, 256, 2048), 811)
    for net in (model.teacher, model.student):
        net.canny_fn = lambda x: maps.cuda()
    x = seeded_input("bf16.gscnn.x", (1, 3, 256, 2048), scale=30.0)
    model._x = x.cuda()
    with _lib.kernel_log() as log:
        out_st, out_tc, hint, kd, _ = _step(model)
    for k in ("conv3x3_small_kernel<64>", "gated_conv_mfma_kernel", "conv_row_lw_kernel", "conv_igemm_persist_kernel<pp>"):
        assert log.counts.get(k, 0) > 0, (k, log.counts)
    tsd = seeded_gscnn_sd()
    ssd = net_ref.make_student_sd(tsd, plan, seeded_cheap_weights(tsd, plan))
    _threads()
    r = net_ref.kd_step(tsd, ssd, x, None, plan, canny=maps.unsqueeze(1))
    errs = {"student logits": _rel_l2(out_st, r["student_logits"]), "teacher logits": _rel_l2(out_tc, r["teacher_logits"])}
    rows = _grad_report(model, r["grads"])
    print("bf16 GSCNN 1x256x2048 vs net_ref:", errs, "hint", hint.item(), r["hint_loss"].item(), rows)
    assert max(errs.values()) < 1.5e-2, errs       # measured 5.8e-3 / 5.9e-3
    assert abs(hint.item() - r["hint_loss"].item()) <= 5e-3 * abs(r["hint_loss"].item())
    bad = [(n, c, q) for n, c, q in rows if c < 0.9995 or abs(q - 1) > 0.01]
    assert len(rows) == 2 * len(plan) and not bad, bad


# ----------------------------------------------------------------------------------------------- the bench step itself
def _bench_run(steps=2, share=False, batch=8, seed_data=1000, plan="P92", mode="A", arch="deeplab"):
    """bench.py's own construction (build / kd_step, bf16, 1024x2048) for `steps` train steps."""
    import bench
    model, crit, opt, _ = bench.build(bench.PLANS[plan], BF, torch.device("cuda", 0), mode=mode, arch=arch)
    model.share_frozen_prefix = share
    g = torch.Generator().manual_seed(seed_data)
    data = torch.randn((batch, 3, 1024, 2048), generator=g).cuda()
    target = torch.randint(0, 19, (batch, 1024, 2048), generator=g)
    target[:, :32] = 255
    target = target.cuda()
    losses_seen, grads = [], None
    for i in range(steps):
        out_st, out_tc = model(data)
        sup, kd, tl = crit[0](out_st, target), crit[1](out_st, out_tc), crit[0](out_tc, target)
        hint = 0
        for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
            hint = hint + crit[2](s, t)
        (kd + hint if mode == "B" else hint).backward()      # bench.kd_step: mode B back-propagates the KD term too
        if i == steps - 1:
            grads = {n: p.grad.clone() for n, p in model.student.named_parameters() if p.requires_grad}
            logits = out_st.detach().clone()
            hints = [h.detach().clone() for h in model.student_hidden_outputs]
        opt.step()
        opt.zero_grad()
        losses_seen.append(torch.stack([hint.detach(), sup.detach(), kd.detach(), tl.detach()]))
    torch.cuda.synchronize()
    params = {n: p.detach().clone() for n, p in model.student.named_parameters() if p.requires_grad}
    return dict(losses=torch.stack(losses_seen).cpu(), grads=grads, params=params, logits=logits, hints=hints, model=model,
                data=data)


def test_fullsize_bench_step_determinism_batch_independence_and_prefix_sharing():
    """The step bench.py times (P92, 8 x 1024 x 2048, bf16 -- bench.py's default per-GPU batch), two train steps:
      * run-to-run: losses, gradients and updated parameters are bit-identical between two fresh runs (every reduction is
        fixed-order: DESIGN.md section 3);
      * batch independence: image 0 alone gives bit-identical logits / hints to image 0 of the batch of eight (tiles never
        straddle images);
      * share_frozen_prefix equals the two-full-forwards result bit for bit at full size;
      * the selected kernels are the shipped ones."""
    from kdcc_amd import _lib
    with _lib.kernel_log() as log:
        a = _bench_run()
    missing = [k for k in SHIPPED_MODE_A if log.counts.get(k, 0) == 0]
    assert not missing, (missing, log.counts)
    b = _bench_run()
    assert torch.equal(a["losses"], b["losses"]), (a["losses"], b["losses"])
    assert torch.isfinite(a["losses"]).all() and float(a["losses"][0, 0]) > 0
    for n in a["grads"]:
        assert torch.equal(a["grads"][n], b["grads"][n]), f"gradient of {n} differs between two identical runs"
        assert torch.equal(a["params"][n], b["params"][n]), f"{n} differs after two identical runs"
    # batch independence of the forward (fresh model in the same state as run a's second step is not needed: use b's model,
    # whose parameters equal a's after 2 steps -- compare image 0 of the 8-batch with the 1-batch forward)
    model = b["model"]
    with torch.no_grad():
        full_st, _ = model(b["data"])
        full_h = [h.clone() for h in model.student_hidden_outputs]
        one_st, _ = model(b["data"][:1].contiguous())
        one_h = list(model.student_hidden_outputs)
    assert torch.equal(full_st[:1], one_st)
    for hf, ho in zip(full_h, one_h):
        assert torch.equal(hf[:1], ho)
    del b, model, full_st, one_st, full_h, one_h
    torch.cuda.empty_cache()
    c = _bench_run(share=True)
    assert torch.equal(a["losses"], c["losses"])
    for n in a["grads"]:
        assert torch.equal(a["grads"][n], c["grads"][n]), f"share_frozen_prefix changed the gradient of {n}"
    assert torch.equal(a["logits"], c["logits"])


@pytest.mark.parametrize("name,kw,must", [
    ("modeB", dict(mode="B"), ("conv_wgrad_lw_kernel", "conv_wgrad_pw_lw_kernel", "conv_wgrad_wide_kernel", "conv_row_lw_kernel", "conv_igemm_persist_kernel<pp>",
                              "conv_row_tall_kernel", "dw_mfma_wgrad_multi_kernel<3>", "bn_sums_epilogue", "stem_wgrad_mfma_kernel")),
    ("gscnn_P86", dict(arch="gscnn", plan="P86"), ("conv_row_lw_kernel", "conv_igemm_persist_kernel<pp>", "conv_row_tall_kernel",
                                                   "conv3x3_small_kernel<64>", "gated_conv_mfma_kernel", "dw_lw_fan3_kernel")),
])
def test_fullsize_bench_subrecords_are_deterministic(name, kw, must):
    """The other two steps bench.py times as sub-records -- mode B (loss = KLDiv + hints, every student parameter trainable) and
    the Gated-SCNN student of BASELINE config 5 (plan P86) -- at the bench's own size, 8 x 1024 x 2048, two train steps: the
    kernels the record's rooflines name are the ones selected, and losses, every gradient and every updated parameter are
    bit-identical between two fresh runs."""
    from kdcc_amd import _lib
    with _lib.kernel_log() as log:
        a = _bench_run(**kw)
    missing = [k for k in must if log.counts.get(k, 0) == 0]
    assert not missing, (name, missing, log.counts)
    la = a["losses"]
    assert torch.isfinite(la).all() and float(la[0, 0]) > 0 and float(la[1, 0]) != float(la[0, 0])      # the parameters moved
    grads_a = {n: g.cpu() for n, g in a["grads"].items()}
    params_a = {n: g.cpu() for n, g in a["params"].items()}
    del a
    torch.cuda.empty_cache()
    b = _bench_run(**kw)
    print(name, "losses (hint, supervised, kd, teacher) per step:", la.tolist())
    assert torch.equal(la, b["losses"]), (la, b["losses"])
    for n in grads_a:
        assert torch.equal(grads_a[n], b["grads"][n].cpu()), f"{name}: gradient of {n} differs between two identical runs"
        assert torch.equal(params_a[n], b["params"][n].cpu()), f"{name}: {n} differs after two identical runs"


def test_bench_two_ranks_child_process(tmp_path):
    """bench.py's N > 1 path end to end as the driver launches it (python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2), in a FRESH child process: both ranks share this box's one GPU (KDCC_DIST_SHARE_GPU=1) and gloo carries the
    device buckets (RCCL refuses two ranks per device).  Each rank trains on its own data shard, so bit-identical replicas
    after the run prove that every gradient bucket was exchanged before the optimizer read it."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:                   # a free rendezvous port, not a fixed one
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, KDCC_DIST_SHARE_GPU="1", KDCC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "1", "--steps", "3", "--warmup", "1",
           "--height", "512", "--width", "1024", "--no-batch-sweep", "--no-sub-records", "--full-record", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]      # rank 0 prints ONE JSON line
    assert len(lines[0]) < 4096                   # ... short enough for the driver's stdout window
    rec = json.loads(lines[0])
    assert rec["roofline"]["frac"] > 0 and json.load(open(tmp_path / "full.json"))["value"] == pytest.approx(rec["value"], rel=1e-3)
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["replicas_identical_after_run"] is True
    assert rec["value"] > 0 and rec["steps"] == 3


def test_bench_gpus_2_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2 ...` exactly as the driver types it -- no torch.distributed.run on the command line, no WORLD_SIZE:
    bench.py starts its own two ranks as fresh child processes (bench._launch_ranks), relays rank 0's ONE JSON line and the exit
    code.  Both ranks share this box's GPU and gloo carries the buckets, as in the test above."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KDCC_DIST_SHARE_GPU="1", KDCC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-sub-records", "--no-batch-sweep", "--batch", "1", "--height", "512", "--width", "1024",
           "--full-record", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]     # the parent's stdout is rank 0's record, nothing else
    assert len(lines[0]) < 4096
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["ranks_seen"] == 2 and rec["config"]["backend"] == "gloo"
    assert rec["config"]["replicas_identical_after_run"] is True
    assert rec["config"]["rank_devices"] == [0, 0] and rec["config"]["global_batch"] == 2
    assert rec["value"] > 0 and rec["steps"] == 2
