"""Network-level parity: DepthwiseStudent (HIP engine) vs golden vectors captured from the reference itself
(tools/make_golden.py: g_student_step) on identical seeded weights and inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _seeded import sample_idx, seeded_fill_, seeded_input  # noqa: E402


def build_model(plan, dtype):
    import kdcc_amd
    from kdcc_amd.models import DeepWV3Plus
    from kdcc_amd.models.students import DepthwiseStudent
    teacher = DeepWV3Plus(num_classes=19)
    seeded_fill_(teacher, "teacher.")
    teacher.eval()
    model = DepthwiseStudent(teacher, None, dtype=dtype)
    model.replace([{"name": n, "epoch": 1} for n in plan], kernel_size=9, padding=20, dilation=5)
    model.register_hint_layers(plan)
    model.unfreeze(plan)
    for n in plan:
        seeded_fill_(model.get_block(n, model.student), f"student.{n}.")
    return model.cuda()


def check_summary(t, g, key, tol, what, allow_kinks=False):
    """t: NCHW-logical tensor; g[key.*]: the reference's fixed subsample + moments."""
    f = t.detach().float().contiguous().reshape(-1).cpu()
    assert list(t.shape) == [int(v) for v in g[f"{key}.shape"]], what
    ref = g[f"{key}.sample"].astype(np.float64)
    got = f[sample_idx(f.numel())].numpy().astype(np.float64)
    scale = max(np.abs(ref).max(), 1e-12)
    diff = np.abs(got - ref)
    l2 = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
    assert l2 < tol, f"{what}: relative L2 err {l2:.3e} (tol {tol})"
    # element-wise: within tol of the range, except isolated ReLU-kink flips (a pre-activation within fp32 rounding of 0
    # switches d relu/dx between 0 and 1 -- the reference has the same discontinuity against exact arithmetic)
    frac_bad = float((diff > tol * scale).mean())
    assert frac_bad <= (0.002 if allow_kinks else 0.0), f"{what}: {frac_bad:.4%} of samples off by more than {tol} of range"
    assert diff.max() / scale < (5e-2 if allow_kinks else tol), f"{what}: max err {diff.max() / scale:.3e}"
    ssq = float((f.double() ** 2).sum())
    assert abs(ssq - float(g[f"{key}.sumsq"][0])) <= 4 * tol * float(g[f"{key}.sumsq"][0]) + 1e-12, f"{what}: sumsq"


@pytest.fixture(scope="module")
def step_f32(golden):
    from kdcc_amd import losses
    g = golden("student_step_g4")
    plan = [str(s) for s in g["plan"]]
    model = build_model(plan, torch.float32)
    x = seeded_input(str(g["x_key"]), (2, 3, 64, 128)).cuda()
    tgt = torch.from_numpy(g["target"].astype(np.int64)).cuda()
    out_st, out_tc = model(x)
    crit = [losses.CrossEntropyLoss2d(ignore_index=255), losses.KLDivergenceLoss(1), losses.MSELoss(num_classes=1000),
            losses.MSELoss(num_classes=1)]
    res = dict(sup=crit[0](out_st, tgt), kd=crit[1](out_st, out_tc), tl=crit[0](out_tc, tgt), kd_mse=crit[3](out_st, out_tc))
    hint, per = 0, []
    for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        l = crit[2](s, t)
        per.append(l)
        hint = hint + l
    hint.backward()
    torch.cuda.synchronize()
    res.update(hint=hint, per=per, model=model, out_st=out_st, out_tc=out_tc, g=g, plan=plan)
    return res


def test_forward_parity_fp32(step_f32):
    r, g = step_f32, step_f32["g"]
    check_summary(r["out_tc"], g, "teacher_logits", 1e-3, "teacher logits (PyTorch-ROCm)")
    check_summary(r["out_st"], g, "student_logits", 1e-3, "student logits (HIP engine)")
    m = r["model"]
    assert len(m.student_hidden_outputs) == len(r["plan"]) == len(m.teacher_hidden_outputs)
    for i, (s, t) in enumerate(zip(m.student_hidden_outputs, m.teacher_hidden_outputs)):
        check_summary(t, g, f"hint_t{i}", 1e-3, f"teacher hint {i}")
        check_summary(s, g, f"hint_s{i}", 1e-3, f"student hint {i}")


def test_losses_parity_fp32(step_f32):
    r, g = step_f32, step_f32["g"]
    np.testing.assert_allclose(r["hint"].item(), float(g["hint_loss"]), rtol=1e-3)
    np.testing.assert_allclose([p.item() for p in r["per"]], g["per_hint"], rtol=1e-3)
    np.testing.assert_allclose(r["kd"].item(), float(g["kd_loss"]), rtol=1e-3)
    np.testing.assert_allclose(r["kd_mse"].item(), float(g["kd_mse"]), rtol=1e-3)
    np.testing.assert_allclose(r["sup"].item(), float(g["supervised_loss"]), rtol=1e-3)
    np.testing.assert_allclose(r["tl"].item(), float(g["teacher_loss"]), rtol=1e-3)


def test_gradients_parity_fp32(step_f32):
    r, g = step_f32, step_f32["g"]
    names = [str(s) for s in g["trainable"]]
    got = {n: p for n, p in r["model"].student.named_parameters() if p.requires_grad}
    assert sorted(got) == sorted(names)
    for n in names:
        assert got[n].grad is not None, n
        check_summary(got[n].grad, g, f"grad:{n}", 1e-3, f"grad {n}", allow_kinks=True)


def test_bf16_step_tracks_fp32(golden):
    """The measured (bf16) path on the same inputs: loose agreement with the reference, finite gradients."""
    from kdcc_amd import losses
    g = golden("student_step_g4")
    plan = [str(s) for s in g["plan"]]
    model = build_model(plan, torch.bfloat16)
    x = seeded_input(str(g["x_key"]), (2, 3, 64, 128)).cuda()
    out_st, out_tc = model(x)
    crit = losses.MSELoss(num_classes=1000)
    hint = 0
    for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        hint = hint + crit(s, t)
    hint.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(hint.item(), float(g["hint_loss"]), rtol=5e-2)
    ref = g["student_logits.sample"]
    got = out_st.detach().float().contiguous().reshape(-1).cpu()[sample_idx(out_st.numel())].numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 8e-2
    for n, p in model.student.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
            rs = g[f"grad:{n}.sumsq"][0]
            assert abs(float((p.grad.double() ** 2).sum()) - rs) < 0.15 * rs, n


def test_weighted_hint_step_vs_network_oracle():
    """BASELINE config 4 shape: WeightedHintMSELoss feature-hint KD (filter_weight = rand(C) per hint) on a plan / size
    other than the golden's, checked against the network-level CPU oracle (oracle/net_ref.py) in fp32."""
    from kdcc_amd import losses
    from oracle import net_ref
    from _netutil import seeded_cheap_weights, seeded_teacher_sd
    plan = ["mod4.block3.convs.conv2", "mod5.block2.convs.conv2", "aspp.features.3.0"]
    model = build_model(plan, torch.float32)
    x = seeded_input("whint.x", (1, 3, 48, 96))
    ws = [torch.rand(c, generator=torch.Generator().manual_seed(7 + i)) for i, c in enumerate((512, 1024, 256))]
    out_st, out_tc = model(x.cuda())
    crit = losses.WeightedHintMSELoss()
    hint = 0
    for i, (s, t) in enumerate(zip(model.student_hidden_outputs, model.teacher_hidden_outputs)):
        hint = hint + crit(s, t, ws[i].cuda())
    hint.backward()
    torch.cuda.synchronize()
    assert model.student_hint_names == plan
    tsd = seeded_teacher_sd()
    ssd = net_ref.make_student_sd(tsd, plan, seeded_cheap_weights(tsd, plan))
    torch.set_num_threads(8)
    r = net_ref.kd_step(tsd, ssd, x, None, plan, hint_weights=ws)
    np.testing.assert_allclose(hint.item(), r["hint_loss"].item(), rtol=1e-3)
    ref_l, got_l = r["student_logits"].numpy(), out_st.detach().float().cpu().numpy()
    assert np.abs(ref_l - got_l).max() / np.abs(ref_l).max() < 1e-3
    for n, p in model.student.named_parameters():
        if p.requires_grad:
            ref = r["grads"][n].numpy().astype(np.float64)
            got = p.grad.cpu().numpy().astype(np.float64)
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-3, n


def test_torch_teacher_backend_step_matches_reference(golden):
    """The split north_star describes: frozen teacher as a PyTorch-ROCm module (MIOpen) on a side HIP stream with
    record_stream hand-over, student on the HIP engine; same golden as the default (engine-teacher) path."""
    import os
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    from kdcc_amd import losses
    g = golden("student_step_g4")
    plan = [str(s) for s in g["plan"]]
    model = build_model(plan, torch.float32)
    model.teacher_backend = "torch"
    model.overlap_teacher = True
    x = seeded_input(str(g["x_key"]), (2, 3, 64, 128)).cuda()
    out_st, out_tc = model(x)
    assert model._side_stream is not None            # the teacher really ran on the side stream
    crit = losses.MSELoss(num_classes=1000)
    hint = 0
    for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        hint = hint + crit(s, t)
    hint.backward()
    torch.cuda.synchronize()
    check_summary(out_tc, g, "teacher_logits", 1e-3, "teacher logits (PyTorch-ROCm, side stream)")
    check_summary(out_st, g, "student_logits", 1e-3, "student logits")
    assert len(model.teacher_hidden_outputs) == len(plan)
    for i, t in enumerate(model.teacher_hidden_outputs):
        check_summary(t, g, f"hint_t{i}", 1e-3, f"teacher hint {i} (hooked PyTorch module)")
    np.testing.assert_allclose(hint.item(), float(g["hint_loss"]), rtol=1e-3)
    for n, p in model.student.named_parameters():
        if p.requires_grad:
            check_summary(p.grad, g, f"grad:{n}", 1e-3, f"grad {n}", allow_kinks=True)


def test_p92_step_midsize_vs_network_oracle():
    """BASELINE config 2's plan (P92, six cheap-conv blocks incl. all three ASPP branches) forward + hint-loss backward at
    256x512, fp32, against the network-level CPU oracle (oracle/net_ref.py): logits, every hint, loss, every gradient."""
    from kdcc_amd import losses
    from oracle import net_ref
    from _netutil import seeded_cheap_weights, seeded_teacher_sd
    plan = ["mod4.block2.convs.conv2", "mod4.block3.convs.conv1", "mod7.block1.convs.conv2",
            "aspp.features.1.0", "aspp.features.2.0", "aspp.features.3.0"]
    model = build_model(plan, torch.float32)
    x = seeded_input("p92mid.x", (1, 3, 256, 512))
    out_st, out_tc = model(x.cuda())
    crit = losses.MSELoss(num_classes=1000)
    hint = 0
    for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        hint = hint + crit(s, t)
    hint.backward()
    torch.cuda.synchronize()
    tsd = seeded_teacher_sd()
    ssd = net_ref.make_student_sd(tsd, plan, seeded_cheap_weights(tsd, plan))
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    r = net_ref.kd_step(tsd, ssd, x, None, plan)
    assert model.student_hint_names == r["hint_names"]
    np.testing.assert_allclose(hint.item(), r["hint_loss"].item(), rtol=1e-3)

    def relerr(got, ref):
        got, ref = got.detach().float().cpu().numpy().astype(np.float64), ref.numpy().astype(np.float64)
        return np.abs(got - ref).max() / np.abs(ref).max(), np.linalg.norm(got - ref) / np.linalg.norm(ref)
    for name, got, ref in [("student logits", out_st, r["student_logits"]), ("teacher logits", out_tc, r["teacher_logits"])]:
        mx, l2 = relerr(got, ref)
        assert mx < 1e-3 and l2 < 1e-3, (name, mx, l2)
    for i, (s, t) in enumerate(zip(model.student_hidden_outputs, model.teacher_hidden_outputs)):
        assert relerr(s, r["student_hints"][i])[0] < 1e-3, f"student hint {i}"
        assert relerr(t, r["teacher_hints"][i])[0] < 1e-3, f"teacher hint {i}"
    n_grads = 0
    for n, p in model.student.named_parameters():
        if p.requires_grad:
            n_grads += 1
            assert relerr(p.grad, r["grads"][n])[1] < 1e-3, n
    assert n_grads == 12


def test_shared_frozen_prefix_is_bit_identical(golden):
    """Opt-in prefix sharing: the student reuses the teacher's activations of the layers both networks hold bit-identical and
    frozen (stem .. the block before the first cheap conv).  Everything -- logits, hints, every gradient -- must equal the
    two-full-forwards result bit for bit (same kernels on the same inputs), and the prefix must end where the plan starts."""
    from kdcc_amd import losses
    g = golden("student_step_g4")
    plan = [str(s) for s in g["plan"]]
    x = seeded_input(str(g["x_key"]), (2, 3, 64, 128)).cuda()

    def run(share):
        model = build_model(plan, torch.bfloat16)
        model.share_frozen_prefix = share
        out_st, out_tc = model(x)
        crit = losses.MSELoss(num_classes=1000)
        hint = 0
        for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
            hint = hint + crit(s, t)
        hint.backward()
        torch.cuda.synchronize()
        return model, out_st, out_tc, hint
    m0, s0, t0, h0 = run(False)
    m1, s1, t1, h1 = run(True)
    k = m1._student_engine().shareable_prefix(m1._teacher_engine)
    names = [n for n, _ in m1._student_engine()._flat_blocks()]
    assert names[k] == "mod4.block2"          # first block holding a cheap conv (plan g4): mod2, mod3, mod4.block1 are shared
    assert torch.equal(s0, s1) and torch.equal(t0, t1) and torch.equal(h0.detach(), h1.detach())
    for (n0, p0), (n1, p1) in zip(m0.student.named_parameters(), m1.student.named_parameters()):
        if p0.requires_grad:
            assert torch.equal(p0.grad, p1.grad), n0
    # unfreezing a prefix layer or hinting inside it shortens the shared part
    m1.student.mod3.block2.convs.conv1.weight.requires_grad = True
    assert names[m1._student_engine().shareable_prefix(m1._teacher_engine)] == "mod3.block2"


@pytest.mark.parametrize("backend", ["hip", "torch"])
def test_teacher_prefetched_under_the_previous_backward_is_bit_identical(backend):
    """DepthwiseStudent.prefetch_teacher: the frozen teacher's forward for the next batch launched on the side stream before
    loss.backward() (north_star's placement; reference models/students/depthwise_student.py:168-177 runs it inside forward).  Three
    steps with and without it: losses, hints, every gradient and every updated parameter bit-identical, and every step after the first
    took the prefetched outputs."""
    import bench
    from kdcc_amd.utils.optim import RAdam
    plan = bench.PLANS["P92"]
    g = torch.Generator().manual_seed(77)
    xs = [torch.randn((1, 3, 128, 256), generator=g).cuda() for _ in range(3)]
    tg = torch.randint(0, 19, (1, 128, 256), generator=g).cuda()

    def run(prefetch):
        model, crit, opt, _ = bench.build(plan, torch.bfloat16, torch.device("cuda", 0))
        model.teacher_backend = backend      # "torch": hooked PyTorch-ROCm teacher, whose hooks APPEND to teacher_hidden_outputs
        model.prefetch_hits = 0
        out = []
        for i, x in enumerate(xs):
            out_st, out_tc = model(x)
            hint = 0
            for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
                hint = hint + crit[2](s, t)
            kd = crit[1](out_st, out_tc)
            if prefetch and i + 1 < len(xs):
                assert model.prefetch_teacher(xs[i + 1])
            hint.backward()
            grads = [p.grad.detach().clone() for p in model.student.parameters() if p.requires_grad]
            opt.step()
            opt.zero_grad()
            torch.cuda.synchronize()
            out.append((hint.detach().clone(), kd.detach().clone(), [t.detach().clone() for t in model.teacher_hidden_outputs], grads,
                        [p.detach().clone() for p in model.student.parameters() if p.requires_grad]))
        return out, model.prefetch_hits

    ref, hits0 = run(False)
    got, hits1 = run(True)
    assert hits0 == 0 and hits1 == len(xs) - 1
    for a, b in zip(ref, got):
        assert len(a[2]) == len(b[2]) == len(plan)        # one teacher hint per hint site: this batch's, not two batches' worth
        if backend == "torch":
            # MIOpen picks its solvers per process state, so two builds of the PyTorch teacher are compared to bf16 tolerance; a
            # hint list paired with the PREVIOUS batch's teacher hints (the aliasing this guards against) is off by O(1)
            assert torch.allclose(a[0], b[0], rtol=2e-2) and torch.allclose(a[1], b[1], rtol=2e-2, atol=1e-3)
            for u, v in zip(a[2], b[2]):
                assert (u.float() - v.float()).norm() <= 2e-2 * u.float().norm()
            continue
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for u, v in zip(a[2] + a[3] + a[4], b[2] + b[3] + b[4]):
            assert torch.equal(u, v)
    # a different tensor than the one announced: the prefetched outputs are dropped, the teacher runs in forward
    model, crit, opt, _ = bench.build(plan, torch.bfloat16, torch.device("cuda", 0))
    model(xs[0]); model.prefetch_teacher(xs[1]); model(xs[2])
    assert getattr(model, "prefetch_hits", 0) == 0
