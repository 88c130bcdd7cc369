"""Gated-SCNN (BASELINE config 5) on the GPU: shape-stream kernels, the full GSCNN forward through the engine against the
reference's own forward (tests/golden/gscnn.npz: cv2.Canny replaced on both sides by the same seeded 0/255 map -- the
Canny operator itself is parity-unpinned and checked against oracle.canny_ref), and a KD step of the shipped GSCNN plan
against oracle/net_ref.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _netutil import canny_stub_map, gscnn_key_inventory, seeded_cheap_weights, seeded_gscnn_sd  # noqa: E402
from _seeded import sample_idx, seeded_fill_, seeded_input, seeded_value  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from test_modeb_gpu import _check  # noqa: E402


def _gscnn(dtype=torch.float32):
    import kdcc_amd
    from kdcc_amd.models import GSCNN
    net = GSCNN(num_classes=19)
    seeded_fill_(net, "gscnn.")
    for p in net.parameters():      # DepthwiseStudent freezes its teacher / student copies the same way
        p.requires_grad = False
    return net.eval()


def test_gated_conv_and_basic_block_goldens(golden):
    """kd_gated_conv (C = 16) and the BN-folded BasicBlock (16 channels in a 64-channel buffer) vs the reference's modules."""
    from kdcc_amd import ops
    from kdcc_amd.engine import StudentEngine
    from kdcc_amd.models.gscnn import BasicBlock, GatedSpatialConv2d
    g = golden("gscnn")
    gate = GatedSpatialConv2d(16, 16)
    seeded_fill_(gate, "gscnn.blk.gate.")
    gate = gate.eval().cuda()
    eng = StudentEngine(None, torch.float32)
    eng.device = torch.device("cuda")
    f, a = seeded_input("gscnn.blk.gate.f", (2, 16, 12, 20)), seeded_input("gscnn.blk.gate.a", (2, 1, 12, 20))
    buf = torch.zeros((2, 12, 20, 64), device="cuda")
    buf[..., :16] = f.permute(0, 2, 3, 1).cuda()
    y = ops.gated_conv(buf, a.permute(0, 2, 3, 1).contiguous().cuda(), eng._gate_params(gate), 16)
    ref = g["blk_gate.y"]
    assert np.abs(y.cpu().numpy().transpose(0, 3, 1, 2) - ref).max() < 1e-3 * np.abs(ref).max()
    with torch.no_grad():   # the torch container agrees as well (teacher_backend="torch" path)
        assert np.abs(gate(f.cuda(), a.cuda()).cpu().numpy() - ref).max() < 1e-3 * np.abs(ref).max()
    blk = BasicBlock(16)
    seeded_fill_(blk, "gscnn.blk.res.")
    blk = blk.eval().cuda()
    for p in blk.parameters():
        p.requires_grad = False
    xb = seeded_input("gscnn.blk.res.x", (2, 16, 12, 20))
    bx = torch.zeros((2, 12, 20, 64), device="cuda")
    bx[..., :16] = xb.permute(0, 2, 3, 1).cuda()
    yb = eng._basic_block(blk, bx)
    ref = g["blk_res.y"]
    assert np.abs(yb[..., :16].cpu().numpy().transpose(0, 3, 1, 2) - ref).max() < 1e-3 * np.abs(ref).max()
    assert float(yb[..., 16:].abs().max()) == 0.0


@pytest.mark.parametrize("C", [8, 16, 32])
def test_gated_conv_bf16_matrix_core_kernel(C):
    """The bf16 kd_gated_conv (two MFMAs sharing the feature operand, gate_spatial_conv.py:50-60) against the fp32 per-pixel
    kernel on the same bf16-rounded features: ragged last group of 16 pixels, features as a slice of a wider buffer."""
    from kdcc_amd import ops
    from kdcc_amd.engine import StudentEngine
    from kdcc_amd.models.gscnn import GatedSpatialConv2d
    gate = GatedSpatialConv2d(C, C)
    seeded_fill_(gate, f"gscnn.gate{C}.")
    gate = gate.eval().cuda()
    eng = StudentEngine(None, torch.float32)
    eng.device = torch.device("cuda")
    prm = eng._gate_params(gate)
    N, H, W = 2, 13, 21   # 546 pixels: not a multiple of 16
    gen = torch.Generator(device="cuda").manual_seed(3 + C)
    buf = torch.zeros((N, H, W, 64), device="cuda", dtype=torch.bfloat16)
    buf[..., :C] = torch.randn((N, H, W, C), device="cuda", generator=gen).bfloat16()
    side = torch.randn((N, H, W, 1), device="cuda", generator=gen).bfloat16()
    ref = ops.gated_conv(buf.float(), side.float(), prm, C)
    got = ops.gated_conv(buf, side, prm, C)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (N, H, W, C)
    err = (got.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
    assert err < 2e-2, f"C={C}: {err:.3e}"
    dense = ops.gated_conv(buf[..., :C].contiguous(), side, prm, C)   # dense features: same numbers
    assert torch.equal(dense, got)


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 16), (16, 8)])
def test_pointwise_small_vs_reference(cin, cout):
    """kd_pointwise_small (the shape stream's squeezes d1 / d2 / d3, gscnn.py:232-235) against torch's fp32 1x1 conv on the same
    bf16-rounded operands; ragged last pixel group, input as a channel slice."""
    from kdcc_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(cin)
    N, H, W = 2, 9, 29   # 522 pixels: not a multiple of 16
    xb = torch.zeros((N, H, W, cin + 16), device="cuda", dtype=torch.bfloat16)
    xb[..., :cin] = torch.randn((N, H, W, cin), device="cuda", generator=gen).bfloat16()
    w = (torch.randn((cout, cin, 1, 1), device="cuda", generator=gen) / cin ** 0.5)
    b = torch.randn(cout, device="cuda", generator=gen)
    got = ops.pointwise_small(xb[..., :cin], w, b)
    wq = w.bfloat16().float()
    ref = torch.nn.functional.conv2d(xb[..., :cin].float().permute(0, 3, 1, 2), wq, b).permute(0, 2, 3, 1)
    err = (got.float() - ref).abs().max().item() / ref.abs().max().item()
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (N, H, W, cout) and err < 1e-2, err


@pytest.mark.parametrize("C", [16, 32, 64])
@pytest.mark.parametrize("shape", [(1, 16, 256), (2, 37, 45), (1, 20, 300), (1, 33, 513), (1, 3, 7)])
def test_conv3x3_small_vs_oracle(C, shape):
    """kd_conv3x3_small (the 16 / 32-channel BasicBlock convs of the shape stream, Resnet.py:64-99) against the oracle's conv on
    bf16-rounded operands: bias + residual + ReLU, views into wider buffers, ragged segments and row groups."""
    from kdcc_amd import ops
    from kdcc_amd._lib import KD_PACK_FWD
    N, H, W = shape
    rng = np.random.default_rng(100 + C + H + W)
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()
    x, r = bf(rng.standard_normal((N, C, H, W)).astype(np.float32)), bf(rng.standard_normal((N, C, H, W)).astype(np.float32))
    w = bf((rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32))
    b = rng.standard_normal(C).astype(np.float32)
    nhwc = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 3, 1))).to(torch.bfloat16).cuda()
    wp = ops.pack_conv_weight(torch.from_numpy(w).cuda(), torch.bfloat16, KD_PACK_FWD)
    conv = orc.conv2d_fwd(x, w, pad=1)
    # plain, dense
    y = ops.conv3x3_small(nhwc(x), wp, relu=False)
    ref = conv
    assert np.abs(y.float().cpu().numpy().transpose(0, 3, 1, 2) - ref).max() < 1.5e-2 * np.abs(ref).max()
    # bias + residual + relu, input / residual / output as channel slices of 64-channel buffers
    xb = torch.zeros((N, H, W, 80), dtype=torch.bfloat16, device="cuda"); xb[..., :C] = nhwc(x)
    rb = torch.zeros((N, H, W, 80), dtype=torch.bfloat16, device="cuda"); rb[..., 8:8 + C] = nhwc(r)
    ob = torch.full((N, H, W, 80), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.conv3x3_small(xb[..., :C], wp, torch.from_numpy(b).cuda(), res=rb[..., 8:8 + C], relu=True, out=ob[..., :C])
    ref = np.maximum(conv + b[None, :, None, None] + r, 0)
    assert np.abs(ob[..., :C].float().cpu().numpy().transpose(0, 3, 1, 2) - ref).max() < 1.5e-2 * max(np.abs(ref).max(), 1e-6)
    assert float((ob[..., C:] - 7.0).abs().max()) == 0.0   # nothing written outside the slice


def test_basic_block_small_channels_bf16(golden):
    """The bf16 engine path of res2 / res3: BasicBlock on kd_conv3x3_small (dense 16 channels in, zero-padded 64-channel buffer
    out) against the reference module's golden, and the zero pad that the next 1x1 conv relies on, over two calls (the padded
    buffer persists)."""
    from kdcc_amd.engine import StudentEngine
    from kdcc_amd.models.gscnn import BasicBlock
    g = golden("gscnn")
    eng = StudentEngine(None, torch.bfloat16)
    eng.device = torch.device("cuda")
    blk = BasicBlock(16)
    seeded_fill_(blk, "gscnn.blk.res.")
    blk = blk.eval().cuda()
    for p in blk.parameters():
        p.requires_grad = False
    xb = seeded_input("gscnn.blk.res.x", (2, 16, 12, 20)).permute(0, 2, 3, 1).contiguous().cuda().bfloat16()
    ref = g["blk_res.y"]
    for _ in range(2):
        yb = eng._basic_block(blk, xb)
        assert tuple(yb.shape) == (2, 12, 20, 64)
        assert np.abs(yb[..., :16].float().cpu().numpy().transpose(0, 3, 1, 2) - ref).max() < 2e-2 * np.abs(ref).max()
        assert float(yb[..., 16:].abs().max()) == 0.0


def test_canny_kernel_vs_published_algorithm():
    """kd_canny == oracle.canny_ref (numpy restatement of cv2.Canny's documented algorithm), bit for bit, on images with real
    structure (blurred blobs + noise: long weak chains exercise the hysteresis rounds) and on the normalised-float regime the
    reference actually feeds it (values around 0 cast to uint8)."""
    from kdcc_amd import ops
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:96, 0:160]
    base = 120 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 25 * np.sin((xx + 2 * yy) / 3.0)
    img = np.stack([base + rng.normal(0, s, base.shape) for s in (2.0, 6.0, 12.0)], 0)        # (3, H, W) floats in ~[0, 255]
    batch = np.stack([img, rng.normal(0, 1.2, img.shape) * 3.0]).astype(np.float32)           # second image: "normalised" input
    got = ops.canny(torch.from_numpy(batch).cuda(), 10, 100, sweeps=2).cpu().numpy()
    for n in range(2):
        u8 = (batch[n].transpose(1, 2, 0).astype(np.int64) & 0xFF).astype(np.uint8)           # the kernel's (and numpy's) cast
        ref = orc.canny_ref(u8, 10, 100)
        assert set(np.unique(got[n])) <= {0.0, 255.0}
        assert np.array_equal(got[n].astype(np.uint8), ref), f"image {n}: {(got[n] != ref).sum()} pixels differ"
    assert got[0].sum() > 0


def test_edge_attention_and_edge_aspp():
    from kdcc_amd import ops
    rng = np.random.default_rng(9)
    cs = rng.standard_normal((2, 10, 14, 8)).astype(np.float32)
    canny = (rng.random((2, 10, 14)) < 0.2).astype(np.float32) * 255
    w = rng.standard_normal(10).astype(np.float32) * 0.3
    w[9] *= 0.02
    buf = torch.zeros((2, 10, 14, 64), device="cuda")
    buf[..., :8] = torch.from_numpy(cs).cuda()
    acts = ops.edge_attention(buf, torch.from_numpy(canny).cuda(), torch.from_numpy(w).cuda())
    sig = lambda v: 1 / (1 + np.exp(-v))
    ref = sig(w[8] * sig(cs @ w[:8]) + w[9] * canny)
    np.testing.assert_allclose(acts.cpu().numpy(), ref, rtol=1e-4, atol=1e-6)
    wc, sc, sh = rng.standard_normal(16).astype(np.float32), rng.random(16).astype(np.float32) + 0.5, rng.standard_normal(16).astype(np.float32) * 0.1
    out = torch.zeros((2, 4, 6, 48), device="cuda")
    ops.edge_aspp(acts, torch.from_numpy(wc).cuda(), torch.from_numpy(sc).cuda(), torch.from_numpy(sh).cuda(), out[..., 16:32])
    e = orc.upsample_bilinear_ac(ref[:, None].astype(np.float32), (4, 6))[:, 0]
    exp = np.maximum(e[..., None] * wc * sc + sh, 0)
    np.testing.assert_allclose(out[..., 16:32].cpu().numpy(), exp, rtol=1e-4, atol=1e-5)
    assert float(out[..., :16].abs().max()) == 0 and float(out[..., 32:].abs().max()) == 0


def test_gscnn_forward_matches_reference(golden):
    """GSCNN(19) through the engine (fp32 parity mode) == the reference's forward: logits, edge attention, ASPP output."""
    from kdcc_amd.engine import StudentEngine
    g = golden("gscnn")
    net = _gscnn().cuda()
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == gscnn_key_inventory()
    maps = torch.stack([canny_stub_map((64, 128), int(s)) for s in g["canny_seeds"]]).cuda()
    net.canny_fn = lambda x: maps
    x = seeded_input(str(g["x_key"]), (2, 3, 64, 128), scale=float(g["x_scale"])).cuda()
    eng = StudentEngine(net, torch.float32)
    eng.hint_names = []
    with torch.no_grad():
        logits, _ = eng.forward(x)
        tape = eng._tape
    _check(logits.permute(0, 3, 1, 2), g, "logits", 4096, 1e-3, "GSCNN logits")
    _check(tape["acts"].unsqueeze(1), g, "acts", 4096, 1e-3, "edge attention")
    _check(tape["aspp"]["cat"].permute(0, 3, 1, 2), g, "aspp", 4096, 1e-3, "edge-aware ASPP output")
    # the torch container (teacher_backend="torch") agrees too
    net.canny_fn = lambda x: maps.unsqueeze(1)
    with torch.no_grad():
        _check(net(x), g, "logits", 4096, 1e-3, "GSCNN logits (torch container)")


def test_gscnn_kd_step_vs_network_oracle():
    """The shipped GSCNN plan's shape (cfg/cityscapes/51M_gscnn_all.json: cheap convs in mod4 / mod7 / ASPP, hints = plan,
    loss = hint MSE): DepthwiseStudent(GSCNN) forward + backward vs oracle/net_ref.py (gscnn_forward), fp32."""
    from kdcc_amd import losses
    from kdcc_amd.models.students import DepthwiseStudent
    from oracle import net_ref
    plan = ["mod4.block2.convs.conv2", "mod4.block3.convs.conv1", "mod7.block1.convs.conv2", "aspp.features.1.0", "aspp.features.3.0"]
    teacher = _gscnn()
    model = DepthwiseStudent(teacher, None, dtype=torch.float32)
    model.replace([{"name": n, "epoch": 1} for n in plan], kernel_size=9, padding=20, dilation=5)
    model.register_hint_layers(plan)
    model.unfreeze(plan)
    for n in plan:
        seeded_fill_(model.get_block(n, model.student), f"student.{n}.")
    model = model.cuda()
    maps = torch.stack([canny_stub_map((64, 128), 700 + i) for i in range(2)])
    for net in (model.teacher, model.student):
        net.canny_fn = lambda x: maps.cuda()
    x = seeded_input("gscnn.step.x", (2, 3, 64, 128), scale=30.0)
    out_st, out_tc = model(x.cuda())
    crit = losses.MSELoss(num_classes=1000)
    hint = 0
    for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        hint = hint + crit(s, t)
    hint.backward()
    torch.cuda.synchronize()
    tsd = seeded_gscnn_sd()
    ssd = net_ref.make_student_sd(tsd, plan, seeded_cheap_weights(tsd, plan))
    torch.set_num_threads(8)
    r = net_ref.kd_step(tsd, ssd, x, None, plan, canny=maps.unsqueeze(1))
    np.testing.assert_allclose(hint.item(), r["hint_loss"].item(), rtol=1e-3)
    for name, got, ref in [("student logits", out_st, r["student_logits"]), ("teacher logits", out_tc, r["teacher_logits"])]:
        d = np.abs(got.detach().float().cpu().numpy() - ref.numpy()).max() / np.abs(ref.numpy()).max()
        assert d < 1e-3, (name, d)
    for n, p in model.student.named_parameters():
        if p.requires_grad:
            ref = r["grads"][n].numpy().astype(np.float64)
            got = p.grad.cpu().numpy().astype(np.float64)
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-3, n


def test_gscnn_mode_b_through_the_shape_stream_matches_reference(golden):
    """Mode B on Gated-SCNN: every student parameter trainable -- trunk, ASPP with the edge branch, decoder and the whole shape
    stream (dsn3/4/7, res1-3, d1-3, gate1-3, fuse, cw) -- with hints on a block conv, an ASPP branch conv and the `aspp` module,
    loss = KLDivergenceLoss(1) + hint MSEs: every one of the 203 gradients against the reference's own DepthwiseStudent(GSCNN) run
    (tests/golden/gscnn_step_full.npz; cv2.Canny replaced by the same seeded maps on both sides), fp32."""
    from kdcc_amd import losses
    from kdcc_amd.models.students import DepthwiseStudent
    from test_student_gpu import check_summary
    g = golden("gscnn_step_full")
    plan, hints = [str(s) for s in g["plan"]], [str(s) for s in g["hints"]]
    teacher = _gscnn()
    model = DepthwiseStudent(teacher, None, dtype=torch.float32)
    model.replace([{"name": n, "epoch": 1} for n in plan], kernel_size=9, padding=20, dilation=5)
    for n in plan:
        seeded_fill_(model.get_block(n, model.student), f"student.{n}.")
    model.register_hint_layers(hints)
    for p in model.student.parameters():
        p.requires_grad = True
    model.logits_need_grad = True
    model = model.cuda()
    maps = torch.stack([canny_stub_map((64, 128), int(s)) for s in g["canny_seeds"]]).cuda()
    for net in (model.teacher, model.student):
        net.canny_fn = lambda x: maps
    x = seeded_input(str(g["x_key"]), (2, 3, 64, 128), scale=float(g["x_scale"])).cuda()
    out_st, out_tc = model(x)
    kd = losses.KLDivergenceLoss(1)(out_st, out_tc)
    crit = losses.MSELoss(num_classes=1000)
    hint, per = 0, []
    for s_, t_ in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        l = crit(s_, t_)
        per.append(l.item())
        hint = hint + l
    (kd + hint).backward()
    torch.cuda.synchronize()
    assert model.student_hint_names == hints
    _check(out_st, g, "student_logits", 1024, 1e-3, "GSCNN student logits")
    np.testing.assert_allclose(kd.item(), float(g["kd_loss"]), rtol=1e-3)
    np.testing.assert_allclose(per, g["per_hint"], rtol=1e-3)
    names = [str(s) for s in g["trainable"]]
    got = {n: p for n, p in model.student.named_parameters() if p.grad is not None}
    assert sorted(got) == sorted(names) and len(names) == 203           # everything but the unused dsn1 (gscnn.py:269-314)
    bad = []
    for n in names:
        ref, gr = g[f"grad:{n}.sample"].astype(np.float64), got[n].grad.detach().float().reshape(-1).cpu()
        smp = gr[sample_idx(gr.numel(), 256)].numpy().astype(np.float64)
        err = np.linalg.norm(smp - ref) / max(np.linalg.norm(ref), 1e-30)
        ssq = float((gr.double() ** 2).sum())
        if err >= 2e-3 or abs(ssq - float(g[f"grad:{n}.sumsq"][0])) > 8e-3 * float(g[f"grad:{n}.sumsq"][0]) + 1e-30:
            bad.append((n, err, ssq, float(g[f"grad:{n}.sumsq"][0])))
    assert not bad, bad[:12]
    # a logit loss on a model that was not told to keep the shape stream's intermediates is refused, not silently wrong
    from kdcc_amd.engine import EngineError
    model.logits_need_grad = False
    model.register_hint_layers(plan)
    with pytest.raises(EngineError):                      # trainable shape-stream parameters: refused at the forward
        model(x)
    for name in ("dsn3", "dsn4", "dsn7", "res1", "res2", "res3", "d1", "d2", "d3", "gate1", "gate2", "gate3", "fuse", "cw"):
        for p in getattr(model.student, name).parameters():
            p.requires_grad = False
    out_st, out_tc = model(x)
    with pytest.raises(EngineError):                      # frozen stream, but a loss on the logits: refused at the backward
        losses.KLDivergenceLoss(1)(out_st, out_tc).backward()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_shape_stream_backward_pieces_vs_autograd(dt):
    """kd_small_linear / kd_small_wgrad / kd_gate_mix_bwd / kd_edge_attention_bwd / kd_rank1_add and the non-aligned bilinear
    transpose against torch autograd of the reference's own expressions (gate_spatial_conv.py:50-60, gscnn.py:308-314, :323)."""
    import torch.nn.functional as F
    from kdcc_amd import ops
    g0 = torch.Generator().manual_seed(5)
    tol = 2e-5 if dt == torch.float32 else 2e-2
    rnd = lambda *s: torch.randn(*s, generator=g0)

    def close(got, ref, what, t=tol):
        got, ref = got.detach().float().cpu().double(), ref.detach().double()
        assert float((got - ref).norm() / ref.norm().clamp_min(1e-30)) < t, what
    # a 1x1 map with odd channel counts, bias, ReLU, a channel-sliced input view and accumulation
    x, w, b = rnd(2, 5, 7, 40), rnd(33, 33) * 0.2, rnd(33) * 0.1
    xd = x.to(dt).cuda()
    xin = xd[..., 3:36]
    ref = F.relu(xin.float().cpu() @ w.t() + b)
    close(ops.small_linear(xin, w.cuda(), b.cuda(), relu=True, out_dtype=torch.float32), ref, "small_linear")
    base = rnd(2, 5, 7, 33)
    acc = base.clone().cuda()
    mask = rnd(2, 5, 7, 33)
    ops.small_linear(xin, w.cuda(), None, out=acc, accumulate=True, mask=mask.cuda())
    close(acc, base + torch.where(mask > 0, xin.float().cpu() @ w.t(), torch.zeros(())), "small_linear accumulate + mask")
    # weight / bias gradient over a pixel count that is not a multiple of the staging chunk
    a, bb = rnd(3, 17, 29, 24).to(dt), rnd(3, 17, 29, 9).to(dt)
    dw, db = ops.small_wgrad(a.cuda(), bb.cuda(), want_bias=True)
    close(dw, torch.einsum("nhwb,nhwa->ba", bb.float(), a.float()), "small_wgrad", 1e-4 if dt == torch.float32 else tol)
    close(db, bb.float().sum((0, 1, 2)), "small_wgrad bias", 1e-4 if dt == torch.float32 else tol)
    # the gate's tail
    feat, al, gv = rnd(2, 6, 5, 16).to(dt), rnd(2, 6, 5), rnd(2, 6, 5, 16)
    fr, ar = feat.float().clone().requires_grad_(True), al.clone().requires_grad_(True)
    v_ref = fr * (torch.sigmoid(ar).unsqueeze(3) + 1)
    (v_ref * gv).sum().backward()
    gfeat, ga, v = ops.gate_mix_bwd(feat.cuda(), al.cuda(), gv=gv.cuda(), want_v=True)
    close(v, v_ref, "gate mix v"); close(gfeat, fr.grad, "gate mix d feat"); close(ga, ar.grad, "gate mix d a")
    # edge attention
    cs, canny, wts, g = rnd(2, 9, 11, 8).to(dt), (torch.rand(2, 9, 11, generator=g0) < 0.2).float() * 255, rnd(10) * 0.3, rnd(2, 9, 11)
    wts[9] *= 0.01
    cr, wr = cs.float().clone().requires_grad_(True), wts.clone().requires_grad_(True)
    eo = torch.sigmoid((cr * wr[:8]).sum(3))
    acts = torch.sigmoid(wr[8] * eo + wr[9] * canny)
    (acts * g).sum().backward()
    g_t, g_s, eoc = ops.edge_attention_bwd(cs.cuda(), canny.cuda(), wts.cuda(), g.cuda())
    close(ops.small_linear(g_s.unsqueeze(3), wts[:8].view(8, 1).cuda()), cr.grad, "edge attention d cs")
    close(ops.small_wgrad(cs.cuda(), g_s.unsqueeze(3))[0].reshape(-1), wr.grad[:8], "d fuse")
    close(ops.small_wgrad(eoc, g_t.unsqueeze(3))[0].reshape(-1), wr.grad[8:], "d cw")
    # rank-1 update and the non-aligned bilinear transpose
    y, gp, wv = rnd(2, 4, 6, 32).to(dt), rnd(2, 4, 6), rnd(32)
    yd = y.clone().cuda()
    ops.rank1_add(yd, gp.cuda().reshape(-1), wv.cuda())
    close(yd, y.float() + gp.unsqueeze(3) * wv, "rank1_add")
    for align in (False, True):
        for (hin, win, ho, wo) in ((6, 10, 12, 20), (16, 24, 5, 7)):      # enlarging (the logits) and shrinking (the edge map)
            src = rnd(2, 3, hin, win).requires_grad_(True)
            gy = rnd(2, 3, ho, wo)
            (F.interpolate(src, size=(ho, wo), mode="bilinear", align_corners=align) * gy).sum().backward()
            got = ops.upsample_bilinear_ac_bwd(gy.permute(0, 2, 3, 1).contiguous().cuda(), (hin, win), out_dtype=torch.float32,
                                               align_corners=align)
            close(got.permute(0, 3, 1, 2), src.grad, f"bilinear transpose align={align} {hin}x{win}->{ho}x{wo}", 2e-5)
