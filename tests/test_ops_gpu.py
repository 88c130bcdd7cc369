"""Per-kernel parity: HIP path (through the C-ABI) vs the CPU oracle, seeded inputs.

fp32 tolerance 1e-3 relative (north_star's bar; most kernels hold 1e-5);
bf16 kernels are compared against the oracle evaluated on the bf16-rounded
inputs with a 1.5e-2 relative-to-range tolerance (bf16 has 8 significant bits).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def K():
    import kdcc_amd
    from kdcc_amd import ops
    assert torch.cuda.is_available()
    return ops


def selected(want, what=""):
    """The dispatcher's choice for the launch just made (kd_debug_last_kernel): a case that says it covers a kernel must reach
    it -- an edit of the selection rules (conv_igemm.hip kd_conv2d_fwd, dwconv_mfma.hip) then fails here instead of silently
    moving the coverage."""
    from kdcc_amd import _lib
    got = _lib.last_kernel()
    assert got == want, f"{what}: dispatched to {got}, this case is meant to cover {want}"


DT = {"f32": torch.float32, "bf16": torch.bfloat16}
RNG = np.random.default_rng(1234)


def rnd(*shape, scale=1.0):
    return (RNG.standard_normal(shape) * scale).astype(np.float32)


def q(a, dt):
    """Round a numpy array to the kernel's storage dtype (so the oracle sees identical inputs)."""
    return torch.from_numpy(np.ascontiguousarray(a)).to(DT[dt]).float().numpy()


def dev_nhwc(a_nchw, dt, ld=None):
    t = torch.from_numpy(np.ascontiguousarray(a_nchw.transpose(0, 2, 3, 1))).to(DT[dt]).cuda()
    if ld is None:
        return t
    N, H, W, C = t.shape
    buf = torch.zeros((N, H, W, ld), dtype=t.dtype, device="cuda")
    buf[..., 8:8 + C] = t
    return buf[..., 8:8 + C]


def host_nchw(t):
    return t.float().cpu().numpy().transpose(0, 3, 1, 2)


def assert_close(got, ref, dt, what=""):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = max(np.abs(ref).max(), 1e-6)
    tol = 1e-3 if dt == "f32" else 1.5e-2
    err = np.abs(got - ref).max() / scale
    assert err < tol, f"{what}: max err {err:.3e} of range (tol {tol})"
    # the max-abs bound catches wrong taps / halos; a small systematic bias (a dropped rounding term, a scale applied twice to a
    # few channels) hides under it, so bound the relative L2 error too: one bf16 output rounding is ~1.1e-3 rms
    l2 = np.sqrt(((got - ref) ** 2).sum() / max((ref ** 2).sum(), 1e-30))
    tol2 = 2e-4 if dt == "f32" else 5e-3
    assert l2 < tol2, f"{what}: relative L2 error {l2:.3e} (tol {tol2})"


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 8, 16, 64, 128, 1, 1, 0, 1),
    (1, 12, 16, 64, 64, 3, 1, 1, 1),
    (2, 9, 11, 128, 64, 3, 1, 2, 2),     # ragged M, dilation 2
    (1, 16, 16, 64, 128, 3, 2, 1, 1),    # stride 2 (mod4.block1)
    (1, 10, 12, 128, 19, 3, 1, 4, 4),    # odd Cout (classifier-like), dilation 4
    (1, 6, 8, 320, 48, 1, 1, 0, 1),      # Cout 48 (bot_fine-like), Cin 320
    # >= 224 wide tiles whose 256-pixel M tiles are image-row segments: the row-buffer kernel (3x3, stride 1, 'same')
    (2, 112, 256, 64, 256, 3, 1, 1, 1),
    (1, 112, 512, 64, 256, 3, 1, 2, 2),  # two tiles per image row: real column halos
    (1, 224, 256, 128, 256, 3, 1, 12, 12),   # ASPP-like dilation, kernel rows leaving the image at top and bottom
    (1, 224, 256, 64, 256, 3, 1, 36, 36),    # ASPP rate 36: the 384-row buffer (all 160 KiB of LDS)
    (1, 12, 256, 64, 128, 3, 1, 1, 1),       # narrow row-buffer kernel (bf16; fp32 takes the gather kernel)
    (1, 10, 512, 128, 64, 3, 1, 2, 2),
    # gathered-A 256x256 tiles (CfgWide / CfgWideF): Cout > 128 and >= 224 wide tiles, 1x1 / stride-2 / W % 256 != 0
    (1, 112, 512, 64, 256, 1, 1, 0, 1),      # 1x1: 224 x 1 tiles, K = one 128-B stage (bf16) / two (f32)
    (2, 57, 509, 128, 512, 1, 1, 0, 1),      # 1x1: ragged last M tile, two N tiles, several K stages
    (1, 224, 1024, 64, 256, 3, 2, 1, 1),     # stride-2 3x3 (the mod4.block1 class) on wide tiles
    (1, 120, 480, 64, 320, 3, 1, 2, 2),      # 3x3 whose rows are not tile segments: gathered im2col with border taps, ragged Cout
]
# what each of the shape-selected cases above must dispatch to: index -> (bf16 kernel, f32 kernel)
CONV_SELECTS = {
    0: ("conv_igemm_kernel<narrow2>", "conv_igemm_kernel<f32,narrow>"),
    6: ("conv_row_lw_kernel", "conv_igemm_row_kernel<f32,wide>"),
    7: ("conv_row_lw_kernel", "conv_igemm_row_kernel<f32,wide>"),
    8: ("conv_row_lw_kernel", "conv_igemm_row_kernel<f32,wide>"),
    9: ("conv_igemm_row_kernel<x>", "conv_igemm_row_kernel<f32,x>"),
    10: ("conv_igemm_row_kernel<narrow>", "conv_igemm_kernel<f32,narrow>"),
    11: ("conv_igemm_row_kernel<narrow>", "conv_igemm_kernel<f32,narrow>"),
    12: ("conv_igemm_persist_kernel<pp>", "conv_igemm_kernel<f32,wide>"),
    13: ("conv_igemm_kernel<wide>", "conv_igemm_kernel<f32,wide>"),
    14: ("conv_igemm_kernel<wide>", "conv_igemm_kernel<f32,wide>"),
    15: ("conv_igemm_kernel<wide>", "conv_igemm_kernel<f32,wide>"),
}


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(K, dt, case):
    N, H, W, Cin, Cout, k, s, p, d = case
    x = q(rnd(N, Cin, H, W), dt)
    w = q(rnd(Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5), dt)
    ref = orc.conv2d_fwd(x, w, stride=s, pad=p, dil=d)
    xd = dev_nhwc(x, dt)
    wp = K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt])
    Ho, Wo = ref.shape[2:]
    out = torch.empty((N, Ho, Wo, Cout), dtype=DT[dt], device="cuda")
    K.conv2d(xd, wp, s, p, d, out_raw=out)
    idx = CONV_CASES.index(case)
    if idx in CONV_SELECTS:
        selected(CONV_SELECTS[idx][0 if dt == "bf16" else 1], f"conv {case} {dt}")
    assert_close(host_nchw(out), ref, dt, f"conv {case}")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_epilogue_forward(K, dt):
    """shortcut add + raw/act dual output with folded BN + ReLU, reading/writing channel slices."""
    N, H, W, Cin, Cout = 2, 8, 8, 64, 64
    x, w = q(rnd(N, Cin, H, W), dt), q(rnd(Cout, Cin, 3, 3, scale=0.06), dt)
    res = q(rnd(N, Cout, H, W), dt)
    scale, shift = rnd(Cout) * 0.2 + 1.0, rnd(Cout) * 0.3
    conv = orc.conv2d_fwd(x, w, pad=1)
    raw_ref = conv + res
    act_ref = np.maximum(raw_ref * scale[None, :, None, None] + shift[None, :, None, None], 0)
    xd = dev_nhwc(x, dt, ld=Cin + 16)           # input is a slice of a wider buffer
    wide = torch.zeros((N, H, W, 3 * Cout), dtype=DT[dt], device="cuda")
    out_act = wide[..., Cout:2 * Cout]           # concat-free write into a slice
    out_raw = torch.empty((N, H, W, Cout), dtype=DT[dt], device="cuda")
    wp = K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt])
    K.conv2d(xd, wp, 1, 1, 1, res_pre=dev_nhwc(res, dt), out_raw=out_raw, out_act=out_act,
             act_scale=torch.from_numpy(scale).cuda(), act_shift=torch.from_numpy(shift).cuda(), act_relu=True)
    assert_close(host_nchw(out_raw), raw_ref, dt, "raw")
    assert_close(host_nchw(out_act), act_ref, dt, "act")
    assert float(wide[..., :Cout].abs().max()) == 0 and float(wide[..., 2 * Cout:].abs().max()) == 0


BN_SUMS_CASES = [
    # N, H, W, Cin, Cout, k, dil, operands, kernel -- the backward of conv -> BN -> ReLU at shapes that select each kernel whose
    # epilogue can take the eval-BN parameter sums (mask alone, mask + res_post, res_pre + mask), and one that selects none
    (1, 128, 512, 256, 256, 3, 1, "m", "conv_row_lw_kernel"),
    (2, 96, 256, 256, 512, 3, 2, "mq", "conv_row_lw_kernel"),
    (1, 128, 512, 128, 256, 1, 1, "pm", "conv_igemm_persist_kernel<pp>"),
    (1, 128, 512, 256, 256, 1, 1, "mq", "conv_igemm_persist_kernel<pp>"),
    (1, 256, 512, 128, 128, 3, 1, "m", "conv_row_tall_kernel"),
    (1, 256, 512, 64, 128, 3, 1, "mq", "conv_row_tall_kernel"),
    (1, 64, 256, 128, 64, 3, 1, "m", None),
]


@pytest.mark.parametrize("case", BN_SUMS_CASES)
def test_conv_epilogue_bn_sums(K, case):
    """kd_conv2d_fwd(bn_sums): the sums eval-mode BN's weight / bias gradients need (autograd of bn -> relu in
    IdentityResidualBlock, wider_resnet.py:124-167), taken in the epilogue of the input-gradient conv, against kd_channel_sums
    of the stored result and against fp64 sums of the same expression."""
    N, H, W, Cin, Cout, k, d, opnds, kernel = case
    dt = "bf16"
    x = dev_nhwc(q(rnd(N, Cin, H, W), dt), dt)
    wp = K.pack_conv_weight(torch.from_numpy(q(rnd(Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5), dt)).cuda(), DT[dt])
    act = dev_nhwc(q(np.maximum(rnd(N, Cout, H, W), 0), dt), dt)         # relu(bn(x)): zero where the unit is off
    scale = torch.from_numpy(rnd(Cout) * 0.2 + 1.0).cuda()
    res_pre = dev_nhwc(q(rnd(N, Cout, H, W), dt), dt) if "p" in opnds else None
    res_post = dev_nhwc(q(rnd(N, Cout, H, W), dt), dt) if "q" in opnds else None
    pad = d if k == 3 else 0
    out = torch.empty((N, H, W, Cout), dtype=DT[dt], device="cuda")
    got = []
    K.conv2d(x, wp, 1, pad, d, res_pre=res_pre, mask=act, mask_scale=scale, res_post=res_post, out_raw=out, bn_sums=got)
    if kernel is None:
        assert got == [], "a kernel without the fused sums reported them"
        return
    selected(kernel, f"bn_sums {case}")
    assert len(got) == 1
    s1, s2 = got[0]
    # the same launch without the sums: the stored gradient must not change
    out2 = torch.empty_like(out)
    K.conv2d(x, wp, 1, pad, d, res_pre=res_pre, mask=act, mask_scale=scale, res_post=res_post, out_raw=out2)
    assert torch.equal(out, out2)
    c1, c2 = K.channel_sums(out, sub=res_post, a=act)
    # fp64 restatement from the unmasked conv: g = (act > 0) * (conv + res_pre) * scale
    raw = torch.empty_like(out)
    K.conv2d(x, wp, 1, pad, d, res_pre=res_pre, out_raw=raw)
    g = torch.where(act > 0, raw.double() * scale.double(), torch.zeros((), dtype=torch.float64, device="cuda"))
    r1, r2 = g.sum((0, 1, 2)), (g * act.double()).sum((0, 1, 2))
    for name, a, b, ref in (("s1", s1, c1, r1), ("s2", s2, c2, r2)):
        norm = float(ref.abs().max()) + 1e-6
        # the stored gradient is rounded to bf16 before kd_channel_sums reads it; the fused sums see fp32 values of bf16 `raw`
        assert float((a.double() - ref).abs().max()) / norm < 4e-3, f"{name} fused vs fp64 {case}"
        assert float((a.double() - b.double()).abs().max()) / norm < 8e-3, f"{name} fused vs kd_channel_sums {case}"
    # run to run
    again = []
    K.conv2d(x, wp, 1, pad, d, res_pre=res_pre, mask=act, mask_scale=scale, res_post=res_post, out_raw=out2, bn_sums=again)
    assert torch.equal(again[0][0], s1) and torch.equal(again[0][1], s2)


WIDE_EPI_CASES = [
    # N, H, W, Cin, Cout, k, pad, dil -- every one selects a 256 x 256 tile config (>= 224 wide tiles, Cout > 128)
    (1, 112, 512, 64, 256, 1, 0, 1),      # gathered 1x1 (CfgWide, pipelined loop in bf16)
    (1, 112, 512, 64, 256, 3, 1, 1),      # row-buffer 3x3 (CfgRow)
    (1, 120, 480, 64, 256, 3, 2, 2),      # gathered 3x3 (CfgWide)
    # more tiles than CUs: the persistent bf16 kernels walk two tiles per workgroup on a quarter of the chip (next tile's
    # first stages issued before the epilogue, counted wait that leaves the epilogue's stores outstanding)
    (1, 160, 512, 64, 256, 1, 0, 1),      # 320 tiles, gathered
    (1, 80, 512, 64, 512, 3, 1, 1),       # 160 x 2 tiles, row buffers, two N tiles
    (1, 121, 480, 64, 256, 3, 2, 2),      # M % 256 != 0: ragged last M tile, one-tile-per-workgroup kernel
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("raw_f32", [False, True])
@pytest.mark.parametrize("case", WIDE_EPI_CASES)
def test_conv_wide_tile_full_epilogue(K, dt, raw_f32, case):
    """The wide-tile epilogue (bf16: the batched EB=4 operand loads) with every operand set at once:
    v = acc + res_pre; v = mask > 0 ? v * mask_scale : 0; v += res_post; out_raw = v (bf16 or fp32);
    out_act = relu(v * act_scale + act_shift) -- operands and outputs are channel slices of wider buffers."""
    N, H, W, Cin, Cout, k, p, d = case
    if raw_f32 and dt == "f32":
        pytest.skip("raw_f32 only differs from the plain output on the bf16 path")
    x = q(rnd(N, Cin, H, W), dt)
    w = q(rnd(Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5), dt)
    pre, post = q(rnd(N, Cout, H, W), dt), q(rnd(N, Cout, H, W), dt)
    mask = q(np.maximum(rnd(N, Cout, H, W), 0), dt)
    mscale, ascale, ashift = rnd(Cout) * 0.2 + 1.0, rnd(Cout) * 0.2 + 1.0, rnd(Cout) * 0.3
    bc = lambda v: v[None, :, None, None]
    raw_ref = np.where(mask > 0, (orc.conv2d_fwd(x, w, pad=p, dil=d) + pre) * bc(mscale), 0.0) + post
    act_ref = np.maximum(raw_ref * bc(ascale) + bc(ashift), 0)
    wp = K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt])
    out_raw = torch.zeros((N, H, W, Cout + 32), dtype=torch.float32 if raw_f32 else DT[dt], device="cuda")[..., 16:16 + Cout]
    out_act = torch.zeros((N, H, W, 2 * Cout), dtype=DT[dt], device="cuda")[..., Cout:]
    cu = lambda v: torch.from_numpy(v).cuda()
    K.conv2d(dev_nhwc(x, dt), wp, 1, p, d, res_pre=dev_nhwc(pre, dt, ld=Cout + 16), mask=dev_nhwc(mask, dt),
             mask_scale=cu(mscale), res_post=dev_nhwc(post, dt, ld=Cout + 24), out_raw=out_raw, out_act=out_act,
             act_scale=cu(ascale), act_shift=cu(ashift), act_relu=True)
    assert_close(host_nchw(out_raw), raw_ref, dt, f"raw {case}")
    # the activated output is derived from the unrounded value; compare against the oracle chain directly
    assert_close(host_nchw(out_act), act_ref, dt, f"act {case}")


PERSIST_CASES = [
    # (N, H, W, Cin, Cout, k, pad, dil), operands, outputs -- bf16 persistent kernels: more tiles than CUs, one or two epilogue
    # operands (the template variants), both tile orders (Cout = 2048: N tiles walked four at a time)
    ((1, 160, 512, 64, 256, 1, 0, 1), ("pre",), ("raw", "act")),
    ((1, 160, 512, 64, 256, 1, 0, 1), ("mask", "post"), ("raw",)),
    ((1, 80, 512, 64, 512, 3, 1, 1), ("mask",), ("raw",)),
    ((1, 80, 512, 64, 512, 3, 2, 2), ("pre", "post"), ("act",)),
    ((1, 80, 512, 128, 512, 3, 1, 1), ("pre", "mask"), ("raw", "act")),
    ((1, 20, 512, 64, 2048, 1, 0, 1), ("post",), ("raw", "act")),
    ((1, 20, 512, 64, 2048, 1, 0, 1), (), ("act",)),
    ((1, 160, 512, 64, 256, 1, 0, 1), ("pre", "mask", "post"), ("raw",)),          # three operands (dgrad: mask + both residuals)
    ((1, 80, 512, 64, 512, 3, 2, 2), ("pre", "mask", "post"), ("raw", "act")),
    # dil < H < 2 dil (ASPP rate 24 on a 256 x 2048 crop): output rows H - dil <= ho < dil see ONE kernel row, which the lone-wave loop's
    # period hand-over does not model -> the ping-pong kernel must take the launch (round-4 advisor finding); H = 2 dil is the first lone-wave size
    ((8, 32, 256, 64, 256, 3, 24, 24), ("pre",), ("raw", "act")),
    ((8, 48, 256, 64, 256, 3, 24, 24), ("mask",), ("raw", "act")),
    # Cout = 128, W % 512 == 0: 512 x 128 tiles, conv_row_tall_kernel (64-B K stages; KDCC_CONV_LW=0: conv_row_pp128_kernel, tools/lw_check.py)
    ((1, 40, 1024, 192, 128, 3, 3, 3), ("pre", "mask"), ("raw", "act")),     # 18 / 12 periods per tile: both entry phases of the loop body
    ((3, 9, 1536, 64, 128, 3, 4, 4), ("pre",), ("raw", "act")),              # three tiles per image row (first / inner / last), image boundaries, 81 tiles
    ((1, 17, 512, 128, 128, 3, 16, 16), (), ("act",)),                       # H = dil + 1: no image row has all three kernel rows inside the image (8-period tiles only)
    ((1, 24, 512, 64, 128, 3, 1, 1), ("pre",), ("raw", "act")),
    ((1, 20, 1024, 128, 128, 3, 2, 2), (), ("act",)),
    ((2, 150, 512, 64, 128, 3, 1, 1), ("mask", "post"), ("raw",)),      # 300 tiles: two per workgroup on part of the chip
    ((1, 40, 512, 64, 128, 3, 16, 16), ("post",), ("raw", "act")),      # largest dilation of that kernel
]


@pytest.mark.parametrize("case,opnds,outs", PERSIST_CASES)
def test_conv_persistent_epilogues(K, case, opnds, outs):
    """v = acc (+ res_pre); v = mask > 0 ? v * mask_scale : 0 (if mask); v += res_post; raw = v; act = relu(v * s + b)."""
    dt = "bf16"
    N, H, W, Cin, Cout, k, p, d = case
    x = q(rnd(N, Cin, H, W), dt)
    w = q(rnd(Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5), dt)
    pre, post = q(rnd(N, Cout, H, W), dt), q(rnd(N, Cout, H, W), dt)
    mask = q(np.maximum(rnd(N, Cout, H, W), 0), dt)
    mscale, ascale, ashift = rnd(Cout) * 0.2 + 1.0, rnd(Cout) * 0.2 + 1.0, rnd(Cout) * 0.3
    bc = lambda v: v[None, :, None, None]
    ref = orc.conv2d_fwd(x, w, pad=p, dil=d)
    if "pre" in opnds:
        ref = ref + pre
    if "mask" in opnds:
        ref = np.where(mask > 0, ref * bc(mscale), 0.0)
    if "post" in opnds:
        ref = ref + post
    act_ref = np.maximum(ref * bc(ascale) + bc(ashift), 0)
    wp = K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt])
    cu = lambda v: torch.from_numpy(v).cuda()
    out_raw = torch.zeros((N, H, W, Cout + 32), dtype=DT[dt], device="cuda")[..., 16:16 + Cout] if "raw" in outs else None
    out_act = torch.zeros((N, H, W, 2 * Cout), dtype=DT[dt], device="cuda")[..., Cout:] if "act" in outs else None
    K.conv2d(dev_nhwc(x, dt), wp, 1, p, d,
             res_pre=dev_nhwc(pre, dt, ld=Cout + 16) if "pre" in opnds else None,
             mask=dev_nhwc(mask, dt) if "mask" in opnds else None, mask_scale=cu(mscale) if "mask" in opnds else None,
             res_post=dev_nhwc(post, dt, ld=Cout + 24) if "post" in opnds else None,
             out_raw=out_raw, out_act=out_act, act_scale=cu(ascale) if "act" in outs else None,
             act_shift=cu(ashift) if "act" in outs else None, act_relu="act" in outs)
    selected("conv_igemm_persist_kernel<pp>" if k == 1 else "conv_row_tall_kernel" if Cout == 128 else
             "conv_row_persist_kernel<pp>" if H < 2 * d else "conv_row_lw_kernel", f"{case} {opnds}")
    if out_raw is not None:
        assert_close(host_nchw(out_raw), ref, dt, f"raw {case} {opnds}")
    if out_act is not None:
        assert_close(host_nchw(out_act), act_ref, dt, f"act {case} {opnds}")


@pytest.mark.parametrize("case,opnds,outs", [
    # (N, H, W, Cin1, Cin2, Cout), epilogue operands, outputs -- the K-concatenated 1x1 conv of the bottleneck blocks
    ((1, 160, 512, 64, 128, 256), (), ("raw", "act")),              # forward form: conv3 | proj_conv -> block output + next activation
    ((1, 160, 512, 128, 64, 256), ("mask",), ("raw",)),             # backward form: conv1 | proj_conv input gradients through bn1's mask
    ((2, 20, 512, 192, 64, 2048), ("mask", "post"), ("raw",)),      # N tiles walked four at a time, two operands
    ((1, 80, 1024, 64, 64, 256), ("pre", "mask", "post"), ("raw", "act")),
])
def test_conv1x1_dual(K, case, opnds, outs):
    """kd_conv1x1_dual_fwd: [x | x2] . [w | w2]^T in one accumulator chain (wider_resnet.py:143-182: conv3 + proj_conv onto the block
    output; their transposes in the backward) against the oracle's two convs, and against the two-launch form it replaces."""
    dt = "bf16"
    N, H, W, C1, C2, Cout = case
    x1, x2 = q(rnd(N, C1, H, W), dt), q(np.maximum(rnd(N, C2, H, W), 0), dt)
    w1 = q(rnd(Cout, C1, 1, 1, scale=(1.0 / C1) ** 0.5), dt)
    w2 = q(rnd(Cout, C2, 1, 1, scale=(1.0 / C2) ** 0.5), dt)
    pre, post = q(rnd(N, Cout, H, W), dt), q(rnd(N, Cout, H, W), dt)
    mask = q(np.maximum(rnd(N, Cout, H, W), 0), dt)
    mscale, ascale, ashift = rnd(Cout) * 0.2 + 1.0, rnd(Cout) * 0.2 + 1.0, rnd(Cout) * 0.3
    bc = lambda v: v[None, :, None, None]
    ref = orc.conv2d_fwd(x1, w1) + orc.conv2d_fwd(x2, w2)
    if "pre" in opnds:
        ref = ref + pre
    if "mask" in opnds:
        ref = np.where(mask > 0, ref * bc(mscale), 0.0)
    if "post" in opnds:
        ref = ref + post
    act_ref = np.maximum(ref * bc(ascale) + bc(ashift), 0)
    cu = lambda v: torch.from_numpy(v).cuda()
    wcat = torch.cat([K.pack_conv_weight(cu(w1), DT[dt]), K.pack_conv_weight(cu(w2), DT[dt])], dim=3).contiguous()
    x1d, x2d = dev_nhwc(x1, dt), dev_nhwc(x2, dt, ld=C2 + 16)
    assert K.conv1x1_dual_ok(x1d, x2d, Cout, len(opnds))
    kw = dict(res_pre=dev_nhwc(pre, dt) if "pre" in opnds else None,
              mask=dev_nhwc(mask, dt) if "mask" in opnds else None, mask_scale=cu(mscale) if "mask" in opnds else None,
              res_post=dev_nhwc(post, dt) if "post" in opnds else None,
              act_scale=cu(ascale) if "act" in outs else None, act_shift=cu(ashift) if "act" in outs else None, act_relu="act" in outs)
    mk = lambda: (torch.zeros((N, H, W, Cout), dtype=DT[dt], device="cuda") if "raw" in outs else None,
                  torch.zeros((N, H, W, Cout), dtype=DT[dt], device="cuda") if "act" in outs else None)
    out_raw, out_act = mk()
    K.conv2d(x1d, wcat, x2=x2d, out_raw=out_raw, out_act=out_act, **kw)
    selected("conv_igemm_persist_kernel<pp,dual>", f"{case}")
    if out_raw is not None:
        assert_close(host_nchw(out_raw), ref, dt, f"dual raw {case} {opnds}")
    if out_act is not None:
        assert_close(host_nchw(out_act), act_ref, dt, f"dual act {case} {opnds}")
    if "pre" not in opnds:     # the form it replaces: conv(x2) stored in bf16, then conv(x1) with it as res_pre
        short = torch.empty((N, H, W, Cout), dtype=DT[dt], device="cuda")
        K.conv2d(x2d, K.pack_conv_weight(cu(w2), DT[dt]), out_raw=short)
        r2, a2 = mk()
        K.conv2d(x1d, K.pack_conv_weight(cu(w1), DT[dt]), out_raw=r2, out_act=a2, **{**kw, "res_pre": short})
        for a, b in ((out_raw, r2), (out_act, a2)):
            if a is not None:
                assert_close(host_nchw(a), host_nchw(b), dt, f"dual vs two launches {case}")
    # a shape the persistent kernel does not take must be refused loudly, not computed some other way
    small = dev_nhwc(q(rnd(1, C1, 8, 16), dt), dt)
    with pytest.raises(K.DualUnsupported):
        K.conv2d(small, wcat, x2=dev_nhwc(q(rnd(1, C2, 8, 16), dt), dt), out_raw=torch.zeros((1, 8, 16, Cout), dtype=DT[dt], device="cuda"))


@pytest.mark.parametrize("case,kernel", [
    ((1, 24, 512, 4096, 1280, 3, 12, 12), "conv_row_lw_kernel"),       # ASPP rate 12 at the network's real K = 9 x 4096, H = 2 dil
    ((1, 8, 1536, 2048, 1280, 3, 4, 4), "conv_row_lw_kernel"),         # mod7 dil 4, K = 9 x 2048: three tiles per image row
    ((1, 56, 1024, 4096, 256, 1, 0, 1), "conv_igemm_persist_kernel<pp>"),   # the 4096 -> 256 pointwise conv behind a depthwise branch
])
def test_conv_at_real_reduction_depth_per_tile(K, case, kernel):
    """The shipped 3x3 / 1x1 kernels at the K the networks run (PERSIST_CASES stop at Cin 192): the oracle on EVERY pixel for a
    subset of output channels that touches every 256-wide N tile and both 128-channel halves of it (a conv's output channel depends
    on its own filter only, so the oracle runs on the sliced weight: 1/40 of the work), judged per 256-pixel x channel-subset tile --
    the worst tile, not a global norm (wider_resnet.py:124-167, deeplabv3.py:53-62)."""
    dt = "bf16"
    N, H, W, Cin, Cout, k, p, d = case
    x = q(np.maximum(rnd(N, Cin, H, W), 0), dt)
    w = q(rnd(Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5), dt)
    ascale, ashift = rnd(Cout) * 0.2 + 1.0, rnd(Cout) * 0.3
    wp = K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt])
    out_raw = torch.zeros((N, H, W, Cout), dtype=DT[dt], device="cuda")
    out_act = torch.zeros((N, H, W, Cout), dtype=DT[dt], device="cuda")
    K.conv2d(dev_nhwc(x, dt), wp, 1, p, d, out_raw=out_raw, out_act=out_act, act_scale=torch.from_numpy(ascale).cuda(),
             act_shift=torch.from_numpy(ashift).cuda(), act_relu=True)
    selected(kernel, f"{case}")
    cs = np.arange(0, Cout, 40) if Cout > 256 else np.arange(0, Cout, 8)     # 32 channels: >= 6 per N tile, both halves of each
    ref = orc.conv2d_fwd(x, w[cs], pad=p, dil=d)
    act_ref = np.maximum(ref * ascale[cs][None, :, None, None] + ashift[cs][None, :, None, None], 0)
    for what, got, want in (("raw", host_nchw(out_raw)[:, cs], ref), ("act", host_nchw(out_act)[:, cs], act_ref)):
        assert_close(got, want, dt, f"{what} {case}")
        # per (256-pixel segment, N tile): relative L2 of the sampled channels
        tn = cs // 256
        worst = 0.0
        for t in np.unique(tn):
            e = ((got[:, tn == t] - want[:, tn == t]) ** 2).reshape(N, -1, H, W // 256, 256).sum(axis=(1, 4))
            r = (want[:, tn == t] ** 2).reshape(N, -1, H, W // 256, 256).sum(axis=(1, 4))
            worst = max(worst, float(np.sqrt((e / np.maximum(r, r.mean() / 16)).max())))
        assert worst < 1e-2, f"{what} {case}: worst tile relative L2 {worst:.2e} (bf16 output rounding is 2e-3)"


@pytest.mark.parametrize("case,kernel", [
    ((8, 128, 256, 128, 768, 3, 2, 2), "conv_row_lw_kernel"),              # 1024 M tiles x 3 N tiles = 3072 tiles, 8 images, dil 2
    ((8, 128, 256, 256, 768, 1, 0, 1), "conv_igemm_persist_kernel<pp>"),   # the same grid on the gathered 1x1 kernel
])
def test_conv_at_bench_batch_per_tile(K, case, kernel):
    """The persistent kernels at the BENCH's batch: 8 images and more than 2048 tiles in one launch, so the image stride of the
    input / output addressing and the tile walk past 2048 tiles are checked by the oracle, not only by the determinism / property
    tests of tests/test_fullsize_gpu.py.  Every pixel of all 8 images for 32 output channels spread over the three N tiles (the
    oracle runs on the sliced weight), judged per 256-pixel x N-tile tile (wider_resnet.py:124-167)."""
    dt = "bf16"
    N, H, W, Cin, Cout, k, p, d = case
    x = q(np.maximum(rnd(N, Cin, H, W), 0), dt)
    w = q(rnd(Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5), dt)
    ascale, ashift = rnd(Cout) * 0.2 + 1.0, rnd(Cout) * 0.3
    out_raw = torch.zeros((N, H, W, Cout), dtype=DT[dt], device="cuda")
    out_act = torch.zeros((N, H, W, Cout), dtype=DT[dt], device="cuda")
    K.conv2d(dev_nhwc(x, dt), K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt]), 1, p, d, out_raw=out_raw, out_act=out_act,
             act_scale=torch.from_numpy(ascale).cuda(), act_shift=torch.from_numpy(ashift).cuda(), act_relu=True)
    selected(kernel, f"{case}")
    assert (N * H * W // 256) * (Cout // 256) > 2048
    cs = np.arange(0, Cout, 24)
    ci = torch.from_numpy(cs).cuda()
    ref = orc.conv2d_fwd(x, w[cs], pad=p, dil=d)
    act_ref = np.maximum(ref * ascale[cs][None, :, None, None] + ashift[cs][None, :, None, None], 0)
    sl = lambda t: t.index_select(3, ci).float().cpu().numpy().transpose(0, 3, 1, 2)     # (slice on the device: the full tensors are 0.8 GB in fp32)
    for what, got, want in (("raw", sl(out_raw), ref), ("act", sl(out_act), act_ref)):
        assert_close(got, want, dt, f"{what} {case}")
        tn = cs // 256
        worst = 0.0
        for t in np.unique(tn):
            e = ((got[:, tn == t] - want[:, tn == t]) ** 2).reshape(N, -1, H, W // 256, 256).sum(axis=(1, 4))
            r = (want[:, tn == t] ** 2).reshape(N, -1, H, W // 256, 256).sum(axis=(1, 4))
            worst = max(worst, float(np.sqrt((e / np.maximum(r, r.mean() / 16)).max())))
        assert worst < 1e-2, f"{what} {case}: worst tile relative L2 {worst:.2e} (bf16 output rounding is 2e-3)"


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_wide_tile_dgrad_epilogue(K, dt):
    """Backward use of the wide tiles: dgrad (KD_PACK_DGRAD weights) of a 3x3 dil-2 conv on row-buffer tiles and of a 1x1 on
    gathered tiles, with the ReLU/BN mask and the shortcut gradient (res_post), plus res_pre (projection-path gradient)."""
    from kdcc_amd._lib import KD_PACK_DGRAD
    for (N, H, W, Cin, Cout, k, p, d) in [(1, 112, 512, 256, 64, 3, 2, 2), (1, 112, 512, 256, 128, 1, 0, 1)]:
        w = q(rnd(Cout, Cin, k, k, scale=0.05), dt)
        gy = q(rnd(N, Cout, H, W), dt)
        act = q(np.maximum(rnd(N, Cin, H, W), 0), dt)
        bn_scale = rnd(Cin) * 0.2 + 1.0
        short, proj = q(rnd(N, Cin, H, W), dt), q(rnd(N, Cin, H, W), dt)
        ref = orc.conv2d_dgrad(gy, w, (N, Cin, H, W), pad=p, dil=d) + proj
        ref = np.where(act > 0, ref * bn_scale[None, :, None, None], 0) + short
        wp = K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt], KD_PACK_DGRAD)
        out = torch.empty((N, H, W, Cin), dtype=DT[dt], device="cuda")
        K.conv2d(dev_nhwc(gy, dt), wp, 1, d * (k - 1) - p, d, res_pre=dev_nhwc(proj, dt), mask=dev_nhwc(act, dt),
                 mask_scale=torch.from_numpy(bn_scale).cuda(), res_post=dev_nhwc(short, dt), out_raw=out)
        assert_close(host_nchw(out), ref, dt, f"wide dgrad k={k}")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("k,p,d", [(3, 1, 1), (3, 4, 4), (1, 0, 1)])
def test_conv_dgrad_with_bn_relu_mask(K, dt, k, p, d):
    """dgrad = conv with KD_PACK_DGRAD weights; epilogue applies d(relu(bn(x)))/dx and adds the shortcut gradient."""
    from kdcc_amd._lib import KD_PACK_DGRAD
    N, H, W, Cin, Cout = 2, 8, 12, 64, 128
    w = q(rnd(Cout, Cin, k, k, scale=0.05), dt)
    gy = q(rnd(N, Cout, H, W), dt)
    act = q(np.maximum(rnd(N, Cin, H, W), 0), dt)      # saved activated input (mask)
    bn_scale = rnd(Cin) * 0.2 + 1.0
    short = q(rnd(N, Cin, H, W), dt)
    ref = orc.conv2d_dgrad(gy, w, (N, Cin, H, W), pad=p, dil=d)
    ref = np.where(act > 0, ref * bn_scale[None, :, None, None], 0) + short
    wp = K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt], KD_PACK_DGRAD)
    out = torch.empty((N, H, W, Cin), dtype=DT[dt], device="cuda")
    K.conv2d(dev_nhwc(gy, dt), wp, 1, d * (k - 1) - p, d, mask=dev_nhwc(act, dt),
             mask_scale=torch.from_numpy(bn_scale).cuda(), res_post=dev_nhwc(short, dt), out_raw=out)
    assert_close(host_nchw(out), ref, dt, "dgrad")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 8, 16, 64, 128), (1, 13, 9, 128, 64), (1, 40, 64, 256, 64),
                                   (1, 12, 64, 512, 256), (1, 3, 64, 256, 256)])    # (whole 256 x 256 tiles: conv_wgrad_pw_lw_kernel in bf16)
def test_pw_wgrad(K, dt, shape):
    N, H, W, Cin, Cout = shape
    a, gy = q(rnd(N, Cin, H, W), dt), q(rnd(N, Cout, H, W), dt)
    ref = orc.conv2d_wgrad(a, gy, (Cout, Cin, 1, 1))
    dw = torch.full((Cout, Cin, 1, 1), 7.0, device="cuda")
    K.pw_wgrad(dev_nhwc(a, dt), dev_nhwc(gy, dt), dw)
    if dt == "bf16" and Cin % 256 == 0 and Cout % 256 == 0:
        selected("conv_wgrad_pw_lw_kernel", f"pw_wgrad {shape}")
    assert_close(dw.cpu().numpy(), ref, dt, "pw_wgrad")
    K.pw_wgrad(dev_nhwc(a, dt), dev_nhwc(gy, dt), dw, accumulate=True)
    assert_close(dw.cpu().numpy(), 2 * ref, dt, "pw_wgrad accumulate")


DW_CASES = [(2, 24, 32, 16, 9, 20, 5), (1, 8, 8, 16, 3, 1, 1), (1, 23, 37, 72, 9, 20, 5), (2, 6, 50, 8, 9, 20, 5),
            # matrix-core path (bf16, C % 16 == 0): several lattice tiles with real halos, dil 1, ragged last tiles
            (1, 140, 270, 32, 9, 20, 5), (1, 40, 70, 16, 9, 4, 1), (2, 64, 128, 48, 9, 20, 5)]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", DW_CASES)
def test_dwconv_fwd_dgrad_wgrad(K, dt, case):
    N, H, W, Cc, k, p, d = case
    x, w = q(rnd(N, Cc, H, W), dt), rnd(Cc, 1, k, k, scale=1.0 / k)
    gy = q(rnd(N, Cc, H, W), dt)
    wd = torch.from_numpy(w).cuda()
    y = K.dwconv(dev_nhwc(x, dt), K.pack_dw_weight(wd), k, p, d)
    mfma = dt == "bf16" and k == 9 and Cc % 16 == 0    # the matrix-core kernels' domain (dwconv_mfma.hip)
    selected("dw_mfma_fwd_kernel<1,false>" if mfma else f"dwconv_fwd_kernel<{dt}>", f"dw fwd {case} {dt}")
    assert_close(host_nchw(y), orc.conv2d_fwd(x, w, pad=p, dil=d, groups=Cc), dt, "dw fwd")
    gx = K.dwconv(dev_nhwc(gy, dt), K.pack_dw_weight(wd, flip=True), k, p, d)
    assert_close(host_nchw(gx), orc.conv2d_dgrad(gy, w, x.shape, pad=p, dil=d, groups=Cc), dt, "dw dgrad")
    dw = torch.zeros((Cc, 1, k, k), device="cuda")
    K.dwconv_wgrad(dev_nhwc(x, dt), dev_nhwc(gy, dt), dw, k, p, d)
    selected("dw_mfma_wgrad_kernel" if mfma else f"dwconv_wgrad_kernel<{dt}>", f"dw wgrad {case} {dt}")
    assert_close(dw.cpu().numpy(), orc.conv2d_wgrad(x, gy, w.shape, pad=p, dil=d, groups=Cc), dt, "dw wgrad")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", [
    # N, H, W, C, k, pad, dil, inputs
    (2, 24, 32, 16, 9, 20, 5, 3),        # one lattice tile per residue class (the ASPP shape in small)
    (1, 140, 270, 32, 9, 20, 5, 3),      # several tiles with real halos, ragged last tiles
    (1, 40, 70, 16, 9, 4, 1, 2),         # two inputs, dil 1
    (2, 64, 128, 48, 9, 20, 5, 3),       # three channel groups
    (1, 23, 37, 72, 9, 20, 5, 3),        # C % 16 != 0: register kernels chained through res_post
    (1, 24, 32, 16, 9, 20, 5, 4),        # more inputs than one launch sums
    (1, 8, 8, 16, 3, 1, 1, 2),           # 3x3
])
def test_dwconv_sum(K, dt, case):
    """kd_dwconv_fwd_sum: the ASPP input gradient = sum of the branches' depthwise input gradients (deeplabv3.py:64-75),
    against the sum of the oracle's per-branch dgrads; bf16 sums in fp32 registers inside one launch (n <= 3)."""
    N, H, W, Cc, k, p, d, n = case
    gs = [q(rnd(N, Cc, H, W), dt) for _ in range(n)]
    ws = [rnd(Cc, 1, k, k, scale=1.0 / k) for _ in range(n)]
    ref = sum(orc.conv2d_dgrad(g, w, g.shape, pad=p, dil=d, groups=Cc) for g, w in zip(gs, ws))
    taps = [K.pack_dw_weight(torch.from_numpy(w).cuda(), flip=True) for w in ws]
    out = K.dwconv_sum([dev_nhwc(g, dt) for g in gs], taps, k, d * (k - 1) - p, d)
    if dt == "bf16" and k == 9 and Cc % 16 == 0 and n <= 3:
        selected(f"dw_mfma_fwd_kernel<{n},false>", f"dw sum {case}")
    assert_close(host_nchw(out), ref, dt, f"dw sum of {n}")
    if n <= 3:   # same numbers as the chained launches up to the rounding of the running bf16 sum
        run = None
        for g, t in zip(gs, taps):
            run = K.dwconv(dev_nhwc(g, dt), t, k, d * (k - 1) - p, d, res_post=run)
        assert_close(host_nchw(out), host_nchw(run), dt, "dw sum vs chain")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", [
    (2, 24, 32, 16, 9, 20, 5, 3),        # one half-height tile pair per residue class (the ASPP shape in small)
    (1, 140, 270, 32, 9, 20, 5, 3),      # several tiles in both directions with real halos, ragged last tiles
    (1, 40, 70, 16, 9, 4, 1, 2),         # two branches, dil 1: four row tiles, two column tiles
    (2, 64, 128, 48, 9, 20, 5, 3),       # three channel groups
    (1, 67, 33, 16, 9, 20, 5, 3),        # 14 lattice rows: a 13-row and a 1-row tile (fewer rows than fetch slots)
    (1, 23, 37, 72, 9, 20, 5, 3),        # C % 16 != 0: one register-kernel launch per branch
    (1, 24, 32, 16, 9, 20, 5, 4),        # four branches: three fused + one single
    (1, 8, 8, 16, 3, 1, 1, 2),           # 3x3
])
def test_dwconv_wgrad_multi(K, dt, case):
    """kd_dwconv_wgrad_multi: the depthwise weight gradients of the ASPP branches over their one input (deeplabv3.py:71-75),
    each against the oracle; bf16 9x9 branches share one staging of x and its operand windows (dw_mfma_wgrad_multi_kernel)."""
    N, H, W, Cc, k, p, d, n = case
    x = q(rnd(N, Cc, H, W), dt)
    gs = [q(rnd(N, Cc, H, W), dt) for _ in range(n)]
    dws = [torch.full((Cc, 1, k, k), 5.0, device="cuda") for _ in range(n)]
    xd, gd = dev_nhwc(x, dt), [dev_nhwc(g, dt) for g in gs]
    K.dwconv_wgrad_multi(xd, gd, dws, k, p, d)
    if dt == "bf16" and k == 9 and Cc % 16 == 0:
        selected("dw_mfma_wgrad_kernel" if n % 3 == 1 else f"dw_mfma_wgrad_multi_kernel<{3 if n % 3 == 0 else 2}>", f"dw wgrad of {n} {case}")
    refs = [orc.conv2d_wgrad(x, g, (Cc, 1, k, k), pad=p, dil=d, groups=Cc) for g in gs]
    for i, (dw, ref) in enumerate(zip(dws, refs)):
        assert_close(dw.cpu().numpy(), ref, dt, f"dw wgrad of {n}, branch {i}")
    K.dwconv_wgrad_multi(xd, gd, dws, k, p, d, accumulate=True)
    for i, (dw, ref) in enumerate(zip(dws, refs)):
        assert_close(dw.cpu().numpy(), 2 * ref, dt, f"dw wgrad of {n} accumulate, branch {i}")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", [
    (2, 24, 32, 16, 9, 20, 5, 3), (1, 140, 270, 32, 9, 20, 5, 3), (1, 40, 70, 16, 9, 4, 1, 2), (2, 64, 128, 48, 9, 20, 5, 3),
    (1, 23, 37, 72, 9, 20, 5, 3), (1, 24, 32, 16, 9, 20, 5, 5), (1, 8, 8, 16, 3, 1, 1, 2),
])
def test_dwconv_fanout(K, dt, case):
    """kd_dwconv_fwd_fanout: the ASPP branches' depthwise convs over one input (deeplabv3.py:71-75), each output against the
    oracle.  Three bf16 9x9 branches run on the lone-wave kernel (dwconv_lw.hip: resident operands, K slots, generated item loop);
    its fp32 accumulation order differs from the single launch's (28 K groups in 7 MFMAs instead of 9 tap rows), so against that it
    is held to one bf16 rounding on a few elements; every other combination is bit-identical to the single launch."""
    N, H, W, Cc, k, p, d, n = case
    x = q(rnd(N, Cc, H, W), dt)
    ws = [rnd(Cc, 1, k, k, scale=1.0 / k) for _ in range(n)]
    taps = [K.pack_dw_weight(torch.from_numpy(w).cuda()) for w in ws]
    xd = dev_nhwc(x, dt)
    outs = K.dwconv_fanout(xd, taps, k, p, d)
    mfma = dt == "bf16" and k == 9 and Cc % 16 == 0
    if mfma and n <= 3:
        selected("dw_lw_fan3_kernel" if n == 3 else f"dw_mfma_fwd_kernel<{n},true>", f"dw fan-out {case}")
    for i, (o, w) in enumerate(zip(outs, ws)):
        assert_close(host_nchw(o), orc.conv2d_fwd(x, w, pad=p, dil=d, groups=Cc), dt, f"dw fan-out {i}")
        single = K.dwconv(xd, taps[i], k, p, d)
        if mfma and i < 3 * (n // 3):     # (branches 0 .. 2 of every full group of three went through the lone-wave kernel)
            a, b = o.float(), single.float()
            diff = (a - b).abs()
            assert float(diff.max()) <= 2.0 ** -7 * float(b.abs().max()), f"fan-out output {i}: more than one bf16 rounding from the single launch"
            assert float((diff > 0).float().mean()) < 0.05, f"fan-out output {i}: too many elements differ from the single launch"
        else:
            assert torch.equal(o, single), f"fan-out output {i} differs from the single launch"


@pytest.mark.parametrize("case", [
    # N, H, W, C: shapes that walk the lone-wave fan-out kernel's item pipeline (dwconv_lw.hip), 9x9 / dilation 5 / 3 branches
    (1, 128, 256, 32),       # the ASPP map: 50 items per (image, channel group), every item with 4 column tiles
    (2, 65, 130, 16),        # H, W multiples of dil; 13-row tiles exactly
    (1, 131, 523, 16),       # three row tiles (one of a single row), three column tiles per class, ragged ones
    (3, 7, 9, 16),           # a residue class of 2 x 2 pixels: one short item per class, single column tile
    (1, 5, 5, 48),           # one pixel per class, three channel groups
    (1, 266, 40, 16),        # five row tiles, 8 columns
    (2, 10, 30, 16, 1),      # dilation 1, ONE work item per workgroup (the item loop's exit on its first pass)
    (1, 20, 30, 32, 1),      # dilation 1, two items per workgroup (nothing to stage behind the second)
    (1, 40, 120, 16, 2),     # dilation 2: four classes of 20 x 60, two row tiles and two column tiles each
])
def test_dwconv_fanout_lone_wave_shapes(K, case):
    """dw_lw_fan3_kernel against the oracle on the geometries its work-item descriptors distinguish: tiles in both directions,
    ragged last tiles, classes shorter than a tile, items with fewer than four column tiles, one-item workgroups."""
    N, H, W, Cc = case[:4]
    d = case[4] if len(case) > 4 else 5
    k, p, dt = 9, 4 * d, "bf16"
    x = q(rnd(N, Cc, H, W), dt)
    ws = [rnd(Cc, 1, k, k, scale=1.0 / k) for _ in range(3)]
    taps = [K.pack_dw_weight(torch.from_numpy(w).cuda()) for w in ws]
    outs = [torch.full((N, H, W, Cc), 7.0, dtype=torch.bfloat16, device="cuda") for _ in range(3)]     # stale data: every pixel must be written
    K.dwconv_fanout(dev_nhwc(x, dt), taps, k, p, d, outs=outs)
    selected("dw_lw_fan3_kernel", f"dw fan-out {case}")
    for i, (o, w) in enumerate(zip(outs, ws)):
        assert_close(host_nchw(o), orc.conv2d_fwd(x, w, pad=p, dil=d, groups=Cc), dt, f"dw fan-out {case} branch {i}")


LATTICE_CASES = [
    # N, H, W, C, k, pad, dil, branches -- bf16 only (the lattice-planar intermediates of the replaced ASPP branches)
    (2, 24, 32, 16, 9, 20, 5, 3),        # H, W not multiples of dil: classes one row / column shorter, padded cells
    (1, 128, 256, 32, 9, 20, 5, 3),      # the ASPP map itself: 26 x 52 lattice, one tile per class (last row / column padded)
    (1, 140, 270, 32, 9, 20, 5, 3),      # several tiles per class with real halos, ragged last tiles
    (2, 40, 70, 16, 9, 4, 1, 2),         # two branches, dil 1: one class, many tiles, nothing padded
    (2, 64, 128, 48, 9, 20, 5, 2),       # three channel groups (three planes), two branches
    (3, 65, 130, 16, 9, 20, 5, 3),       # H, W multiples of dil: no padded cells; tail rows of the plane only
]


def _lattice_of(K, a, dil):
    """NCHW numpy -> ops.Lattice through the row-move kernel (image order -> lattice order), plane by plane."""
    t = dev_nhwc(a, "bf16")
    N, H, W, Cc = t.shape
    rows = K.image_to_lattice(t, dil)
    return K.Lattice(N, H, W, Cc, dil, t=rows.reshape(rows.shape[0], Cc // 16, 16).permute(1, 0, 2).contiguous())


@pytest.mark.parametrize("case", LATTICE_CASES)
def test_lattice_rows_round_trip(K, case):
    """kd_lattice_rows_move: image order -> lattice order -> image order is the identity; rows of cells without a pixel are zero;
    the row of pixel (n, y, x) is the one include/kdcc.h states."""
    N, H, W, Cc, k, p, d, n = case
    x = q(rnd(N, Cc, H, W), "bf16")
    xd = dev_nhwc(x, "bf16")
    rows = K.image_to_lattice(xd, d)
    assert rows.shape[0] == K.lattice_rows(N, H, W, d) and rows.shape[0] % 256 == 0
    Ly, Lx = -(-H // d), -(-W // d)
    r = rows.cpu().float().numpy()
    expect = np.zeros_like(r)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for i in range(N):
        idx = ((i * d * d + (yy % d) * d + (xx % d)) * Ly + yy // d) * Lx + xx // d
        expect[idx.reshape(-1)] = x[i].reshape(Cc, -1).T
    assert np.array_equal(r, expect), "lattice row order / zero padding"
    back = torch.full_like(xd, 7.0)
    K.lattice_to_image(rows, back, d)
    assert torch.equal(back, xd)


@pytest.mark.parametrize("case", LATTICE_CASES)
def test_dwconv_fanout_lattice(K, case):
    """kd_dwconv_fwd_fanout_lattice: the same values as kd_dwconv_fwd_fanout, bit for bit, in the lattice-planar layout -- padded
    cells and tail rows zero -- and each output against the oracle."""
    N, H, W, Cc, k, p, d, n = case
    dt = "bf16"
    x = q(rnd(N, Cc, H, W), dt)
    ws = [rnd(Cc, 1, k, k, scale=1.0 / k) for _ in range(n)]
    taps = [K.pack_dw_weight(torch.from_numpy(w).cuda()) for w in ws]
    xd = dev_nhwc(x, dt)
    assert K.dwconv_lattice_ok(xd, n, k, p, d)
    outs = [K.Lattice(N, H, W, Cc, d) for _ in range(n)]
    for o in outs:
        o.t.fill_(3.0)                 # stale data everywhere: the kernel must zero the padded cells itself
        used = N * d * d * (-(-H // d)) * (-(-W // d))
        o.t[:, used:].zero_()
    K.dwconv_fanout_lattice(xd, taps, k, p, d, outs=outs)
    selected(f"dw_mfma_fwd_kernel<{n},true,lattice>", f"dw fan-out lattice {case}")
    plain = [K.dwconv(xd, t, k, p, d) for t in taps]     # (the 8-wave kernel's arithmetic; the NHWC fan-out of three runs on dw_lw_fan3_kernel)
    for i, (o, w) in enumerate(zip(outs, ws)):
        assert torch.equal(o.to_nhwc(), plain[i]), f"lattice fan-out output {i} differs from the NHWC launch"
        want = K.image_to_lattice(plain[i], d)
        got = o.t.permute(1, 0, 2).reshape(o.rows, Cc)
        assert torch.equal(got, want), f"lattice fan-out output {i}: padded cells / tail rows must be zero"
        assert_close(host_nchw(o.to_nhwc()), orc.conv2d_fwd(x, w, pad=p, dil=d, groups=Cc), dt, f"dw fan-out lattice {i}")


@pytest.mark.parametrize("case", LATTICE_CASES)
def test_dwconv_sum_lattice(K, case):
    """kd_dwconv_fwd_sum_lattice: bit-identical to kd_dwconv_fwd_sum on the same values, and against the oracle's summed dgrads."""
    N, H, W, Cc, k, p, d, n = case
    dt = "bf16"
    gs = [q(rnd(N, Cc, H, W), dt) for _ in range(n)]
    ws = [rnd(Cc, 1, k, k, scale=1.0 / k) for _ in range(n)]
    taps = [K.pack_dw_weight(torch.from_numpy(w).cuda(), flip=True) for w in ws]
    lat = [_lattice_of(K, g, d) for g in gs]
    out = K.dwconv_sum_lattice(lat, taps, k, d * (k - 1) - p, d)
    selected(f"dw_mfma_fwd_kernel<{n},false,lattice>", f"dw sum lattice {case}")
    plain = K.dwconv_sum([dev_nhwc(g, dt) for g in gs], taps, k, d * (k - 1) - p, d)
    assert torch.equal(out, plain), "lattice sum differs from the NHWC launch"
    ref = sum(orc.conv2d_dgrad(g, w, g.shape, pad=p, dil=d, groups=Cc) for g, w in zip(gs, ws))
    assert_close(host_nchw(out), ref, dt, f"dw sum lattice of {n}")


@pytest.mark.parametrize("case", LATTICE_CASES)
def test_dwconv_wgrad_multi_lattice(K, case):
    """kd_dwconv_wgrad_multi_lattice: bit-identical to kd_dwconv_wgrad_multi, each branch against the oracle, accumulate too."""
    N, H, W, Cc, k, p, d, n = case
    dt = "bf16"
    x = q(rnd(N, Cc, H, W), dt)
    gs = [q(rnd(N, Cc, H, W), dt) for _ in range(n)]
    xd = dev_nhwc(x, dt)
    lat = [_lattice_of(K, g, d) for g in gs]
    dws = [torch.full((Cc, 1, k, k), 5.0, device="cuda") for _ in range(n)]
    K.dwconv_wgrad_multi_lattice(xd, lat, dws, k, p, d)
    selected(f"dw_mfma_wgrad_multi_kernel<{n},lattice>", f"dw wgrad lattice {case}")
    plain = [torch.empty((Cc, 1, k, k), device="cuda") for _ in range(n)]
    K.dwconv_wgrad_multi(xd, [dev_nhwc(g, dt) for g in gs], plain, k, p, d)
    refs = [orc.conv2d_wgrad(x, g, (Cc, 1, k, k), pad=p, dil=d, groups=Cc) for g in gs]
    for i in range(n):
        assert torch.equal(dws[i], plain[i]), f"lattice weight gradient {i} differs from the NHWC launch"
        assert_close(dws[i].cpu().numpy(), refs[i], dt, f"dw wgrad lattice, branch {i}")
    K.dwconv_wgrad_multi_lattice(xd, lat, dws, k, p, d, accumulate=True)
    for i in range(n):
        assert_close(dws[i].cpu().numpy(), 2 * refs[i], dt, f"dw wgrad lattice accumulate, branch {i}")


@pytest.mark.parametrize("case", [(2, 24, 32, 16, 5), (1, 131, 261, 32, 5), (1, 30, 60, 16, 2)])
def test_dwconv_epilogue_bf16(K, case):
    """bias + res_pre + BN/ReLU mask + res_post through the depthwise epilogue, multi-tile shapes included."""
    N, H, W, Cc, d = case
    k, p, dt = 9, 4 * d, "bf16"
    x, w = q(rnd(N, Cc, H, W), dt), rnd(Cc, 1, k, k, scale=1.0 / k)
    bias, scale = rnd(Cc), np.abs(rnd(Cc)) + 0.5
    pre, post = q(rnd(N, Cc, H, W), dt), q(rnd(N, Cc, H, W), dt)
    act = q(np.maximum(rnd(N, Cc, H, W), 0), dt)
    y = K.dwconv(dev_nhwc(x, dt, ld=Cc + 16), K.pack_dw_weight(torch.from_numpy(w).cuda()), k, p, d,
                 bias=torch.from_numpy(bias).cuda(), res_pre=dev_nhwc(pre, dt), mask=dev_nhwc(act, dt),
                 mask_scale=torch.from_numpy(scale).cuda(), res_post=dev_nhwc(post, dt))
    ref = orc.conv2d_fwd(x, w, pad=p, dil=d, groups=Cc) + bias[None, :, None, None] + pre
    ref = np.where(act > 0, ref * scale[None, :, None, None], 0.0) + post
    assert_close(host_nchw(y), ref, dt, "dw epilogue")


def test_dwsep_block_golden(K, golden):
    """DepthwiseSeparableBlock fwd + all grads against the reference's own outputs (fp32 path)."""
    g = golden("dwsep")
    for tag in ("k9d5", "k3d1"):
        Cc, Co, k, p, d, H, W = [int(v) for v in g[f"{tag}.cfg"]]
        # pad channels to the GEMM granule (32 for f32) with zeros: result is unchanged
        Cp = 32
        x = np.zeros((2, Cp, H, W), np.float32); x[:, :Cc] = g[f"{tag}.x"]
        wdw = np.zeros((Cp, 1, k, k), np.float32); wdw[:Cc] = g[f"{tag}.w_dw"]
        wpw = np.zeros((Co, Cp, 1, 1), np.float32); wpw[:, :Cc] = g[f"{tag}.w_pw"]
        xd = dev_nhwc(x, "f32")
        mid = K.dwconv(xd, K.pack_dw_weight(torch.from_numpy(wdw).cuda()), k, p, d)
        out = torch.empty((2, H, W, Co), device="cuda")
        K.conv2d(mid, K.pack_conv_weight(torch.from_numpy(wpw).cuda(), torch.float32), out_raw=out)
        assert_close(host_nchw(out), g[f"{tag}.y"], "f32", f"{tag} y")
        gy = dev_nhwc(g[f"{tag}.gy"], "f32")
        gw_pw = torch.empty((Co, Cp, 1, 1), device="cuda")
        K.pw_wgrad(mid, gy, gw_pw)
        assert_close(gw_pw.cpu().numpy()[:, :Cc], g[f"{tag}.gw_pw"], "f32", f"{tag} gw_pw")
        # dgrad of the 1x1: Cout (=K dim) must be a multiple of 32 -> pad gy / weight rows with zeros
        Kp = ((Co + 31) // 32) * 32
        gyp = np.zeros((2, Kp, H, W), np.float32); gyp[:, :Co] = g[f"{tag}.gy"]
        wpp = np.zeros((Kp, Cp, 1, 1), np.float32); wpp[:Co] = wpw
        from kdcc_amd._lib import KD_PACK_DGRAD
        gmid = torch.empty((2, H, W, Cp), device="cuda")
        K.conv2d(dev_nhwc(gyp, "f32"), K.pack_conv_weight(torch.from_numpy(wpp).cuda(), torch.float32, KD_PACK_DGRAD), out_raw=gmid)
        gw_dw = torch.empty((Cp, 1, k, k), device="cuda")
        K.dwconv_wgrad(xd, gmid, gw_dw, k, p, d)
        assert_close(gw_dw.cpu().numpy()[:Cc], g[f"{tag}.gw_dw"], "f32", f"{tag} gw_dw")
        gx = K.dwconv(gmid, K.pack_dw_weight(torch.from_numpy(wdw).cuda(), flip=True), k, p, d)
        assert_close(host_nchw(gx)[:, :Cc], g[f"{tag}.gx"], "f32", f"{tag} gx")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_scale_by_device_scalar(K, dt):
    """Upstream-gradient factor of the fused loss Functions: x * 1 is skipped on the device, any other factor is applied in place."""
    for n in (8 * 1000, 8 * 37 + 5, 3):
        x = torch.from_numpy(rnd(n)).to(DT[dt]).cuda()
        keep = x.clone()
        K.scale_by_device_scalar_(x, torch.ones((), device="cuda"))
        assert torch.equal(x, keep)
        K.scale_by_device_scalar_(x, torch.tensor(0.37, device="cuda"))
        ref = (keep.float() * torch.tensor(0.37, device="cuda")).to(DT[dt])
        assert torch.equal(x, ref)
    x4 = torch.from_numpy(rnd(2, 5, 4, 6)).cuda().contiguous(memory_format=torch.channels_last)
    K.scale_by_device_scalar_(x4, torch.tensor([2.0], device="cuda"))
    assert x4.is_contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("shape", [(2, 20, 28), (1, 64, 128), (1, 37, 45), (1, 33, 100), (2, 2, 2), (1, 70, 29)])
def test_stem_conv_pool_fused(K, shape):
    """kd_stem_conv_pool (mod1 -> pool2 -> bn1/relu of mod2.block1, wider_resnet.py:307-309, 353-356) against the oracle's
    conv + max-pool, and bit for bit against the two-kernel path (round(max) == max(round)); odd sizes: ragged last column
    group, last pooled row with one stem row, several row segments."""
    N, H, W = shape
    x = rnd(N, 3, H, W)
    w = rnd(64, 3, 3, 3, scale=0.2)
    scale, shift = rnd(64) * 0.2 + 1.0, rnd(64) * 0.2
    xd, wd, sd, hd = (torch.from_numpy(a).cuda() for a in (x, w, scale, shift))
    raw, act = K.stem_conv_pool(xd, wd, sd, hd)
    selected("stem_pool_kernel", "stem + pool2")
    y = K.stem_conv(xd, wd, torch.bfloat16)
    raw2, act2 = K.maxpool3x3s2(y, sd, hd)
    assert torch.equal(raw, raw2) and torch.equal(act, act2)
    _, act3 = K.stem_conv_pool(xd, wd, sd, hd, want_raw=False)
    assert torch.equal(act3, act2)
    pref = orc.maxpool3x3s2(orc.conv2d_fwd(q(x, "bf16"), q(w, "bf16"), pad=1))
    assert_close(host_nchw(raw), pref, "bf16", "stem+pool raw")
    assert_close(host_nchw(act), np.maximum(pref * scale[None, :, None, None] + shift[None, :, None, None], 0), "bf16", "stem+pool act")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_stem_pool_upsample_imagepool(K, dt):
    x = rnd(2, 3, 20, 28)
    w = rnd(64, 3, 3, 3, scale=0.2)
    y = K.stem_conv(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), DT[dt])
    ref = orc.conv2d_fwd(x, w, pad=1)
    assert_close(host_nchw(y), ref, dt, "stem")
    # max-pool (+ BN/ReLU second output) on the stem output as stored
    yq = host_nchw(y)
    scale, shift = rnd(64) * 0.2 + 1.0, rnd(64) * 0.2
    raw, act = K.maxpool3x3s2(y, torch.from_numpy(scale).cuda(), torch.from_numpy(shift).cuda())
    pref = orc.maxpool3x3s2(yq)
    assert_close(host_nchw(raw), pref, dt, "pool raw")
    assert_close(host_nchw(act), np.maximum(pref * scale[None, :, None, None] + shift[None, :, None, None], 0), dt, "pool act")
    # upsample x4 into a channel slice, and an odd-channel fp32-out case (the logits path)
    up = torch.zeros((2, 40, 56, 80), dtype=DT[dt], device="cuda")
    K.upsample_bilinear_ac(raw, (40, 56), out=up[..., 16:80])
    assert_close(host_nchw(up[..., 16:80]), orc.upsample_bilinear_ac(host_nchw(raw), (40, 56)), dt, "upsample")
    lg = rnd(2, 19, 7, 9)
    lgd = torch.from_numpy(np.ascontiguousarray(lg.transpose(0, 2, 3, 1))).cuda()  # fp32 decoder output
    o = K.upsample_bilinear_ac(lgd, (14, 18), out_dtype=torch.float32)
    assert_close(host_nchw(o), orc.upsample_bilinear_ac(lg, (14, 18)), "f32", "logit upsample")
    lg2 = rnd(1, 19, 21, 333)                      # several 256-pixel output segments per row, ragged last one
    lgd2 = torch.from_numpy(np.ascontiguousarray(lg2.transpose(0, 2, 3, 1))).cuda()
    o2 = K.upsample_bilinear_ac(lgd2, (42, 666), out_dtype=torch.float32)
    assert_close(host_nchw(o2), orc.upsample_bilinear_ac(lg2, (42, 666)), "f32", "logit upsample (segments)")
    o2b = K.upsample_bilinear_ac(lgd2, (42, 668), out_dtype=torch.float32)   # 668 * 19 % 4 == 0: the 16-B-store kernel, ragged last block
    assert_close(host_nchw(o2b), orc.upsample_bilinear_ac(lg2, (42, 668)), "f32", "logit upsample (flat rows)")
    # odd channel count (the 19-class logits), both corner modes
    import torch.nn.functional as F
    lg3 = rnd(2, 19, 9, 10)
    lgd3 = torch.from_numpy(np.ascontiguousarray(lg3.transpose(0, 2, 3, 1))).cuda()
    for align in (True, False):
        o3 = K.upsample_bilinear_ac(lgd3, (18, 20), out_dtype=torch.float32, align_corners=align)
        ref3 = F.interpolate(torch.from_numpy(lg3), size=(18, 20), mode="bilinear", align_corners=align).numpy()
        assert_close(host_nchw(o3), ref3, "f32", f"logit upsample align={align}")
    o4 = K.upsample_bilinear_ac(raw, (40, 56), align_corners=False)     # vector path, GSCNN's non-aligned mode
    ref4 = F.interpolate(torch.from_numpy(host_nchw(raw)), size=(40, 56), mode="bilinear", align_corners=False).numpy()
    assert_close(host_nchw(o4), ref4, dt, "upsample align_corners=False")
    # ASPP image pooling branch
    Cin, Cout = 64, 16
    xi = q(rnd(2, Cin, 6, 10), dt)
    wi, sc, sh = rnd(Cout, Cin, scale=0.2), rnd(Cout) * 0.2 + 1, rnd(Cout) * 0.2
    cat = torch.zeros((2, 6, 10, 48), dtype=DT[dt], device="cuda")
    K.aspp_image_pool(dev_nhwc(xi, dt), torch.from_numpy(wi).cuda(), torch.from_numpy(sc).cuda(), torch.from_numpy(sh).cuda(),
                      cat[..., :Cout])
    v = np.maximum((orc.gap(xi) @ wi.T) * sc + sh, 0)
    assert_close(host_nchw(cat[..., :Cout]), np.broadcast_to(v[:, :, None, None], (2, Cout, 6, 10)), dt, "image pool")


def test_losses_vs_golden_and_oracle(K, golden):
    g = golden("losses")
    for tag, T in (("kld_T1", 1.0), ("kld_T5", 5.0), ("kld2d_T5", 5.0)):
        s, t = torch.from_numpy(g[f"{tag}.s"]).cuda(), torch.from_numpy(g[f"{tag}.t"]).cuda()
        loss, grad = K.kldiv(s, t, T)
        np.testing.assert_allclose(loss.item(), g[f"{tag}.loss"], rtol=1e-4)
        np.testing.assert_allclose(grad.cpu().numpy(), g[f"{tag}.grad"], rtol=1e-3, atol=1e-7)
    # same values through a channels_last (NHWC memory) view: the engine's layout
    s = torch.from_numpy(g["kld_T1.s"]).cuda().contiguous(memory_format=torch.channels_last)
    t = torch.from_numpy(g["kld_T1.t"]).cuda().contiguous(memory_format=torch.channels_last)
    loss, grad = K.kldiv(s, t, 1.0)
    np.testing.assert_allclose(loss.item(), g["kld_T1.loss"], rtol=1e-4)
    np.testing.assert_allclose(grad.cpu().numpy(), g["kld_T1.grad"], rtol=1e-3, atol=1e-7)
    for tag, nc in (("mse_1000", 1000), ("mse_1", 1)):
        for fmt in (torch.contiguous_format, torch.channels_last):
            s = torch.from_numpy(g[f"{tag}.s"]).cuda().contiguous(memory_format=fmt)
            t = torch.from_numpy(g[f"{tag}.t"]).cuda().contiguous(memory_format=fmt)
            loss, grad = K.hint_mse(s, t, nc)
            np.testing.assert_allclose(loss.item(), g[f"{tag}.loss"], rtol=1e-5)
            np.testing.assert_allclose(grad.cpu().numpy(), g[f"{tag}.grad"], rtol=1e-4, atol=1e-9)
    # mixed layouts -> strided kernel
    s = torch.from_numpy(g["mse_1000.s"]).cuda()
    t = torch.from_numpy(g["mse_1000.t"]).cuda().contiguous(memory_format=torch.channels_last)
    loss, grad = K.hint_mse(s, t, 1000)
    np.testing.assert_allclose(loss.item(), g["mse_1000.loss"], rtol=1e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), g["mse_1000.grad"], rtol=1e-4, atol=1e-9)
    for tag in ("whmse_c", "whmse_nc"):
        for fmt in (torch.contiguous_format, torch.channels_last):
            s = torch.from_numpy(g[f"{tag}.s"]).cuda().contiguous(memory_format=fmt)
            t = torch.from_numpy(g[f"{tag}.t"]).cuda().contiguous(memory_format=fmt)
            loss, grad = K.weighted_hint_mse(s, t, torch.from_numpy(g[f"{tag}.w"]).cuda())
            np.testing.assert_allclose(loss.item(), g[f"{tag}.loss"], rtol=1e-5)
            np.testing.assert_allclose(grad.cpu().numpy(), g[f"{tag}.grad"], rtol=1e-4, atol=1e-9)
    ce = K.ce2d(torch.from_numpy(g["ce.x"]).cuda(), torch.from_numpy(g["ce.target"]).cuda())
    np.testing.assert_allclose(ce.item(), g["ce.loss"], rtol=1e-5)


def test_mse_sum_reduction_vs_reference_and_oracle(golden):
    """MSELoss(reduction='sum') (reference losses/MSELoss.py:9-16 passes `reduction` to nn.MSELoss): value and gradient against the
    reference's own (tests/golden/variants.npz) and, on other data, against the oracle."""
    from kdcc_amd import losses
    g = golden("variants")
    s = torch.from_numpy(g["mse_sum.s"]).cuda().requires_grad_(True)
    loss = losses.MSELoss(reduction='sum', num_classes=19)(s, torch.from_numpy(g["mse_sum.t"]).cuda())
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["mse_sum.loss"], rtol=2e-6)
    np.testing.assert_allclose(s.grad.cpu().numpy(), g["mse_sum.grad"], rtol=1e-5, atol=1e-6)
    a, b = rnd(3, 40, 7, 33), rnd(3, 40, 7, 33)
    s = torch.from_numpy(a).cuda().requires_grad_(True)
    loss = losses.MSELoss(reduction='sum', num_classes=1000)(s, torch.from_numpy(b).cuda())
    loss.backward()
    rl, rg = orc.mse_sum(a, b, 1000)
    np.testing.assert_allclose(loss.item(), rl, rtol=2e-6)
    np.testing.assert_allclose(s.grad.cpu().numpy(), rg, rtol=1e-5, atol=1e-4)


def test_biased_cheap_conv_block_vs_reference_and_oracle(golden):
    """DepthwiseSeparableBlock(bias=True) (reference depthwise_separable_conv.py:7-9 passes `bias` to both convs): forward and all six
    gradients in fp32 against the reference's own block (tests/golden/variants.npz, 1e-3), and on the bf16 MFMA path against the oracle
    restatement on the bf16-rounded parameters."""
    from kdcc_amd.models.students import DepthwiseSeparableBlock
    from test_oracle import dwsep_bias_ref
    g = golden("variants")
    Cc, Co, k, p, d, H, W = [int(v) for v in g["dwsep_bias.cfg"]]

    def run(dtype, src):
        blk = DepthwiseSeparableBlock(Cc, Co, k, p, d, Cc, True).cuda()
        with torch.no_grad():
            blk.separable_conv.weight.copy_(torch.from_numpy(src["dwsep_bias.w_dw"])); blk.separable_conv.bias.copy_(torch.from_numpy(src["dwsep_bias.b_dw"]))
            blk.pointwise_conv.weight.copy_(torch.from_numpy(src["dwsep_bias.w_pw"])); blk.pointwise_conv.bias.copy_(torch.from_numpy(src["dwsep_bias.b_pw"]))
        blk = blk.to(dtype)
        x = torch.from_numpy(src["dwsep_bias.x"]).cuda().to(dtype).requires_grad_(True)
        y = blk(x)
        y.backward(torch.from_numpy(src["dwsep_bias.gy"]).cuda().to(dtype))
        return {"y": y, "gx": x.grad, "gw_dw": blk.separable_conv.weight.grad, "gb_dw": blk.separable_conv.bias.grad,
                "gw_pw": blk.pointwise_conv.weight.grad, "gb_pw": blk.pointwise_conv.bias.grad}

    got = run(torch.float32, g)
    for n, v in got.items():
        ref = g[f"dwsep_bias.{n}"]
        err = np.abs(v.detach().float().cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-6)
        assert err < 1e-3, (n, err)
    # bf16: the oracle on the rounded operands (what the kernels see)
    gq = {kk: (q(np.asarray(g[kk]), "bf16") if kk.split(".")[1] in ("x", "w_dw", "b_dw", "w_pw", "b_pw", "gy") else g[kk]) for kk in g.files}
    ref = dwsep_bias_ref(gq)
    got = run(torch.bfloat16, gq)
    for n, v in got.items():
        a, b = v.detach().float().cpu().numpy().astype(np.float64), np.asarray(ref[n], np.float64)
        err = np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30))
        assert err < 2e-2, (n, err)


@pytest.mark.parametrize("tag,avg,weighted", [("ce_w_mean", True, True), ("ce_w_sum", False, True), ("ce_sum", False, False)])
@pytest.mark.parametrize("fmt", ["nchw", "nhwc", "bf16_nhwc"])
def test_cross_entropy_class_weights_and_sum_vs_reference_and_oracle(golden, tag, avg, weighted, fmt):
    """CrossEntropyLoss2d(weight, size_average) (reference losses/CrossEntropy.py:5-14 = nn.NLLLoss(weight, size_average, ignore_index)
    on log_softmax): value and gradient against the reference's own (tests/golden/variants.npz) through the strided and the NHWC
    kernels; the bf16 operand against the oracle on the rounded logits."""
    from kdcc_amd import losses
    g = golden("variants")
    xh = g["ce.x"]
    x = torch.from_numpy(xh).cuda()
    if fmt != "nchw":
        x = x.contiguous(memory_format=torch.channels_last)
    if fmt == "bf16_nhwc":
        x = x.bfloat16()
        xh = q(xh, "bf16")
    x.requires_grad_(True)
    t = torch.from_numpy(g["ce.target"]).cuda()
    w = torch.from_numpy(g["ce.w"]).cuda() if weighted else None
    loss = losses.CrossEntropyLoss2d(weight=w, size_average=avg, ignore_index=255)(x, t)
    loss.backward()
    rl, rg = orc.ce2d_weighted(xh, g["ce.target"], g["ce.w"] if weighted else None, avg)
    np.testing.assert_allclose(loss.item(), rl, rtol=2e-5)
    tol = dict(rtol=2e-2, atol=2e-3 * np.abs(rg).max()) if fmt == "bf16_nhwc" else dict(rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), rg, **tol)
    if fmt != "bf16_nhwc":
        np.testing.assert_allclose(loss.item(), g[f"{tag}.loss"], rtol=2e-5)
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"{tag}.grad"], rtol=1e-4, atol=1e-6)


def test_losses_bf16_large_vs_oracle(K):
    """bf16 NHWC operands at a hint-like size: vector path, compared with the oracle on the rounded inputs."""
    N, Cc, H, W = 2, 64, 24, 40
    s, t = q(rnd(N, Cc, H, W), "bf16"), q(rnd(N, Cc, H, W), "bf16")
    sd = dev_nhwc(s, "bf16").permute(0, 3, 1, 2)
    td = dev_nhwc(t, "bf16").permute(0, 3, 1, 2)
    loss, grad = K.hint_mse(sd, td, 1000)
    rl, rg = orc.mse(s, t, 1000)
    np.testing.assert_allclose(loss.item(), rl, rtol=1e-5)
    assert_close(grad.float().cpu().numpy(), rg, "bf16", "mse grad")
    loss, grad = K.kldiv(sd[:, :19], td[:, :19], 1.0)
    rl, rg = orc.kldiv(s[:, :19], t[:, :19], 1.0)
    np.testing.assert_allclose(loss.item(), rl, rtol=1e-4)
    assert_close(grad.float().cpu().numpy(), rg, "bf16", "kld grad")


@pytest.mark.parametrize("Cc", [32, 33, 64, 100])
def test_logit_losses_channels_last_any_class_count(K, Cc):
    """KLDiv / CE / confusion on channels-last fp32 logits with class counts on both sides of what the NHWC fast paths stage in
    the default 64 KiB of dynamic LDS (KLDiv: 2 x 256 x C floats, C <= 32; CE: C <= 64; confusion: C <= 51): larger counts -- the
    100-class CIFAR heads -- must take the strided kernels, not fail at launch."""
    N, H, W = 2, 9, 13
    s, t = rnd(N, Cc, H, W), rnd(N, Cc, H, W)
    tgt = RNG.integers(0, Cc, size=(N, H, W)).astype(np.int64)
    tgt[:, :2] = 255
    sd, td = dev_nhwc(s, "f32").permute(0, 3, 1, 2), dev_nhwc(t, "f32").permute(0, 3, 1, 2)
    loss, grad = K.kldiv(sd, td, 2.0)
    rl, rg = orc.kldiv(s, t, 2.0)
    np.testing.assert_allclose(loss.item(), rl, rtol=1e-4)
    assert_close(grad.cpu().numpy(), rg, "f32", f"kld grad C={Cc}")
    np.testing.assert_allclose(K.ce2d(sd, torch.from_numpy(tgt).cuda()).item(), orc.ce2d(s, tgt), rtol=1e-4)
    if Cc <= 64:
        conf = K.confusion(sd, torch.from_numpy(tgt).cuda()).cpu().numpy()
        assert np.array_equal(conf, orc.confusion(s, tgt)), f"confusion C={Cc}"


def test_radam_golden(K, golden):
    g = golden("radam")
    p = torch.from_numpy(g["p"][0].copy()).cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i in range(8):
        K.radam_step(p, torch.from_numpy(g["g"][i]).cuda(), m, v, i + 1, float(g["lr"]), 0.9, 0.999, 1e-8, 0.0)
        np.testing.assert_allclose(p.cpu().numpy(), g["p"][i + 1], rtol=2e-6, atol=1e-7)


def test_radam_multi_tensor_matches_single(K, golden):
    """kd_radam_step_multi: the reference's 8-step golden through the batched launch, next to 60 other tensors of
    odd sizes and different step counts / hyper-parameters, each bit for bit equal to its own kd_radam_step."""
    g = golden("radam")
    gen = torch.Generator(device="cuda").manual_seed(5)
    sizes = [1, 7, 2048, 2049, 5000, 300001] * 10
    others = [dict(p=torch.randn(n, device="cuda", generator=gen), m=torch.zeros(n, device="cuda"), v=torch.zeros(n, device="cuda"),
                   lr=1e-3 * (1 + i % 3), wd=0.0 if i % 2 else 1e-2, step0=i % 7) for i, n in enumerate(sizes)]
    single = [dict(p=o["p"].clone(), m=o["m"].clone(), v=o["v"].clone()) for o in others]
    p = torch.from_numpy(g["p"][0].copy()).cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i in range(8):
        grads = [torch.randn(o["p"].numel(), device="cuda", generator=gen) for o in others]
        items = [(p, torch.from_numpy(g["g"][i]).cuda(), m, v, i + 1, float(g["lr"]), 0.9, 0.999, 1e-8, 0.0)]
        items += [(o["p"], gr, o["m"], o["v"], o["step0"] + i + 1, o["lr"], 0.9, 0.99, 1e-8, o["wd"]) for o, gr in zip(others, grads)]
        K.radam_step_multi(items)
        np.testing.assert_allclose(p.cpu().numpy(), g["p"][i + 1], rtol=2e-6, atol=1e-7)
        for o, sgl, gr in zip(others, single, grads):
            K.radam_step(sgl["p"], gr, sgl["m"], sgl["v"], o["step0"] + i + 1, o["lr"], 0.9, 0.99, 1e-8, o["wd"])
            assert torch.equal(o["p"], sgl["p"]) and torch.equal(o["m"], sgl["m"]) and torch.equal(o["v"], sgl["v"])


def test_errors_are_loud(K):
    from kdcc_amd._lib import KdccError
    x = torch.zeros((1, 4, 4, 48), device="cuda")  # Cin not a multiple of 32
    w = torch.zeros((32, 1, 1, 48), device="cuda")
    with pytest.raises(KdccError):
        K.conv2d(x, w, out_raw=torch.empty((1, 4, 4, 32), device="cuda"))
    with pytest.raises(KdccError):
        K.conv2d(torch.zeros((1, 4, 4, 64)), torch.zeros((32, 1, 1, 64)), out_raw=torch.empty((1, 4, 4, 32)))  # CPU tensors


def test_confusion_exact(K, golden):
    """kd_confusion == the reference's accumulated confusion matrix, bit for bit (int64), for NCHW fp32 logits, the
    engine's NHWC layout, bf16 logits (vs the integer oracle on the rounded values) and a ragged large case."""
    g = golden("confusion")
    conf = None
    for x, t in zip(g["x"], g["target"]):
        xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t.astype(np.int64)).cuda()
        conf = K.confusion(xd, td, conf, accumulate=conf is not None)
    assert conf.dtype == torch.int64 and np.array_equal(conf.cpu().numpy(), g["conf"])
    conf2 = torch.zeros((19, 19), dtype=torch.int64, device="cuda")
    for x, t in zip(g["x"], g["target"]):
        xd = torch.from_numpy(x).cuda().contiguous(memory_format=torch.channels_last)
        K.confusion(xd, torch.from_numpy(t.astype(np.int64)).cuda(), conf2, accumulate=True)
    assert np.array_equal(conf2.cpu().numpy(), g["conf"])
    # bf16 NHWC logits, pixel count not a multiple of the block, labels outside [0, C) other than 255
    x = q(rnd(3, 19, 37, 53), "bf16")
    t = RNG.integers(-1, 21, (3, 37, 53)).astype(np.int64)
    t[:, :2] = 255
    got = K.confusion(dev_nhwc(x, "bf16").permute(0, 3, 1, 2), torch.from_numpy(t).cuda())
    assert np.array_equal(got.cpu().numpy(), orc.confusion(x, t))


# ------------------------------------------------------------------------------------------------- backward plumbing (mode B)
WGRAD_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 12, 20, 64, 128, 3, 1, 1, 1),
    (1, 17, 23, 72, 40, 3, 1, 2, 2),      # ragged channels (Cin, Cout % 8 == 0 only) and pixels, dilation 2
    (1, 16, 24, 64, 64, 3, 2, 1, 1),      # stride 2 (mod4.block1.convs.conv1)
    (2, 10, 14, 128, 256, 1, 2, 0, 1),    # strided projection
    (1, 9, 31, 304, 256, 3, 1, 1, 1),     # decoder: 304 input channels
    (1, 8, 8, 256, 19, 1, 1, 0, 1),       # classifier: 19 output channels (generic transposing path)
    (1, 40, 48, 64, 64, 3, 1, 12, 12),    # ASPP-like dilation: most taps of the border rows fall outside the image
    # Cin, Cout >= 256: the 256 x 256 wide-tile kernel (bf16), several pixel splits, ragged last stage
    (1, 33, 40, 256, 256, 3, 1, 2, 2),
    (2, 21, 31, 512, 256, 1, 1, 0, 1),
    (1, 24, 40, 256, 512, 3, 2, 1, 1),
    # 1x1 / stride 1, whole 256 x 256 tiles, pixels % 64 == 0: conv_wgrad_pw_lw_kernel (one wave per SIMD, 32-pixel stages)
    (1, 8, 64, 256, 512, 1, 1, 0, 1),     # 16 stages in one split... or several: ring fill, steady state, drain
    (2, 16, 32, 512, 256, 1, 1, 0, 1),    # two Cin tiles
    (1, 2, 32, 256, 256, 1, 1, 0, 1),     # two stages: fewer than the prologue stages
    # W % 64 == 0, 3x3 / stride 1 / 'same': the row-buffer kernel (bf16; one dy stage per kernel row, taps at row offsets)
    (2, 9, 64, 64, 128, 3, 1, 1, 1),      # 18 stages: ring fill, steady state and drain; border rows
    (1, 20, 128, 128, 128, 3, 1, 2, 2),   # dilation 2, two stages per image row
    (1, 12, 64, 304, 256, 3, 1, 1, 1),    # 304 input channels: ragged third Cin tile, two Cout tiles
    (1, 40, 64, 72, 40, 3, 1, 16, 16),    # largest dilation the row buffer holds, ragged channels
    (1, 6, 64, 64, 64, 3, 1, 1, 1),       # fewer stages than the ring holds
    # several Cout and Cin tiles, ragged channel tiles at the dilations the trunk uses
    (1, 16, 128, 256, 512, 3, 1, 2, 2),   # four Cout tiles, two Cin tiles, 32 stages: ring fill / steady state / drain
    (2, 8, 64, 128, 320, 3, 1, 4, 4),     # ragged third Cout tile, dilation 4
    (1, 3, 64, 64, 256, 3, 1, 1, 1),      # three stages per split: shorter than the ring
    # Cout % 128 == 0, dil <= 8: conv_wgrad_lw_kernel (one wave per SIMD, generated stage loop; a ragged last Cin tile is masked)
    (1, 6, 64, 200, 128, 3, 1, 2, 2),     # 200 input channels: 72 of the second Cin tile
    (1, 2, 64, 128, 128, 3, 1, 8, 8),     # two stages, the largest dilation: kernel rows 0 and 2 never meet the image
    (3, 5, 192, 256, 128, 3, 1, 5, 5),    # three stages per image row, odd dilation, splits that end inside an image
    (2, 7, 64, 128, 384, 3, 1, 1, 1),     # one stage per image row (first and last tile at once), three Cout tiles
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(K, dt, case):
    N, H, W, Cin, Cout, k, s, p, d = case
    x = q(rnd(N, Cin, H, W), dt)
    Ho, Wo = orc.conv_out(H, k, s, p, d), orc.conv_out(W, k, s, p, d)
    gy = q(rnd(N, Cout, Ho, Wo), dt)
    ref = orc.conv2d_wgrad(x, gy, (Cout, Cin, k, k), stride=s, pad=p, dil=d)
    ldy = ((Cout + 7) // 8) * 8 + 8
    gyd = dev_nhwc(gy, dt, ld=ldy) if Cout % 8 else dev_nhwc(gy, dt)
    dw = torch.full((Cout, Cin, k, k), 3.0, device="cuda")
    K.conv2d_wgrad(dev_nhwc(x, dt, ld=Cin + 16), gyd, dw, s, p, d)
    if dt == "bf16" and k == 3 and s == 1 and p == d and W % 64 == 0 and d <= 16:
        # the row-buffer cases: one wave per SIMD where Cout is a multiple of 128 (dil <= 8), else the 8-wave kernel
        selected("conv_wgrad_lw_kernel" if Cout % 128 == 0 and d <= 8 else "conv_wgrad_row_kernel", f"wgrad {case}")
    elif dt == "bf16" and Cin >= 256 and Cout >= 256:
        # the 256 x 256 tile cases: one wave per SIMD for 1x1 / stride 1 with whole tiles and stages, else the 8-wave kernel
        pw_lw = k == 1 and s == 1 and p == 0 and Cin % 256 == 0 and Cout % 256 == 0 and (N * H * W) % 64 == 0
        selected("conv_wgrad_pw_lw_kernel" if pw_lw else "conv_wgrad_wide_kernel", f"wgrad {case}")
    assert_close(dw.cpu().numpy(), ref, dt, f"wgrad {case}")
    K.conv2d_wgrad(dev_nhwc(x, dt), gyd, dw, s, p, d, accumulate=True)
    assert_close(dw.cpu().numpy(), 2 * ref, dt, f"wgrad accumulate {case}")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_strided_dgrad_via_zero_insert(K, dt):
    """Input gradient of the stride-2 convs (3x3 pad 1 and the 1x1 projection): zero-insert the output gradient, then the
    ordinary stride-1 dgrad conv."""
    from kdcc_amd._lib import KD_PACK_DGRAD
    for (N, H, W, Cin, Cout, k, p) in [(2, 16, 24, 64, 64, 3, 1), (1, 12, 20, 64, 128, 1, 0)]:
        w = q(rnd(Cout, Cin, k, k, scale=0.1), dt)
        Ho, Wo = orc.conv_out(H, k, 2, p, 1), orc.conv_out(W, k, 2, p, 1)
        gy = q(rnd(N, Cout, Ho, Wo), dt)
        ref = orc.conv2d_dgrad(gy, w, (N, Cin, H, W), stride=2, pad=p)
        up = K.zero_insert(dev_nhwc(gy, dt), 2, (H, W))
        assert float(up[:, 1::2].abs().max()) == 0 and float(up[:, :, 1::2].abs().max()) == 0
        out = torch.empty((N, H, W, Cin), dtype=DT[dt], device="cuda")
        K.conv2d(up, K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt], KD_PACK_DGRAD), 1, (k - 1) - p, 1, out_raw=out)
        assert_close(host_nchw(out), ref, dt, f"strided dgrad k={k}")


def _torch_ref(fn, *shape_seed):
    return fn


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_maxpool_upsample_backward(K, dt, golden):
    """Backward of the pool / bilinear kernels against autograd of the reference's own functional calls (stock torch CPU:
    nn.MaxPool2d(3,2,1), F.interpolate(bilinear, align_corners=True)); ties in the pool input exercise the first-maximum rule."""
    import torch.nn.functional as F
    x = q(rnd(2, 16, 13, 18), dt)
    x[0, :, 2:5, 3:6] = x[0, :, 2:3, 3:4]          # a plateau: several equal maxima inside windows
    gy = q(rnd(2, 16, 7, 9), dt)
    xt = torch.from_numpy(x).requires_grad_(True)
    F.max_pool2d(xt, 3, 2, 1).backward(torch.from_numpy(gy))
    got = K.maxpool3x3s2_bwd(dev_nhwc(x, dt), dev_nhwc(gy, dt))
    assert_close(host_nchw(got), xt.grad.numpy(), dt, "maxpool bwd")
    for (C, hin, hout) in [(16, (6, 9), (21, 33)), (19, (7, 10), (14, 20)), (8, (16, 32), (64, 128))]:
        g = q(rnd(2, C, *hout), dt)
        xi = torch.zeros((2, C) + hin, requires_grad=True)
        F.interpolate(xi, size=hout, mode="bilinear", align_corners=True).backward(torch.from_numpy(g))
        gd = torch.from_numpy(np.ascontiguousarray(g.transpose(0, 2, 3, 1))).to(DT[dt]).cuda()
        got = K.upsample_bilinear_ac_bwd(gd, hin, out_dtype=torch.float32)
        assert_close(host_nchw(got), xi.grad.numpy(), "f32" if dt == "f32" else dt, f"upsample bwd C={C}")
    # the ops.npz golden (outputs of the reference run) pins the torch CPU functional used above
    g = golden("ops")
    np.testing.assert_allclose(F.max_pool2d(torch.from_numpy(g["x"]), 3, 2, 1).numpy(), g["maxpool"], rtol=1e-6)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_relu_bn_bwd_channel_sums_bn_params_stem_wgrad(K, dt):
    N, H, W, Cc = 2, 9, 14, 32
    g, pre = q(rnd(N, Cc, H, W), dt), rnd(N, Cc, H, W)
    gamma, beta, mean, var = np.abs(rnd(Cc)) + 0.5, rnd(Cc) * 0.2, rnd(Cc) * 0.1, np.abs(rnd(Cc)) + 0.5
    act = q(orc.bn_eval(pre, gamma, beta, mean, var, relu=True), dt)
    res = q(rnd(N, Cc, H, W), dt)
    scale = (gamma / np.sqrt(var + 1e-5)).astype(np.float32)
    ref = orc.bn_eval_bwd(g, act, gamma, var, relu=True)
    got = K.relu_bn_bwd(dev_nhwc(g, dt), dev_nhwc(act, dt, ld=Cc + 16), torch.from_numpy(scale).cuda(), res=dev_nhwc(res, dt))
    assert_close(host_nchw(got), ref + res, dt, "relu_bn_bwd")
    # channel sums (+ product sums), global and per image, with the `sub` operand
    gq = host_nchw(got)                      # as stored: g_x + res
    s1, s2 = K.channel_sums(got, sub=dev_nhwc(res, dt), a=dev_nhwc(act, dt))
    gx = gq.astype(np.float64) - res
    np.testing.assert_allclose(s1.cpu().numpy(), gx.sum((0, 2, 3)), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(s2.cpu().numpy(), (gx * act).sum((0, 2, 3)), rtol=2e-3, atol=2e-3)
    p1, _ = K.channel_sums(got, per_image=True)
    np.testing.assert_allclose(p1.cpu().numpy(), gq.astype(np.float64).sum((2, 3)), rtol=2e-3, atol=2e-3)
    # eval-BN parameter gradients vs autograd of F.batch_norm(training=False) + relu on the CPU
    import torch.nn.functional as F
    gt, bt = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    y = torch.relu(F.batch_norm(torch.from_numpy(pre), torch.from_numpy(mean), torch.from_numpy(var), gt, bt, False, 0.0, 1e-5))
    y.backward(torch.from_numpy(g))
    gxd = K.relu_bn_bwd(dev_nhwc(g, "f32"), dev_nhwc(y.detach().numpy(), "f32"), torch.from_numpy(scale).cuda())
    s1, s2 = K.channel_sums(gxd, a=dev_nhwc(y.detach().numpy(), "f32"))
    dg, db = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    cu = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
    K.bn_eval_param_grads(s1, s2, cu(scale), cu(gamma), cu(beta), dg, db)
    np.testing.assert_allclose(dg.cpu().numpy(), gt.grad.numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), bt.grad.numpy(), rtol=1e-3, atol=1e-3)
    # stem weight gradient
    xs, gys = rnd(2, 3, 11, 270), q(rnd(2, 64, 11, 270), dt)
    dw = torch.empty((64, 3, 3, 3), device="cuda")
    K.stem_wgrad(torch.from_numpy(xs).cuda(), dev_nhwc(gys, dt), dw)
    assert_close(dw.cpu().numpy(), orc.conv2d_wgrad(xs, gys, (64, 3, 3, 3), pad=1), "f32", "stem wgrad")
    # broadcast add
    v = rnd(2, 32)
    yb = dev_nhwc(q(rnd(2, 32, 5, 7), dt), dt)
    y0 = host_nchw(yb).copy()
    K.broadcast_add(torch.from_numpy(v).cuda(), yb, alpha=0.5)
    assert_close(host_nchw(yb), y0 + 0.5 * v[:, :, None, None], dt, "broadcast add")


# ------------------------------------------------------------------------------------------------- small-shape path (CIFAR)
DIRECT_CASES = [
    # N, C, H, W, K, k, stride, pad, dil, groups, bias
    (4, 3, 32, 32, 16, 3, 1, 1, 1, 1, False),     # ResNet-20 stem
    (3, 16, 17, 15, 32, 3, 2, 1, 1, 1, False),    # stride 2, odd sizes
    (2, 16, 12, 12, 16, 3, 1, 1, 1, 16, True),    # depthwise 3x3 with bias (CIFAR cheap conv)
    (2, 16, 9, 11, 40, 1, 1, 0, 1, 1, True),      # pointwise with bias
    (1, 8, 20, 20, 8, 9, 1, 20, 5, 8, False),     # 9x9 dilation 5 depthwise on a non-multiple-of-16 channel count
    (2, 12, 10, 10, 18, 3, 1, 2, 2, 3, False),    # grouped, dilated
]


@pytest.mark.parametrize("case", DIRECT_CASES)
def test_direct_conv_fwd_dgrad_wgrad(K, case):
    N, Cc, H, W, Kk, k, s, p, d, g, has_b = case
    x, w = rnd(N, Cc, H, W), rnd(Kk, Cc // g, k, k, scale=0.3)
    b = rnd(Kk) if has_b else None
    cu = lambda a: None if a is None else torch.from_numpy(a).cuda()
    y = K.conv2d_direct(cu(x), cu(w), cu(b), s, p, d, g)
    ref = orc.conv2d_fwd(x, w, bias=b, stride=s, pad=p, dil=d, groups=g)
    assert_close(y.cpu().numpy(), ref, "f32", f"direct fwd {case}")
    gy = rnd(*ref.shape)
    gx = K.conv2d_direct_dgrad(cu(gy), cu(w), x.shape, s, p, d, g)
    assert_close(gx.cpu().numpy(), orc.conv2d_dgrad(gy, w, x.shape, stride=s, pad=p, dil=d, groups=g), "f32", f"direct dgrad {case}")
    gw, gb = K.conv2d_direct_wgrad(cu(x), cu(gy), w.shape, s, p, d, g, want_bias=has_b)
    assert_close(gw.cpu().numpy(), orc.conv2d_wgrad(x, gy, w.shape, stride=s, pad=p, dil=d, groups=g), "f32", f"direct wgrad {case}")
    if has_b:
        np.testing.assert_allclose(gb.cpu().numpy(), gy.sum((0, 2, 3)), rtol=1e-4, atol=1e-4)


def test_bn_train_golden_and_oracle(K, golden):
    """BatchNorm2d in training mode + fused ReLU: forward, input / weight / bias gradients against the reference's own run
    (tests/golden/ops.npz, tools/make_golden.py:g_ops) and the oracle; running statistics; eval mode."""
    g = golden("ops")
    x, gamma, beta = g["x"], g["bn_gamma"], g["bn_beta"]
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    rm, rv = torch.zeros(6, device="cuda"), torch.ones(6, device="cuda")
    y, mean, invstd = K.bn2d_fwd(cu(x), cu(gamma), cu(beta), rm, rv, True, 0.1, 1e-5, relu=True)
    np.testing.assert_allclose(y.cpu().numpy(), g["bn_y"], rtol=1e-4, atol=1e-5)
    dx, dg, db = K.bn2d_bwd(cu(g["bn_gy"]), cu(x), y, cu(gamma), mean, invstd, True, relu=True)
    np.testing.assert_allclose(dx.cpu().numpy(), g["bn_gx"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(dg.cpu().numpy(), g["bn_ggamma"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), g["bn_gbeta"], rtol=1e-4, atol=1e-5)
    # oracle agreement (incl. the saved statistics) and nn.BatchNorm2d's running-stat update rule
    yo, mo, io = orc.bn_train_fwd(x, gamma, beta, relu=True)
    np.testing.assert_allclose(mean.cpu().numpy(), mo, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(invstd.cpu().numpy(), io, rtol=1e-5)
    n = x.size // 6
    xc = x.transpose(1, 0, 2, 3).reshape(6, -1).astype(np.float64)
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * xc.mean(1), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rv.cpu().numpy(), 0.9 + 0.1 * xc.var(1) * n / (n - 1), rtol=1e-5)
    # eval mode: running statistics, no update
    rm0, rv0 = rm.clone(), rv.clone()
    ye, _, _ = K.bn2d_fwd(cu(x), cu(gamma), cu(beta), rm, rv, False, 0.1, 1e-5, relu=False)
    np.testing.assert_allclose(ye.cpu().numpy(), orc.bn_eval(x, gamma, beta, rm0.cpu().numpy(), rv0.cpu().numpy()), rtol=1e-4, atol=1e-5)
    assert torch.equal(rm, rm0) and torch.equal(rv, rv0)


def test_cheap_conv_block_module_small_channels(K, golden):
    """DepthwiseSeparableBlock.forward as a module on the CIFAR geometry (16 channels, 3x3 / d1) and on the golden's
    k9/d5 case: forward and every gradient against the reference's outputs (tests/golden/dwsep.npz)."""
    from kdcc_amd.models.students import DepthwiseSeparableBlock
    g = golden("dwsep")
    for tag in ("k3d1", "k9d5"):
        Cc, Co, k, p, d, H, W = [int(v) for v in g[f"{tag}.cfg"]]
        blk = DepthwiseSeparableBlock(Cc, Co, k, p, d, Cc, None).cuda()
        with torch.no_grad():
            blk.separable_conv.weight.copy_(torch.from_numpy(g[f"{tag}.w_dw"]))
            blk.pointwise_conv.weight.copy_(torch.from_numpy(g[f"{tag}.w_pw"]))
        x = torch.from_numpy(g[f"{tag}.x"]).cuda().requires_grad_(True)
        y = blk(x)
        y.backward(torch.from_numpy(g[f"{tag}.gy"]).cuda())
        assert_close(y.detach().cpu().numpy(), g[f"{tag}.y"], "f32", f"{tag} y")
        assert_close(x.grad.cpu().numpy(), g[f"{tag}.gx"], "f32", f"{tag} gx")
        assert_close(blk.separable_conv.weight.grad.cpu().numpy(), g[f"{tag}.gw_dw"], "f32", f"{tag} gw_dw")
        assert_close(blk.pointwise_conv.weight.grad.cpu().numpy(), g[f"{tag}.gw_pw"], "f32", f"{tag} gw_pw")


@pytest.mark.parametrize("dual", [False, True])
def test_conv_output_sums_feed_the_image_pooling(K, dual):
    """conv2d(out_sums=): the ping-pong 1x1 kernel sums its output per channel over blocks of 128 pixels in its epilogue (the tensor
    the ASPP image-pooling branch averages, models/deeplabv3/deeplabv3.py:59-62, produced by mod7's last block) and
    aspp_image_pool(sums=) finishes the branch without reading the map.  Against the sums of the stored tensor (exact per block up to
    the fp32 order), against the separate-pass pooling, and per image: every image has its own rows."""
    dt = "bf16"
    N, H, W, C1, C2, Cout, red = 3, 64, 256, 128, 64, 512, 64
    x1, x2 = q(rnd(N, C1, H, W), dt), q(rnd(N, C2, H, W), dt)
    x1[1] *= 3.0          # images differ in scale: rows of different images must not mix
    w = q(rnd(Cout, C1 + C2 if dual else C1, 1, 1, scale=0.08), dt)
    wp = K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt])
    out = torch.zeros((N, H, W, Cout), dtype=DT[dt], device="cuda")
    sums = []
    if dual:
        K.conv2d(dev_nhwc(x1, dt), wp, x2=dev_nhwc(x2, dt), out_raw=out, out_sums=sums)
        selected("conv_igemm_persist_kernel<pp,dual>", "output sums, dual")
    else:
        K.conv2d(dev_nhwc(x1, dt), wp, out_raw=out, out_sums=sums)
        selected("conv_igemm_persist_kernel<pp>", "output sums")
    assert len(sums) == 1 and tuple(sums[0].shape) == (N * H * W // 128, 2, Cout)
    part = sums[0].cpu().numpy()
    ref_out = orc.conv2d_fwd(np.concatenate([x1, x2], 1) if dual else x1, w)
    assert_close(host_nchw(out), ref_out, dt, "conv with output sums")
    blocks = out.float().reshape(N * H * W // 128, 128, Cout).sum(1).cpu().numpy()      # sums of the STORED (rounded) values
    np.testing.assert_allclose(part[:, 0], blocks, rtol=2e-5, atol=2e-4)
    assert not part[:, 1].any()
    wi = rnd(red, Cout, scale=0.05)
    sc, sh = torch.from_numpy(rnd(red) * 0.2 + 1.0).cuda(), torch.from_numpy(rnd(red) * 0.1).cuda()
    a = torch.zeros((N, H, W, red), dtype=DT[dt], device="cuda")
    b = torch.zeros_like(a)
    K.aspp_image_pool(out, torch.from_numpy(wi).cuda(), sc, sh, a)
    K.aspp_image_pool(out, torch.from_numpy(wi).cuda(), sc, sh, b, sums=sums[0])
    assert float((a.float() - b.float()).abs().max()) <= 2.0 ** -7 * float(a.float().abs().max())
    mean = host_nchw(out).mean(axis=(2, 3))                                               # (N, Cout) of the stored tensor
    want = np.maximum((mean @ wi.T) * sc.cpu().numpy() + sh.cpu().numpy(), 0)
    got = b.float().cpu().numpy()[:, 0, 0, :]
    np.testing.assert_allclose(got, want, rtol=2e-2, atol=2e-3)
    assert float((b.float() - b[:, :1, :1, :].float()).abs().max()) == 0.0                  # broadcast over the pixels


@pytest.mark.parametrize("case", [(1, 224, 256, 64, 19, 1), (2, 128, 256, 128, 19, 2), (1, 240, 512, 64, 32, 1), (1, 224, 256, 64, 5, 1)])
def test_conv_classifier_epilogue(K, case):
    """conv2d(cls_w=, cls_out=): the 1x1 classifier of the reference's `final` head (models/deeplabv3/deeplabv3.py:127-139: conv3x3 -> BN ->
    ReLU -> conv1x1 onto the classes) applied in the epilogue of conv_row_lw_kernel<16> to the activation it would have stored.  Against
    the two-launch form on the same kernels (the activation is bit for bit the stored tensor, the weights the same bf16 values: only the
    fp32 order of the 256-term sums differs) and against the oracle; pixel tiles in both directions, two images, 19 / 32 / 5 classes,
    a stale output buffer."""
    dt = "bf16"
    N, H, W, Cin, ncls, d = case
    Cout = 256
    x = q(rnd(N, Cin, H, W), dt)
    w = q(rnd(Cout, Cin, 3, 3, scale=(2.0 / (9 * Cin)) ** 0.5), dt)
    wc = q(rnd(ncls, Cout, 1, 1, scale=0.08), dt)
    sc, sh = (rnd(Cout) * 0.2 + 1.0).astype(np.float32), (rnd(Cout) * 0.3).astype(np.float32)
    scd, shd = torch.from_numpy(sc).cuda(), torch.from_numpy(sh).cuda()
    wp = K.pack_conv_weight(torch.from_numpy(w).cuda(), DT[dt])
    wcd = torch.zeros((32, Cout, 1, 1), device="cuda")
    wcd[:ncls] = torch.from_numpy(wc).cuda()
    wcp32 = K.pack_conv_weight(wcd, DT[dt])
    xd = dev_nhwc(x, dt)
    assert K.conv_cls_ok(xd, Cout, 3, d)
    fused = torch.full((N, H, W, ncls), 7.0, device="cuda")
    K.conv2d(xd, wp, 1, d, d, act_scale=scd, act_shift=shd, act_relu=True, cls_w=wcp32, cls_out=fused)
    selected("conv_row_lw_kernel", f"classifier epilogue {case}")
    act = torch.zeros((N, H, W, Cout), dtype=DT[dt], device="cuda")
    K.conv2d(xd, wp, 1, d, d, out_act=act, act_scale=scd, act_shift=shd, act_relu=True)
    two = torch.einsum("nhwc,kc->nhwk", act.float().double(), torch.from_numpy(wc[:, :, 0, 0]).cuda().double()).float()      # (the stored activation x the same bf16 weights)
    a, b = fused.cpu().numpy(), two.cpu().numpy()
    scale = np.abs(b).max()
    assert np.isfinite(a).all() and np.abs(a - b).max() <= 2e-5 * scale + 1e-6, (case, np.abs(a - b).max(), scale)
    ref_act = np.maximum(orc.conv2d_fwd(x, w, pad=d, dil=d) * sc[None, :, None, None] + sh[None, :, None, None], 0)
    ref = np.einsum("nchw,kc->nhwk", q(ref_act.astype(np.float32), dt), wc[:, :, 0, 0])
    l2 = float(np.sqrt(((a - ref) ** 2).sum() / max((ref ** 2).sum(), 1e-30)))
    assert l2 <= 4e-3, (case, l2)
    # refusals: anything else in the epilogue, another tile shape
    with pytest.raises(K.ClsUnsupported):
        K.conv2d(xd, wp, 1, d, d, out_act=act, act_scale=scd, act_shift=shd, act_relu=True, cls_w=wcp32, cls_out=fused)
