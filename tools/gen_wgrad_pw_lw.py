#!/usr/bin/env python3
"""Generator of csrc/wgrad_pw_lw_body.inc: the hand-scheduled stage loop of conv_wgrad_pw_lw_kernel (csrc/pw_wgrad.hip), the weight
gradient of the dense 1x1 / stride-1 convolutions and of the cheap-conv blocks' pointwise convolutions (reference: loss.backward()
through models/encoders/wider_resnet.py:143-167 and models/students/transform_blocks/depthwise_separable_conv.py:9-13,
trainer/classification_trainer.py:37-39) with ONE wave per SIMD.

Same decomposition and K order as conv_wgrad_wide_kernel<true> -- a workgroup owns (256 Cout x 256 Cin tile, pixel split) and walks its
pixels 32 at a time into the same fp32 accumulation chains, so the slabs are bit-identical -- , other workers: 4 waves of 512 registers,
wave (wm, wn) = 128 Cout x 128 Cin = 64 accumulator tiles in a[0:255], two fragment sets in v[128:255].  A K stage is 32 pixels: four
8-KiB LDS images [32 pixels][128 channels] (dy channels 0-127 / 128-255, activations 0-127 / 128-255; source-side swizzle and
transposing fragment reads as in pw_wgrad_tr_kernel), four 32-KiB stages in a ring.

Per stage st (64 MFMAs on fragment set st & 1):
  first half : 32 MFMAs; the wave's 8 LDS-DMA pieces of stage st + 3 into the slot of stage st - 1, the staging iterator;
               s_waitcnt vmcnt(16) lgkmcnt(0) -- stage st + 1 has landed, all reads of stage st have returned -- and ONE barrier;
  second half: 32 MFMAs; the 32 ds_read_b64_tr_b16 of stage st + 1 into the other fragment set.
Unrolled over the four ring slots (slot and fragment set are immediates), everything that is not an MFMA in groups of <= 3 instructions
dealt evenly over the gaps, counted lgkmcnt waits from the in-order queue (tools/gen_wgrad_lw.py's machinery).

usage: python tools/gen_wgrad_pw_lw.py   (rewrites csrc/wgrad_pw_lw_body.inc; `--check` exits 1 when the file is stale: tests/test_abi.py)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_wgrad_lw as G0  # noqa: E402  (Gen: emit / in-order queue / counted waits)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "knowledge-distillation-by-replacing-cheap-conv_amd", "csrc", "wgrad_pw_lw_body.inc")

STAGE = 32768
IMG = 8192
NI, NJ = 8, 8
S_YB, S_XB = "s[40:41]", "s[42:43]"       # source bases of the stage being staged
S_LEFT, S_CNT, S_M0, S_T0, S_T1 = 44, 45, 46, 47, 48
CLOBBER_S = list(range(40, 50))
SLACK_DS = SLACK_VM = 0                   # mutation hooks of tools/check_wgrad_lw.py


def acc(i, j):
    return 4 * (i * NJ + j)


def frag(s, what, k):
    base = 128 + 64 * s
    return base + 4 * k if what == "A" else base + 32 + 4 * k


def reads(ring, s):
    """the 32 transposing reads of the stage in ring slot `ring` into fragment set s, in the order the MFMAs need them"""
    p, off = ("v", (ring & 1) * STAGE) if ring < 2 else ("w", (ring & 1) * STAGE)
    a = lambda i: [("DS", f"ds_read_b64_tr_b16 v[{frag(s, 'A', i)}:{frag(s, 'A', i) + 1}], %[{p}a{i}] offset:{off}", f"A{s}{i}a"),
                   ("DS", f"ds_read_b64_tr_b16 v[{frag(s, 'A', i) + 2}:{frag(s, 'A', i) + 3}], %[{p}a{i}] offset:{off + 1024}", f"A{s}{i}b")]
    b = lambda j: [("DS", f"ds_read_b64_tr_b16 v[{frag(s, 'B', j)}:{frag(s, 'B', j) + 1}], %[{p}b{j}] offset:{off}", f"B{s}{j}a"),
                   ("DS", f"ds_read_b64_tr_b16 v[{frag(s, 'B', j) + 2}:{frag(s, 'B', j) + 3}], %[{p}b{j}] offset:{off + 1024}", f"B{s}{j}b")]
    out = a(0)
    for j in range(NJ):
        out += b(j)
    for i in range(1, NI):
        out += a(i)
    return out


def pieces(ring):
    """the wave's 8 pieces of a stage: pieces 2 w + j (j = 0, 1) of the four images -> groups"""
    out = []
    for q in range(4):
        for j in range(2):
            src, base = (f"%[voy{j * 2 + q}]", S_YB) if q < 2 else (f"%[vox{j * 2 + q - 2}]", S_XB)
            out.append([f"s_add_u32 m0, %[sldsw], {ring * STAGE + q * IMG + j * 1024}"])       # (M0 one MFMA gap ahead of its piece; prologue: an s_nop)
            out.append([("NOP",), ("VM", f"global_load_lds_dwordx4 {src}, {base}", f"p{q}{j}")])
    return out


def advance_iterator():
    """the stage being staged moves on by 32 pixels while stages are left (afterwards the last one is staged again: in bounds, never read)"""
    return [[f"s_sub_u32 s{S_LEFT}, s{S_LEFT}, 1", f"s_cmp_gt_i32 s{S_LEFT}, 0", f"s_cselect_b32 s{S_T0}, %[sdy], 0", f"s_cselect_b32 s{S_T1}, %[sdx], 0"],
            [f"s_max_i32 s{S_LEFT}, s{S_LEFT}, 0", "s_add_u32 s40, s40, s%d" % S_T0, "s_addc_u32 s41, s41, 0"],
            ["s_add_u32 s42, s42, s%d" % S_T1, "s_addc_u32 s43, s43, 0"]]


IN_LOOP = [False]


def put(g, ins):
    if isinstance(ins, tuple) and ins[0] == "NOP":
        if not IN_LOOP[0]:
            g.emit("s_nop 0")
    elif isinstance(ins, str):
        g.emit(ins)
    elif ins[0] == "DS":
        g.ds_op(ins[1], ins[2])
    else:
        g.vm_op(ins[1], ins[2])


def deal(slots, groups, first, last):
    n = len(groups)
    for k, grp in enumerate(groups):
        slots[first + (k * (last - first + 1)) // n] += grp


def body(ds_at_top, r):
    """stage st in ring slot r, fragment set r & 1; returns (instructions, LDS queue at its end)"""
    G0.SLACK_DS = SLACK_DS
    IN_LOOP[0] = True
    g = G0.Gen()
    g.ds = list(ds_at_top)
    s = r & 1
    slots = [[] for _ in range(65)]
    deal(slots, pieces((r + 3) & 3) + advance_iterator(), 0, 30)
    deal(slots, [[x] for x in reads((r + 1) & 3, s ^ 1)], 32, 63)
    k = 0
    for i in range(NI):
        for j in range(NJ):
            if k == 32:
                g.emit(f"s_waitcnt vmcnt({16 + SLACK_VM}) lgkmcnt(0)")      # stage st + 1 has landed (st + 2, st + 3 may be in flight); stage st is read
                g.ds, g.vm = [], []
                g.emit("s_barrier")
            g.wait_ds([f"B{s}{j}a", f"B{s}{j}b"] + ([f"A{s}{i}a", f"A{s}{i}b"] if j == 0 else []))
            a, b = frag(s, "A", i), frag(s, "B", j)
            g.emit(f"v_mfma_f32_16x16x32_bf16 a[{acc(i, j)}:{acc(i, j) + 3}], v[{a}:{a + 3}], v[{b}:{b + 3}], a[{acc(i, j)}:{acc(i, j) + 3}]")
            for ins in slots[k]:
                put(g, ins)
            k += 1
    G0.SLACK_DS = 0
    IN_LOOP[0] = False
    return g.L, list(g.ds)


def build():
    g = G0.Gen()
    e = g.emit
    e("s_cmp_eq_u32 %[snst], 0"); e("s_cbranch_scc1 WGP_SKIP_%=")
    e(f"s_mov_b32 s{S_M0}, m0")
    e(f"s_mov_b64 {S_YB}, %[syb]"); e(f"s_mov_b64 {S_XB}, %[sxb]")
    e(f"s_mov_b32 s{S_LEFT}, %[snst]"); e(f"s_mov_b32 s{S_CNT}, %[snst]")
    for st in range(3):                       # prologue: stages 0, 1, 2
        for grp in pieces(st) + advance_iterator():
            for ins in grp:
                put(g, ins)
    e(f"s_waitcnt vmcnt({16 + SLACK_VM})")    # stage 0 has landed
    g.ds, g.vm = [], []
    e("s_barrier")
    for ins in reads(0, 0):
        put(g, ins)
    e("s_waitcnt lgkmcnt(0)")
    e("WGP_LOOP_%=:")
    t = []
    for r in range(8):                        # (two warm-up passes: the queue at a body's top is that of the steady state)
        _, t = body(t, r & 3)
    top = list(t)
    for r in range(4):
        L, t = body(t, r)
        g.L += L
        e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1"); e(f"s_cmp_lg_u32 s{S_CNT}, 0")
        e("s_cbranch_scc1 WGP_LOOP_%=" if r == 3 else "s_cbranch_scc0 WGP_DONE_%=")
    assert t == top, "the LDS queue at the loop top is not a fixpoint of the four bodies"
    e("WGP_DONE_%=:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e(f"s_mov_b32 m0, s{S_M0}")
    e("s_nop 15"); e("s_nop 15")
    e("WGP_SKIP_%=:")
    return g.L


def store_block():
    """accumulators -> the fp32 partial slab: tile (i, j), row r: out + lane offset + (16 i + r) rows + 64 j bytes"""
    L = []
    for i in range(NI):
        for r in range(4):
            L.append(f"v_mov_b32 v127, {16 * i + r}")
            L.append("v_mad_u32_u24 v127, v127, %[vcin4], %[vob]")
            for j in range(NJ):
                L.append(f"global_store_dword v127, a{acc(i, j) + r}, %[sout] offset:{64 * j}")
    L.append("s_waitcnt vmcnt(0)")
    return L


def render():
    o = ["// GENERATED by tools/gen_wgrad_pw_lw.py -- do not edit (tests/test_abi.py checks it is current)", "",
         "#define WGRAD_PW_LW_LOOP_ASM \\", G0.cstr(build()), "",
         "#define WGRAD_PW_LW_STORE_ASM \\", G0.cstr(store_block()), "",
         "#define WGRAD_PW_LW_CLOBBER_S " + ", ".join(f'"s{i}"' for i in CLOBBER_S),
         "#define WGRAD_PW_LW_CLOBBER_V " + ", ".join(f'"v{i}"' for i in range(127, 256)), ""]
    return "\n".join(o)


if __name__ == "__main__":
    txt = render()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == txt else 1)
    with open(OUT, "w") as f:
        f.write(txt)
    print(f"wrote {OUT}: {txt.count(chr(10))} lines")
