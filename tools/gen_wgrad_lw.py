#!/usr/bin/env python3
"""Generator of csrc/wgrad_lw_body.inc: the hand-scheduled stage loop of conv_wgrad_lw_kernel (csrc/pw_wgrad.hip), the dense
3x3 / stride-1 / 'same' weight gradient (reference: loss.backward() through models/encoders/wider_resnet.py:124-137 in mode B,
trainer/classification_trainer.py:37-39) with ONE wave per SIMD.

Same decomposition, LDS images and K order as conv_wgrad_row_kernel (pw_wgrad.hip) -- results are bit-identical -- : a workgroup owns
(kernel row ky, tile of 128 Cout x 128 Cin, pixel split); a K stage is 64 consecutive pixels of one image row: dy image 64 rows x 256 B
and the activations as a ROW BUFFER of up to 80 rows x 256 B (halo included, zeros outside the image) which the three kx taps read
at row offsets kx * dil; four 40-KiB stages in a ring.  What changes is who does the work: 4 waves (wave (wm, wn) = 64 Cout x [3 kx x 64
Cin] = 48 accumulator tiles in a[0:191]) instead of 8 (Cout % 128 == 0; Cin % 8 == 0: a ragged last Cin tile is masked), and every ds_read_b64_tr_b16 fragment read, every LDS-DMA piece and every
address update dealt between the 48 MFMAs of a k-step (32 pixels) by this generator.

Per stage st (two k-steps):
  k-step 0: 48 MFMAs on fragment set 0; the 32 transposing reads of (st, k-step 1) into set 1; the second half of stage st + 3's
            pieces (4 row-buffer pieces per wave); s_waitcnt vmcnt(18) lgkmcnt(0) -- stage st + 1 has landed, st + 2 and st + 3 may be in
            flight -- and ONE barrier: it publishes stage st + 1 and retires stage st (all of whose reads precede it).
  k-step 1: 48 MFMAs on set 1; the 32 reads of (st + 1, k-step 0) into set 0; the first half of stage st + 4's pieces (4 dy pieces +
            1 row-buffer piece per wave) into the slot of stage st.
The loop is unrolled over the FOUR ring slots, so that a stage's slot is an immediate: the 28 fragment addresses exist twice (slots 0 / 1
and slots 2 / 3: a ds offset field holds 16 bits) and never move (28 VALU per stage saved: -3.5 % measured), and M0 of a piece is one
s_add of a literal.
(The first version staged one stage later -- st + 3 into the slot of st - 1 -- and had the second half of a stage in flight for ONE stage
 time, ~1 us: less than the memory latency under load; the ablations of tools/wgrad_lw_ablate.sh showed the wait.)
Row-buffer pieces are issued under EXEC = the lanes whose pixel lies inside the image row (and inside the 64 + 2 dil rows that are read);
the other lanes' 16 bytes are zero-filled by a ds_write under the complementary mask (conv_lw.hip's scheme), skipped when that mask is
empty.  Every wave issues the same 9 vector-memory operations per stage, so the counted waits are the same immediates for all waves.
Everything that is not an MFMA -- the reads, the pieces (cut into groups of 2-4 instructions), the staging iterator (branch-free, in
groups that keep an SCC producer with its consumers) -- is dealt EVENLY over the 48 gaps of a k-step: a lone wave issues one instruction
per ~4 cycles and an MFMA occupies the pipe for 16, so a gap holds three instructions for free and every further one delays the matrix
pipe (the first version put a piece's ten instructions, or the iterator's 35, into one gap: ~350 of its 2230 cycles per stage).

usage: python tools/gen_wgrad_lw.py   (rewrites csrc/wgrad_lw_body.inc; `--check` exits 1 when the file is stale: tests/test_abi.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "knowledge-distillation-by-replacing-cheap-conv_amd", "csrc", "wgrad_lw_body.inc")

STAGE = 40960
XOFF = 16384
NI, NJ = 4, 12          # A fragments (16-row Cout tiles), B fragments ((kx, 16-column Cin tile) pairs: j = kx * 4 + c) per wave


def acc(i, j):
    return 4 * (i * NJ + j)


def frag(s, what, k):
    base = 128 + 64 * s
    return base + 4 * k if what == "A" else base + 16 + 4 * k


# scalar registers (clobbered)
S_YB, S_XB = "s[40:41]", "s[42:43]"       # source bases of the stage being staged: dy, activations (row buffer row 0)
S_LO, S_SPAN = 44, 45                     # its valid row-buffer rows [lo, lo + span)
S_X0, S_HO = 46, 47                       # its first pixel's column / image row
S_LEFT = 48                               # stages left to stage for real (beyond the split's end the last one is staged again, all rows masked)
S_T2 = 49
S_CNT = 51                                # stages left to compute
S_M0 = 54
S_T0, S_T1 = 55, 56
S_XB2 = "s[58:59]"                        # row-buffer source base of the SECOND half-stage (the first half's stage moved on in between)
S_LO2, S_SPAN2 = 60, 61
S_MASK = [62, 64, 66, 68, 70]             # lane masks of the five row-buffer pieces (pairs)
CLOBBER_S = list(range(40, 72))
VTS = [122, 123, 124, 125, 126]           # lane temporaries, one per row-buffer piece
VT = 127
V_CLOBBER0 = 122
PHASES = os.environ.get("KDCC_GEN_WGRAD_PHASES", "")    # experiment: "r0,r1,g0,g1;r0,r1,g0,g1" = gap ranges of the reads / the other groups in k-step 0 ; 1
ABL = int(os.environ.get("KDCC_GEN_WGRAD_ABL", "0"))   # TIMING ablations (tools/wgrad_lw_ablate.sh; results wrong): 1 no LDS-DMA, 2 no fragment reads, 4 no MFMAs, 8 no zero fill, 16 no address steps
SLACK_DS = SLACK_VM = 0                   # mutation hooks of tools/check_wgrad_lw.py's self-test: every counted wait that many operations too lax


class Gen:
    def __init__(self):
        self.L, self.ds, self.vm = [], [], []

    def emit(self, s):
        self.L.append(s)

    def ds_op(self, s, tag):
        self.ds.append(tag)
        self.emit(s)

    def vm_op(self, s, tag):
        self.vm.append(tag)
        self.emit(s)

    def wait_ds(self, tags):
        n = None
        for t in tags:
            if t in self.ds:
                k = self.ds[::-1].index(t)
                n = k if n is None else min(n, k)
        if n is None:
            return
        n = min(n + SLACK_DS, 15)
        self.emit(f"s_waitcnt lgkmcnt({n})")
        self.ds = self.ds[len(self.ds) - n:] if n else []


def reads(g, s, ks, slots, first, last, ring):
    """the 32 transposing reads of a k-step's fragments of the stage in ring slot `ring` into set s, one or two per slot over [first, last]"""
    ins = []
    a, off = ("v", (ring & 1) * STAGE + ks * 8192) if ring < 2 else ("w", (ring & 1) * STAGE + ks * 8192)
    for i in range(NI):
        f = frag(s, "A", i)
        ins.append((f"ds_read_b64_tr_b16 v[{f}:{f + 1}], %[{a}a{i}] offset:{off}", f"A{s}{i}a"))
        ins.append((f"ds_read_b64_tr_b16 v[{f + 2}:{f + 3}], %[{a}a{i}] offset:{off + 1024}", f"A{s}{i}b"))
    for j in range(NJ):
        f = frag(s, "B", j)
        ins.append((f"ds_read_b64_tr_b16 v[{f}:{f + 1}], %[{a}b{j}a] offset:{off}", f"B{s}{j}a"))
        ins.append((f"ds_read_b64_tr_b16 v[{f + 2}:{f + 3}], %[{a}b{j}b] offset:{off}", f"B{s}{j}b"))
    # order: the fragments the next k-step's first MFMAs need come first (A0, then B0.., A1 ..)
    order = [0, 1] + list(range(8, 8 + 2 * NJ)) + [2, 3, 4, 5, 6, 7]
    n = last - first + 1
    for k, idx in enumerate(order):
        slots[first + (k * n) // len(order)].append(("DS",) + ins[idx])


_PIECE = [0]


def dy_piece(k, ring):
    """-> groups of instructions; a group stays together between two MFMAs"""
    # (M0 one gap ahead of the piece: the MFMA between them is the wait state the hardware asks for; in the prologue, where the groups follow
    #  each other directly, the next piece's M0 write is not that piece's own, so an s_nop stands in)
    return [[f"s_add_u32 m0, %[sldsw], {ring * STAGE + k * 4096}"], [("NOP",), ("VM", f"global_load_lds_dwordx4 %[voy{k}], {S_YB}", f"y{k}")]]


def x_piece(k, ring, xb, lo, span):
    """Row-buffer piece k (rows 16 k + 4 wave + lane / 16 of the buffer): the lanes whose pixel lies inside the image row and inside the
    64 + 2 dil rows that are read are loaded, the others zero-filled under the complementary mask (conv_lw.hip's scheme).  Branch-free
    (but for the skip of an empty zero fill) and cut into four groups of 2-4 instructions: the first version had the ten of them between
    ONE pair of MFMAs and the matrix pipe waited ~30 cycles per piece."""
    _PIECE[0] += 1
    n = _PIECE[0]
    vt, m = VTS[k], f"s[{S_MASK[k]}:{S_MASK[k] + 1}]"
    vz = "%[vzl]" if ring < 2 else "%[vzh]"          # this lane's 16 bytes of piece 0 of the row buffer in ring slot 0 / 2
    return [[f"v_add_u32 v{vt}, {16 * k}, %[vr0]", f"v_subrev_u32 v{vt}, s{lo}, v{vt}"],
            [f"v_cmp_gt_u32 {m}, s{span}, v{vt}", f"s_add_u32 m0, %[sldsw], {ring * STAGE + XOFF + k * 4096}",
             f"s_and_b64 {m}, {m}, %[schm{k}]"],        # (and inside the Cin channels of the tensor: the last Cin tile may be ragged)
            [f"s_mov_b64 exec, {m}", ("VM", f"global_load_lds_dwordx4 %[vox{k}], {xb}", f"x{k}"), "s_mov_b64 exec, -1"],
            # (the zero fill is NOT tracked in the LDS queue: it is skipped when no lane needs it, and a counted wait that assumed it
            #  was issued would be too lax; leaving it out makes every count a lower bound of the operations issued behind a read)
            [f"s_not_b64 exec, {m}", f"s_cbranch_execz WGL_NZ{n}_%=", f"ds_write_b128 {vz}, %[vzero] offset:{(ring & 1) * STAGE + k * 4096}",
             f"WGL_NZ{n}_%=:", "s_mov_b64 exec, -1"]]


def advance_iterator():
    """the stage being staged moves on by 64 pixels while stages are left; afterwards the last one is staged again with every row masked.
    Branch-free, in groups that keep an SCC producer with its consumers (MFMA, LDS and vector-memory instructions leave SCC alone)."""
    return [[f"s_sub_u32 s{S_LEFT}, s{S_LEFT}, 1", f"s_cmp_gt_i32 s{S_LEFT}, 0", f"s_cselect_b32 s{S_T0}, %[sdy], 0", f"s_cselect_b32 s{S_T1}, %[sdx], 0",
             f"s_cselect_b32 s{S_T2}, 64, 0"],
            [f"s_max_i32 s{S_LEFT}, s{S_LEFT}, 0", f"s_add_u32 s40, s40, s{S_T0}", "s_addc_u32 s41, s41, 0"],
            [f"s_add_u32 s42, s42, s{S_T1}", "s_addc_u32 s43, s43, 0", f"s_add_u32 s{S_X0}, s{S_X0}, s{S_T2}"],
            [f"s_cmp_ge_u32 s{S_X0}, %[sW]", f"s_cselect_b32 s{S_X0}, 0, s{S_X0}", f"s_cselect_b32 s{S_T0}, 1, 0", f"s_add_u32 s{S_HO}, s{S_HO}, s{S_T0}"],
            [f"s_cmp_ge_u32 s{S_HO}, %[sH]", f"s_cselect_b32 s{S_HO}, 0, s{S_HO}"]] + masks()


def masks():
    """lo / span of the stage at (x0, ho): rows of the buffer inside the image row, none when the kernel row leaves the image or nothing is left"""
    return [[f"s_cmp_eq_u32 s{S_X0}, 0", f"s_cselect_b32 s{S_LO}, %[sd], 0", f"s_add_u32 s{S_T0}, s{S_X0}, 64"],                        # first tile of a row: pixels x0 - d + r < 0
            [f"s_cmp_eq_u32 s{S_T0}, %[sW]", f"s_cselect_b32 s{S_T0}, %[send1], %[send2]", f"s_sub_u32 s{S_SPAN}, s{S_T0}, s{S_LO}"],  # last tile: 64 + d rows, else 64 + 2 d
            [f"s_add_i32 s{S_T1}, s{S_HO}, %[skyd]", f"s_cmp_lt_u32 s{S_T1}, %[sH]", f"s_cselect_b32 s{S_SPAN}, s{S_SPAN}, 0"],          # (unsigned: hi < 0 wraps)
            [f"s_cmp_gt_i32 s{S_LEFT}, 0", f"s_cselect_b32 s{S_SPAN}, s{S_SPAN}, 0"]]


def deal(slots, groups, first, last):
    """the groups, in order, at evenly spaced gaps of [first, last]"""
    n = len(groups)
    for k, grp in enumerate(groups):
        slots[first + (k * (last - first + 1)) // n] += grp


IN_LOOP = [False]


def put(g, ins):
    if isinstance(ins, tuple) and ins[0] == "NOP":
        if not IN_LOOP[0]:
            g.emit("s_nop 0")
        return
    if isinstance(ins, str):
        if (ABL & 8 and ins.startswith("ds_write")) or (ABL & 16 and ins.startswith("v_add_u32 %[v")):
            return
        g.emit(ins)
    elif ins[0] == "DS":
        if ABL & 384 and ins[2].endswith("b"):      # 128: half the reads (half the bytes); 256: pairs as ONE ds_read_b128 (same bytes, half the instructions)
            return
        if ABL & 256:
            import re
            m = re.match(r"ds_read_b64_tr_b16 v\[(\d+):\d+\], (\S+) offset:(\d+)", ins[1])
            g.ds_op(f"ds_read_b128 v[{m.group(1)}:{int(m.group(1)) + 3}], {m.group(2)} offset:{m.group(3)}", ins[2])
            return
        if not ABL & 2:
            g.ds_op(ins[1], ins[2])
    elif ins[0] == "VM":
        if ABL & 32:      # the same memory request into registers instead of LDS (+ 64: and an ordinary ds_write_b128 of as many bytes)
            _, voff, base = ins[1].replace(",", " ").split(None, 2)
            g.vm_op(f"global_load_dwordx4 v[100:103], {voff}, {base}", ins[2])
            if ABL & 64:
                g.emit("ds_write_b128 %[vzl], v[104:107]")
        elif not ABL & 1:
            g.vm_op(ins[1], ins[2])
    else:
        raise ValueError(ins)


def kstep(g, s, first_stage_flag=None):
    """the 48 MFMAs on fragment set s"""
    out = []
    for i in range(NI):
        for j in range(NJ):
            a, b = frag(s, "A", i), frag(s, "B", j)
            out.append((i, j, f"v_mfma_f32_16x16x32_bf16 a[{acc(i, j)}:{acc(i, j) + 3}], v[{a}:{a + 3}], v[{b}:{b + 3}], a[{acc(i, j)}:{acc(i, j) + 3}]"))
    return out


def run_kstep(g, s, slots):
    for k, (i, j, m) in enumerate(kstep(g, s)):
        need = [f"B{s}{j}a", f"B{s}{j}b"] + ([f"A{s}{i}a", f"A{s}{i}b"] if j == 0 else [])
        g.wait_ds(need)
        if not ABL & 4:
            g.emit(m)
        for ins in slots[k]:
            put(g, ins)
    for ins in slots[48]:
        put(g, ins)


def build():
    g = Gen()
    e = g.emit
    _PIECE[0] = 0

    def flat(groups):
        for grp in groups:
            for ins in grp:
                put(g, ins)
    # ---- set-up: the staging iterator starts at stage 0 of the split (an empty split only writes its zero slab)
    e("s_cmp_eq_u32 %[snst], 0"); e("s_cbranch_scc1 WGL_SKIP_%=")
    e("s_mov_b32 s54, m0")
    e(f"s_mov_b64 {S_YB}, %[syb]"); e(f"s_mov_b64 {S_XB}, %[sxb]")
    e(f"s_mov_b32 s{S_X0}, %[sx0]"); e(f"s_mov_b32 s{S_HO}, %[sho]"); e(f"s_mov_b32 s{S_LEFT}, %[snst]"); e(f"s_mov_b32 s{S_CNT}, %[snst]")
    flat(masks())
    # ---- prologue: stages 0, 1 and 2 whole, the first half of stage 3
    for st in range(4):
        for k in range(4):
            flat(dy_piece(k, st))
        flat(x_piece(0, st, S_XB, S_LO, S_SPAN))
        if st < 3:
            for k in range(1, 5):
                flat(x_piece(k, st, S_XB, S_LO, S_SPAN))
        else:
            flat([keep_second_half()])      # the second half of stage 3 is issued in the loop's first k-step: keep its source and masks
        flat(advance_iterator())
    e(f"s_waitcnt vmcnt({23 + SLACK_VM}) lgkmcnt(0)")       # stage 0 has landed (9 + 9 + 5 operations of stages 1, 2, 3 may be outstanding)
    g.ds, g.vm = [], []
    e("s_barrier")
    slots = [[] for _ in range(49)]
    reads(g, 0, 0, slots, 0, 0, 0)
    for ins in slots[0]:
        put(g, ins)
    # the loop is entered with its first fragments landed; its counted waits are those of the STEADY state (the reads of set 0 dealt into
    # the previous stage's second k-step), which the loop bodies below are generated against (fixpoint of the queue at the loop top)
    e("s_waitcnt lgkmcnt(0)")
    e("WGL_LOOP_%=:")
    _, top = loop_body([], 0)
    for r in range(4):
        body, top2 = loop_body(top, r)
        assert top2 == top, "the LDS queue at the loop top is not a fixpoint"
        g.L += body
        e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1"); e(f"s_cmp_lg_u32 s{S_CNT}, 0")
        e("s_cbranch_scc1 WGL_LOOP_%=" if r == 3 else "s_cbranch_scc0 WGL_DONE_%=")
    e("WGL_DONE_%=:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_mov_b32 m0, s54")
    e("s_nop 15"); e("s_nop 15")
    e("WGL_SKIP_%=:")
    return g.L


def keep_second_half():
    return [f"s_mov_b64 {S_XB2}, {S_XB}", f"s_mov_b32 s{S_LO2}, s{S_LO}", f"s_mov_b32 s{S_SPAN2}, s{S_SPAN}"]


def loop_body(ds_at_top, r):
    """one stage whose ring slot is r; returns (instructions, LDS queue at its end)"""
    IN_LOOP[0] = True
    try:
        return _loop_body(ds_at_top, r)
    finally:
        IN_LOOP[0] = False


def _loop_body(ds_at_top, r):
    g = Gen()
    g.ds = list(ds_at_top)
    e = g.emit
    # ---- k-step 0: set 0; reads of (st, 1) into set 1; second half of stage st + 3
    slots = [[] for _ in range(49)]
    ph = [[int(x) for x in q.split(",")] for q in PHASES.split(";")] if PHASES else [[0, 38, 1, 46], [6, 46, 0, 47]]
    reads(g, 1, 1, slots, ph[0][0], ph[0][1], r)
    grp = []
    for k in range(1, 5):
        grp += x_piece(k, (r + 3) & 3, S_XB2, S_LO2, S_SPAN2)
    deal(slots, grp, ph[0][2], ph[0][3])
    run_kstep(g, 0, slots)
    e(f"s_waitcnt vmcnt({18 + SLACK_VM}) lgkmcnt(0)")        # stage st + 1 has landed: only the 18 operations of stages st + 2 and st + 3 are younger
    g.ds, g.vm = [], []
    e("s_barrier")
    # ---- k-step 1: set 1; reads of (st + 1, 0) into set 0; first half of stage st + 4 into the slot of stage st; the staging iterator:
    # this stage's second half keeps (source, masks), then the iterator moves on to stage st + 5
    slots = [[] for _ in range(49)]
    reads(g, 0, 0, slots, ph[1][0], ph[1][1], (r + 1) & 3)
    grp = []
    for k in range(4):
        grp += dy_piece(k, r)
    grp += x_piece(0, r, S_XB, S_LO, S_SPAN) + [keep_second_half()] + advance_iterator()
    deal(slots, grp, ph[1][2], ph[1][3])
    run_kstep(g, 1, slots)
    return g.L, list(g.ds)


def store_block():
    """accumulators -> the fp32 partial slab: tile (i, j = kx * 4 + c), row r: out[kx] + lane offset + (16 i + r) rows + 64 c bytes; the
    lanes of column tile c whose input channel lies beyond Cin (ragged last Cin tile) do not store"""
    L = []
    for c in range(4):
        L.append(f"s_mov_b64 exec, %[sstm{c}]")
        for i in range(NI):
            for r in range(4):
                L.append(f"v_mov_b32 v{VT}, {16 * i + r}")
                L.append(f"v_mad_u32_u24 v{VT}, v{VT}, %[vcin4], %[vob]")
                for kx in range(3):
                    L.append(f"global_store_dword v{VT}, a{acc(i, kx * 4 + c) + r}, %[sout{kx}] offset:{64 * c}")
    L.append("s_mov_b64 exec, -1")
    L.append("s_waitcnt vmcnt(0)")
    return L


def cstr(lines):
    return " \\\n".join('    "' + l + '\\n\\t"' for l in lines)


def render():
    o = ["// GENERATED by tools/gen_wgrad_lw.py -- do not edit (tests/test_abi.py checks it is current)", "",
         "#define WGRAD_LW_LOOP_ASM \\", cstr(build()), "",
         "#define WGRAD_LW_STORE_ASM \\", cstr(store_block()), "",
         "#define WGRAD_LW_ZERO_ASM \\", cstr([f"v_accvgpr_write_b32 a{n}, 0" for n in range(256)]), "",
         "#define WGRAD_LW_CLOBBER_S " + ", ".join(f'"s{i}"' for i in CLOBBER_S),
         "#define WGRAD_LW_CLOBBER_V " + ", ".join(f'"v{i}"' for i in range(100 if ABL & 32 else V_CLOBBER0, 256)), ""]
    return "\n".join(o)


if __name__ == "__main__":
    txt = render()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == txt else 1)
    with open(OUT, "w") as f:
        f.write(txt)
    print(f"wrote {OUT}: {txt.count(chr(10))} lines")
