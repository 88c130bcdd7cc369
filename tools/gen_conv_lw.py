#!/usr/bin/env python3
"""Generator of csrc/conv_lw_body.inc: the hand-scheduled main loop of conv_row_lw_kernel (csrc/conv_lw.hip).

One wave per SIMD (4 waves per workgroup, 512 registers per lane): a wave owns 128 pixels x 128 output channels of the
256 x 256 tile, its 256 accumulator registers live in a[0:255] for the whole kernel, the MFMA operand fragments in
v[128:255] (two sets of 8 A + 8 B fragments, one per k-step parity).  hipcc would schedule none of this the way the matrix
pipe wants it: a lone wave has no partner that hides its LDS reads and LDS-DMA issue, so every
one of them is DEALT into the shadow of the MFMAs by hand -- tools/ubench/lone_wave measures that stream at 1128 cycles per
k-step against 1117 for the bare MFMAs.

LW_TILE_ASM is ONE asm statement that runs every period of a tile (the fragments never live across compiler code).  A period P
= the three kx taps of one (64-channel block, kernel row) = 6 k-steps of 32 channels = 384 MFMAs per wave.  Per k-step p
(global k-step h = 6 P + p):
  * 64 v_mfma_f32_16x16x32_bf16 on fragment set p & 1;
  * 16 ds_read_b128: the fragments of k-step h + 1 into the other set (A from the row buffer of its period at the tap's row
    shift, B from slot (h + 1) & 3);
  * LDS-DMA, 1-KiB pieces (global_load_lds_dwordx4, source = SGPR base + per-lane offset VGPR, M0 written per piece):
      B of k-step h + 4 into slot h & 3 (4 pieces per wave; that slot's fragments were read during k-step h - 1),
      at p = 0, 1, 2 this wave's 4 + 3 + 3 pieces of the NEXT period's row buffer (EXEC = the lanes whose pixel lies inside
      the image row, from a compare of the lane's buffer row against two scalars; the other lanes' bytes are zero-filled by
      a ds_write under the complementary mask);
  * s_waitcnt vmcnt(N_p) lgkmcnt(0): everything issued up to k-step h - 2 has landed (N_p = pieces issued in h - 1 and h;
    in the first period after an epilogue the first two waits also leave that epilogue's stores outstanding); s_barrier.
So a B slot has two to three k-steps (>= 2048 cycles) to land and a row buffer two, and ONE barrier per k-step both publishes
what landed and frees what was read.  Two body copies (P even / odd: row buffer P & 1, B slot phase 2 (P & 1)) form the loop;
behind each, a few SALU instructions step the staging iterator (next kernel row / channel block, or the next TILE's first
period: the staging runs ahead across tiles).

usage: python tools/gen_conv_lw.py   (rewrites csrc/conv_lw_body.inc; `--check` exits 1 when the file is stale: tests/test_abi.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "knowledge-distillation-by-replacing-cheap-conv_amd", "csrc", "conv_lw_body.inc")

ABUF = 320 * 128          # one row buffer (320 rows x 128 B)
BSLOT = 256 * 64          # one B slot (256 output channels x 32 channels of K)
A_PIECES = ((0, 1, 2, 3), (4, 5, 6), (7, 8, 9), (), (), ())     # row-buffer pieces issued at k-step p
ISSUED = [(8 if os.environ.get("KDCC_GEN_LW_BREG", "0") == "1" else 4) + len(a) for a in A_PIECES]                         # VMEM operations per k-step and wave
# scalar registers owned by the statement (clobbered)
SB = "s[88:89]"           # running B source pointer (k-step h + 4)
SAN, SBN = "s[76:77]", "s[78:79]"     # the period being staged: row-buffer source base, weight base
SLO, SSPAN = "s80", "s81"             # ... its valid buffer rows [lo, lo + span)
SKY, SIN, SCNT, SFLAG = "s82", "s83", "s84", "s85"   # kernel-row position of the staged period, in-tile periods left to stage, periods left to compute, stores-outstanding flag
ST0, ST1, ST2 = "s86", "s87", "s91"
SMASK = "s[92:93]"        # lane mask of the row-buffer piece being staged
CLOBBER_S = ["s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93"]
NOBAR = os.environ.get("KDCC_GEN_LW_NOBAR", "0") == "1"
NOVM = os.environ.get("KDCC_GEN_LW_NOVM", "0") == "1"     # TIMING experiment (results wrong): the k-step waits ignore vmcnt = the loop never waits for a piece to land
BREG = os.environ.get("KDCC_GEN_LW_BREG", "0") == "1"     # TIMING experiment (results wrong): the weight operand straight into registers -- per k-step and wave eight global_load_dwordx4 instead of four LDS-DMA pieces + eight ds_read_b128
SPREAD = os.environ.get("KDCC_GEN_LW_SPREAD", "1") != "0"     # experiment switch: 0 = a row-buffer piece's eight instructions between ONE pair of MFMAs (rounds 4-5)
VT = "v127"               # lane temporary (clobbered)


def frag_regs(s):
    base = 128 + 64 * s
    return [base + 4 * i for i in range(8)], [base + 32 + 4 * j for j in range(8)]


BROKEN = False      # main() sets it while it renders LW_TILE_BROKEN_ASM (tuning build only, tests/test_lw_bitwise_gpu.py's self-test)


def kstep(par, p, zero=False):
    """zero: the tile's first k-step -- the MFMAs take 0 as their C operand (the accumulators still hold the previous tile).
    BROKEN: the even body's k-step 3 loses its barrier and wave 0 sleeps ~60 k cycles at its start, so the other waves run one
    k-step ahead of it -- they read fragments of pieces wave 0 has not staged yet, and stage into the slot it has yet to read: the
    defect class (a missing barrier) the bitwise A/B test has to catch, made deterministic."""
    cur = p & 1
    areg, breg = frag_regs(cur)
    nareg, nbreg = frag_regs(1 - cur)
    slots = [[] for _ in range(64)]
    # ---- fragment reads of the next k-step
    q = p + 1
    if q < 6:
        kx, ks, buf = q >> 1, q & 1, par
    else:
        kx, ks, buf = 0, 0, par ^ 1
    slot_next = (2 * par + p + 1) & 3
    reads = [f"ds_read_b128 v[{nareg[i]}:{nareg[i] + 3}], %[va{kx * 2 + ks}] offset:{i * 2048 + buf * ABUF}" for i in range(8)]
    if not BREG:
        reads += [f"ds_read_b128 v[{nbreg[j]}:{nbreg[j] + 3}], %[vb] offset:{slot_next * BSLOT + j * 1024}" for j in range(8)]
    for r, ins in enumerate(reads):
        slots[1 + 3 * r].append(ins)
    # ---- B pieces of k-step h + 4 into slot h & 3
    slot_cur = (2 * par + p) & 3
    if p == 2:
        slots[2].append(f"s_mov_b64 {SB}, {SBN}")          # positions 0 .. 3 of the period being staged
    for j, k in enumerate((4, 20, 36, 52)):
        if BREG:
            slots[k - 8 if k >= 8 else k].append(f"global_load_dwordx4 v[{96 + 8 * j}:{99 + 8 * j}], %[vob{j}], {SB}")
            slots[k].append(f"global_load_dwordx4 v[{100 + 8 * j}:{103 + 8 * j}], %[vob{j}], {SB} offset:512")
        elif SPREAD:     # (M0 one gap ahead of the piece: the MFMA between them is the wait state an s_nop provided)
            slots[k - 1].append(f"s_add_u32 m0, %[sldsB], {slot_cur * BSLOT + j * 1024}")
            slots[k].append(f"global_load_lds_dwordx4 %[vob{j}], {SB}")
        else:
            slots[k] += [f"s_add_u32 m0, %[sldsB], {slot_cur * BSLOT + j * 1024}", "s_nop 0",
                         f"global_load_lds_dwordx4 %[vob{j}], {SB}"]
    # advance the B pointer: +64 B to the second k-half, then on to the next tap
    if p & 1:
        slots[54] += ["s_add_u32 s88, s88, %[s2c]", "s_addc_u32 s89, s89, 0"]
    else:
        slots[54] += ["s_add_u32 s88, s88, 64", "s_addc_u32 s89, s89, 0"]
    # ---- row-buffer pieces of the period being staged
    at = (12, 28, 44, 60)
    for n, j in enumerate(A_PIECES[p]):
        nb = par ^ 1
        if SPREAD:
            # the piece's instructions in four gaps instead of one (round 6: a lone wave hides ~2 instructions per MFMA gap, eight in one gap
            # stall the matrix pipe): lane mask into an SGPR pair, load under it, zero fill under its complement (skipped when empty)
            tag = f"{par}{p}{'z' if zero else ''}{j}"
            slots[at[n]] += [f"v_add_u32 {VT}, {8 * j}, %[vr0]", f"v_subrev_u32 {VT}, {SLO}, {VT}"]      # buffer row of the lane - lo
            slots[at[n] + 1] += [f"v_cmp_gt_u32 {SMASK}, {SSPAN}, {VT}",                                  # lo <= row < lo + span
                                 f"s_add_u32 m0, %[sldsA], {nb * ABUF + j * 1024}"]
            slots[at[n] + 2] += [f"s_mov_b64 exec, {SMASK}", f"global_load_lds_dwordx4 %[voa{j}], {SAN}", "s_mov_b64 exec, -1"]
            slots[at[n] + 3] += [f"s_not_b64 exec, {SMASK}", f"s_cbranch_execz LWNZ{tag}_%=", f"ds_write_b128 %[vz{nb}], %[vzero] offset:{j * 1024}",
                                 f"LWNZ{tag}_%=:", "s_mov_b64 exec, -1"]
            continue
        slots[at[n]] += [f"s_add_u32 m0, %[sldsA], {nb * ABUF + j * 1024}",
                         f"v_add_u32 {VT}, {8 * j}, %[vr0]", f"v_subrev_u32 {VT}, {SLO}, {VT}",     # buffer row of the lane - lo
                         f"v_cmpx_gt_u32 vcc, {SSPAN}, {VT}",                                       # EXEC: lo <= row < lo + span
                         f"global_load_lds_dwordx4 %[voa{j}], {SAN}", "s_not_b64 exec, exec",
                         f"ds_write_b128 %[vz{nb}], %[vzero] offset:{j * 1024}", "s_mov_b64 exec, -1"]
    if SPREAD and p == 5:
        # the staging iterator, branch-free, in the last gaps of the period's last k-step (rounds 4-5: ~18 scalar instructions behind the barrier,
        # with the matrix pipe idle)
        for k, grp in zip((48, 50, 53, 56, 58, 60), step_iterator_groups()):
            slots[k] += grp
    L = []
    broken_here = BROKEN and par == 0 and p == 3 and not zero
    if broken_here:      # (%[sbrk] = 0: the regular schedule)
        L += ["s_cmp_eq_u32 %[sbrk], 0", f"s_cbranch_scc1 LWBRK{par}{p}_%=", "s_cmp_lg_u32 %[swv], 0", f"s_cbranch_scc1 LWBRK{par}{p}_%="] + \
             ["s_sleep 127"] * 8 + [f"LWBRK{par}{p}_%=:"]
    k = 0
    for i in range(8):
        for j in range(8):
            acc = 4 * (8 * i + j)
            L.append(f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], v[{breg[j]}:{breg[j] + 3}], v[{areg[i]}:{areg[i] + 3}], " + ("0" if zero else f"a[{acc}:{acc + 3}]"))
            L += slots[k]
            k += 1
    n = ISSUED[p] + ISSUED[p - 1]          # (p = 0: the previous period's last k-step)
    tag = f"{par}{p}z" if zero else f"{par}{p}"
    if p < 2:
        # right after an epilogue its stores are younger than the pieces this wait is for and older than this period's: leave
        # them outstanding (the counter saturates at 63)
        L += [f"s_cmp_eq_u32 {SFLAG}, 0", f"s_cbranch_scc1 LWN{tag}_%=", "s_waitcnt vmcnt(63) lgkmcnt(0)", f"s_branch LWD{tag}_%=",
              f"LWN{tag}_%=:", f"s_waitcnt vmcnt({63 if NOVM else n}) lgkmcnt(0)", f"LWD{tag}_%=:"]
        if p == 1:
            L.append(f"s_mov_b32 {SFLAG}, 0")
    else:
        L.append(f"s_waitcnt vmcnt({63 if NOVM else n}) lgkmcnt(0)")
    if broken_here:
        L += ["s_cmp_lg_u32 %[sbrk], 0", f"s_cbranch_scc1 LWNOBAR{par}{p}_%=", "s_barrier", f"LWNOBAR{par}{p}_%=:"]
    elif NOBAR and (p & 1) == 0:
        pass               # TIMING experiment (KDCC_GEN_LW_NOBAR=1, results wrong): what a barrier per TWO k-steps could buy at most
    else:
        L.append("s_barrier")
    return L


def step_iterator_groups():
    """step_iterator() without branches, in groups that keep an SCC producer with its consumers: both alternatives are computed, s_cselect picks"""
    return [[f"s_sub_u32 {SIN}, {SIN}, 1", f"s_add_u32 {SKY}, {SKY}, 1"],
            [f"s_cmp_eq_u32 {SKY}, %[snky]", f"s_cselect_b32 {ST0}, %[sdAw], %[sdAs]", f"s_cselect_b32 {ST1}, %[sdBw], %[sdBs]", f"s_cselect_b32 {SKY}, 0, {SKY}"],
            [f"s_ashr_i32 {ST2}, {ST0}, 31", f"s_add_u32 s76, s76, {ST0}", f"s_addc_u32 s77, s77, {ST2}"],
            [f"s_ashr_i32 {ST2}, {ST1}, 31", f"s_add_u32 s78, s78, {ST1}", f"s_addc_u32 s79, s79, {ST2}"],
            [f"s_cmp_gt_i32 {SIN}, 0", f"s_cselect_b64 {SAN}, {SAN}, %[sAnT]", f"s_cselect_b64 {SBN}, {SBN}, %[sBnT]"],
            [f"s_cmp_gt_i32 {SIN}, 0", f"s_cselect_b32 {SLO}, {SLO}, %[sloT]", f"s_cselect_b32 {SSPAN}, {SSPAN}, %[sspT]"]]


def step_iterator(par):
    """the staging iterator moves on by one period: next kernel row / channel block of this tile, or the next tile's first period"""
    if SPREAD:
        return []          # (dealt into k-step 5's gaps: step_iterator_groups)
    return [f"s_sub_u32 {SIN}, {SIN}, 1", f"s_cmp_gt_i32 {SIN}, 0", f"s_cbranch_scc0 LWT{par}_%=",
            f"s_add_u32 {SKY}, {SKY}, 1", f"s_cmp_eq_u32 {SKY}, %[snky]",
            f"s_cselect_b32 {ST0}, %[sdAw], %[sdAs]", f"s_cselect_b32 {ST1}, %[sdBw], %[sdBs]", f"s_cselect_b32 {SKY}, 0, {SKY}",
            f"s_ashr_i32 {ST2}, {ST0}, 31", f"s_add_u32 s76, s76, {ST0}", f"s_addc_u32 s77, s77, {ST2}",
            f"s_ashr_i32 {ST2}, {ST1}, 31", f"s_add_u32 s78, s78, {ST1}", f"s_addc_u32 s79, s79, {ST2}",
            f"s_branch LWU{par}_%=", f"LWT{par}_%=:",
            f"s_mov_b64 {SAN}, %[sAnT]", f"s_mov_b64 {SBN}, %[sBnT]", f"s_mov_b32 {SLO}, %[sloT]", f"s_mov_b32 {SSPAN}, %[sspT]",
            f"LWU{par}_%=:"]


def tile():
    L = ["s_mov_b32 s90, m0", f"s_mov_b64 {SB}, %[sBp]", f"s_mov_b64 {SAN}, %[sAn]", f"s_mov_b64 {SBN}, %[sBn]",
         f"s_mov_b32 {SLO}, %[slo]", f"s_mov_b32 {SSPAN}, %[ssp]", f"s_mov_b32 {SKY}, 1", f"s_sub_u32 {SIN}, %[snper], 1",
         f"s_mov_b32 {SCNT}, %[snper]", f"s_mov_b32 {SFLAG}, %[sflag]", "s_cmp_eq_u32 %[spar], 0", "s_cbranch_scc0 LWODDZ_%=",
         # the tile's first k-step multiplies into zero, then joins the period body behind its first k-step
         ] + kstep(0, 0, True) + ["s_branch LWEVEN1_%=", "LWODDZ_%=:"] + kstep(1, 0, True) + ["s_branch LWODD1_%=",
         "LWEVEN_%=:"]
    for p in range(6):
        L += kstep(0, p)
        if p == 0:
            L.append("LWEVEN1_%=:")
    L += step_iterator(0) + [f"s_sub_u32 {SCNT}, {SCNT}, 1", f"s_cmp_eq_u32 {SCNT}, 0", "s_cbranch_scc1 LWEND_%=", "LWODD_%=:"]
    for p in range(6):
        L += kstep(1, p)
        if p == 0:
            L.append("LWODD1_%=:")
    L += step_iterator(1) + [f"s_sub_u32 {SCNT}, {SCNT}, 1", f"s_cmp_lg_u32 {SCNT}, 0", "s_cbranch_scc1 LWEVEN_%=", "LWEND_%=:",
                             f"s_mov_b64 %[sBp], {SB}", "s_mov_b32 m0, s90"]
    return L


def refill():
    """the first k-step's fragments of a period: A from %[vaf] (row-buffer address of tap 0 / k-half 0), B from %[vbf] (its slot)"""
    areg, breg = frag_regs(0)
    L = [f"ds_read_b128 v[{areg[i]}:{areg[i] + 3}], %[vaf] offset:{i * 2048}" for i in range(8)]
    L += [f"ds_read_b128 v[{breg[j]}:{breg[j] + 3}], %[vbf] offset:{j * 1024}" for j in range(8)]
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]      # (no wave stages into that slot before every wave has read it)
    return L


def read_acc(i, jg):
    regs = [4 * (8 * i + 4 * jg + jj) + r for jj in range(4) for r in range(4)]
    return [f"v_accvgpr_read_b32 %{n}, a{a}" for n, a in enumerate(regs)]      # (the next tile's first k-step multiplies into zero)


# ---- 1x1 / stride 1 (conv_pw_lw_kernel): no row buffer, the A operand is staged per k-step like B ------------------------------------
# LDS: four A slots of 256 pixels x 64 B, then four B slots of 256 channels x 64 B.  Per k-step h: 64 MFMAs, the 16 fragment reads of
# k-step h + 1 (slots (h + 1) & 3), 4 + 4 DMA pieces of k-step h + 4 into slots h & 3, vmcnt(16) + barrier.  The loop body is four
# k-steps (one pass over the slots); the staging pointers run 4 k-steps ahead and move to the NEXT tile's operands before the last pass.
SPA, SPB = "s[76:77]", "s[78:79]"     # running A / B source pointers (k-step h + 4)


def kstep_1x1(p, zero=False):
    cur = p & 1
    areg, breg = frag_regs(cur)
    nareg, nbreg = frag_regs(1 - cur)
    slots = [[] for _ in range(64)]
    sn, sc = (p + 1) & 3, p & 3
    reads = [f"ds_read_b128 v[{nareg[i]}:{nareg[i] + 3}], %[va] offset:{sn * BSLOT + i * 1024}" for i in range(8)]
    reads += [f"ds_read_b128 v[{nbreg[j]}:{nbreg[j] + 3}], %[vb] offset:{sn * BSLOT + j * 1024}" for j in range(8)]
    for r, ins in enumerate(reads):
        slots[1 + 3 * r].append(ins)
    for j, k in enumerate((3, 17, 31, 45)):
        slots[k] += [f"s_add_u32 m0, %[sldsA], {sc * BSLOT + j * 1024}", "s_nop 0", f"global_load_lds_dwordx4 %[voa{j}], {SPA}"]
    for j, k in enumerate((10, 24, 38, 52)):
        slots[k] += [f"s_add_u32 m0, %[sldsB], {sc * BSLOT + j * 1024}", "s_nop 0", f"global_load_lds_dwordx4 %[vob{j}], {SPB}"]
    slots[54] += ["s_add_u32 s76, s76, 64", "s_addc_u32 s77, s77, 0", "s_add_u32 s78, s78, 64", "s_addc_u32 s79, s79, 0"]
    L = []
    k = 0
    for i in range(8):
        for j in range(8):
            acc = 4 * (8 * i + j)
            L.append(f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], v[{breg[j]}:{breg[j] + 3}], v[{areg[i]}:{areg[i] + 3}], " + ("0" if zero else f"a[{acc}:{acc + 3}]"))
            L += slots[k]
            k += 1
    tag = f"{p}z" if zero else f"{p}"
    if p < 2:
        L += [f"s_cmp_eq_u32 {SFLAG}, 0", f"s_cbranch_scc1 LPN{tag}_%=", "s_waitcnt vmcnt(63) lgkmcnt(0)", f"s_branch LPD{tag}_%=",
              f"LPN{tag}_%=:", "s_waitcnt vmcnt(16) lgkmcnt(0)", f"LPD{tag}_%=:"]
        if p == 1:
            L.append(f"s_mov_b32 {SFLAG}, 0")
    else:
        L.append("s_waitcnt vmcnt(16) lgkmcnt(0)")
    L.append("s_barrier")
    return L


def tile_1x1():
    L = ["s_mov_b32 s90, m0", f"s_mov_b64 {SPA}, %[sA]", f"s_mov_b64 {SPB}, %[sB]", f"s_mov_b32 {SCNT}, %[snit]", f"s_mov_b32 {SFLAG}, %[sflag]",
         "s_cmp_lg_u32 %[snit], 1", "s_cbranch_scc1 LPSAMEZ_%=", f"s_mov_b64 {SPA}, %[sAT]", f"s_mov_b64 {SPB}, %[sBT]", "LPSAMEZ_%=:",
         # the tile's first k-step multiplies into zero, then joins the loop body behind its first k-step
         ] + kstep_1x1(0, True) + ["s_branch LPLOOP1_%=",
         "LPLOOP_%=:",
         # the last pass stages the next tile's first four k-steps
         f"s_cmp_lg_u32 {SCNT}, 1", "s_cbranch_scc1 LPSAME_%=", f"s_mov_b64 {SPA}, %[sAT]", f"s_mov_b64 {SPB}, %[sBT]", "LPSAME_%=:"]
    for p in range(4):
        L += kstep_1x1(p)
        if p == 0:
            L.append("LPLOOP1_%=:")
    L += [f"s_sub_u32 {SCNT}, {SCNT}, 1", f"s_cmp_lg_u32 {SCNT}, 0", "s_cbranch_scc1 LPLOOP_%=", "s_mov_b32 m0, s90"]
    return L


def refill_1x1():
    areg, breg = frag_regs(0)
    L = [f"ds_read_b128 v[{areg[i]}:{areg[i] + 3}], %[va] offset:{i * 1024}" for i in range(8)]
    L += [f"ds_read_b128 v[{breg[j]}:{breg[j] + 3}], %[vb] offset:{j * 1024}" for j in range(8)]
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    return L


# ---- "duo": TWO workgroups per CU (conv_row_duo_kernel) -------------------------------------------------------------------------------
# What the lone-wave kernel cannot hide is its epilogue: one wave per SIMD converts, transposes and stores its 128 x 128 outputs with
# the matrix pipe idle (8-13 % of a 3x3 layer, a third of the 128-channel layers: tools/lw_ablate.sh).  Here a workgroup is 4 waves of
# 128 x 64 (128 accumulators in a[0:127], 256 registers per wave) on a 256 x 128 tile and 72 KiB of LDS, so that two INDEPENDENT
# workgroups share a CU: each SIMD holds one wave of either, both run this same hand-dealt stream, and while one workgroup is in its
# epilogue the other one's MFMAs have the pipe to themselves.
# K is staged in 32-channel periods: row buffers of 320 rows x 64 B (two), B in three 8-KiB slots (one per tap of a period).  A period
# P = the three kx taps of one (32-channel block, kernel row) = 3 k-steps; per k-step (P, kx): 32 MFMAs, the 12 fragment reads of the
# next k-step, the two B pieces of (P + 1, kx) into slot kx, and of the row buffers pieces 3, 4 of period P + 1 at kx = 0 and pieces
# 0, 1, 2 of period P + 2 at kx = 2 (that buffer was last read during (P, 1)); s_waitcnt vmcnt(pieces issued in this k-step) -- everything
# older has landed -- and one barrier.  The loop body is two periods (fragment set parity); tiles have an even number of periods.
DUO_ABUF, DUO_BSLOT, DUO_B0 = 320 * 64, 128 * 64, 2 * 320 * 64
D_SAN1, D_SBN1, D_LO1, D_SP1 = "s[76:77]", "s[78:79]", "s80", "s81"       # period P + 1
D_SAN2, D_SBN2, D_LO2, D_SP2 = "s[92:93]", "s[94:95]", "s96", "s97"       # period P + 2
DUO_CLOBBER_S = CLOBBER_S + ["s94", "s95", "s96", "s97"]
D_VT = "v31"


def duo_frags(sset):
    base = 32 + 48 * sset
    return [base + 4 * i for i in range(8)], [base + 32 + 4 * j for j in range(4)]


def duo_a_piece(j, buf, san, lo, span):
    return [f"s_add_u32 m0, %[sldsA], {buf * DUO_ABUF + j * 1024}", f"v_add_u32 {D_VT}, {16 * j}, %[vr0]", f"v_subrev_u32 {D_VT}, {lo}, {D_VT}",
            f"v_cmpx_gt_u32 vcc, {span}, {D_VT}", f"global_load_lds_dwordx4 %[voa{j}], {san}", "s_not_b64 exec, exec",
            f"ds_write_b128 %[vz{buf}], %[vzero] offset:{j * 1024}", "s_mov_b64 exec, -1"]


def duo_shift(tag):
    """period end: the period being staged moves on by one (P + 1 := P + 2; P + 2 := next kernel row / channel block of the tile, then
    the next tile's periods 0 and 1)"""
    return [f"s_mov_b64 {D_SAN1}, {D_SAN2}", f"s_mov_b64 {D_SBN1}, {D_SBN2}", f"s_mov_b32 {D_LO1}, {D_LO2}", f"s_mov_b32 {D_SP1}, {D_SP2}",
            f"s_sub_u32 {SIN}, {SIN}, 1", f"s_cmp_gt_i32 {SIN}, 0", f"s_cbranch_scc0 LDT{tag}_%=",
            f"s_add_u32 {SKY}, {SKY}, 1", f"s_cmp_eq_u32 {SKY}, %[snky]",
            f"s_cselect_b32 {ST0}, %[sdAw], %[sdAs]", f"s_cselect_b32 {ST1}, %[sdBw], %[sdBs]", f"s_cselect_b32 {SKY}, 0, {SKY}",
            f"s_ashr_i32 {ST2}, {ST0}, 31", f"s_add_u32 s92, s92, {ST0}", f"s_addc_u32 s93, s93, {ST2}",
            f"s_ashr_i32 {ST2}, {ST1}, 31", f"s_add_u32 s94, s94, {ST1}", f"s_addc_u32 s95, s95, {ST2}",
            f"s_branch LDU{tag}_%=", f"LDT{tag}_%=:",
            f"s_mov_b32 {D_LO2}, %[sloT]", f"s_mov_b32 {D_SP2}, %[sspT]", f"s_cmp_eq_u32 {SIN}, 0", f"s_cbranch_scc0 LDV{tag}_%=",
            f"s_mov_b64 {D_SAN2}, %[sAnT0]", f"s_mov_b64 {D_SBN2}, %[sBnT0]", f"s_branch LDU{tag}_%=", f"LDV{tag}_%=:",
            f"s_mov_b64 {D_SAN2}, %[sAnT1]", f"s_mov_b64 {D_SBN2}, %[sBnT1]", f"LDU{tag}_%=:"]


def duo_kstep(pp, zero=False):
    """pp = 0 .. 5: k-step of the two-period body (period pp // 3, tap pp % 3)"""
    per, kx, cur = pp // 3, pp % 3, pp & 1
    areg, breg = duo_frags(cur)
    nareg, nbreg = duo_frags(1 - cur)
    slots = [[] for _ in range(32)]
    nkx = (kx + 1) % 3
    nbuf = per if kx < 2 else 1 - per
    reads = [f"ds_read_b128 v[{nareg[i]}:{nareg[i] + 3}], %[va{nkx}] offset:{nbuf * DUO_ABUF + i * 1024}" for i in range(8)]
    reads += [f"ds_read_b128 v[{nbreg[j]}:{nbreg[j] + 3}], %[vb] offset:{nkx * DUO_BSLOT + j * 1024}" for j in range(4)]
    for r, ins in enumerate(reads):
        slots[1 + 2 * r].append(ins)
    # B of (P + 1, kx) into slot kx
    if kx == 0:
        slots[2].append(f"s_mov_b64 {SB}, {D_SBN1}")
    for j, k in enumerate((4, 20)):
        slots[k] += [f"s_add_u32 m0, %[sldsB], {kx * DUO_BSLOT + j * 1024}", "s_nop 0", f"global_load_lds_dwordx4 %[vob{j}], {SB}"]
    slots[22] += ["s_add_u32 s88, s88, %[s2cin]", "s_addc_u32 s89, s89, 0"]
    if kx == 0:
        for n, j in enumerate((3, 4)):
            slots[(12, 28)[n]] += duo_a_piece(j, 1 - per, D_SAN1, D_LO1, D_SP1)
    if kx == 2:
        for n, j in enumerate((0, 1, 2)):
            slots[(10, 16, 26)[n]] += duo_a_piece(j, per, D_SAN2, D_LO2, D_SP2)
    L = []
    k = 0
    for i in range(8):
        for j in range(4):
            acc = 4 * (4 * i + j)
            L.append(f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], v[{breg[j]}:{breg[j] + 3}], v[{areg[i]}:{areg[i] + 3}], " + ("0" if zero else f"a[{acc}:{acc + 3}]"))
            L += slots[k]
            k += 1
    if kx == 2:
        L += duo_shift(f"{pp}z" if zero else f"{pp}")
    n = (4, 2, 5)[kx]
    tag = f"{pp}z" if zero else f"{pp}"
    if pp == 0:
        L += [f"s_cmp_eq_u32 {SFLAG}, 0", f"s_cbranch_scc1 LDN{tag}_%=", "s_waitcnt vmcnt(63) lgkmcnt(0)", f"s_branch LDD{tag}_%=",
              f"LDN{tag}_%=:", f"s_waitcnt vmcnt({n}) lgkmcnt(0)", f"LDD{tag}_%=:", f"s_mov_b32 {SFLAG}, 0"]
    else:
        L.append(f"s_waitcnt vmcnt({n}) lgkmcnt(0)")
    L.append("s_barrier")
    return L


def tile_duo():
    # the multiplying workgroup outranks its CU partner (which may be in its VALU-heavy epilogue) at the instruction arbiter
    L = ["s_setprio 3", "s_mov_b32 s90, m0", f"s_mov_b64 {D_SAN1}, %[sAn1]", f"s_mov_b64 {D_SBN1}, %[sBn1]", f"s_mov_b32 {D_LO1}, %[slo]", f"s_mov_b32 {D_SP1}, %[ssp]",
         f"s_mov_b64 {D_SAN2}, %[sAn2]", f"s_mov_b64 {D_SBN2}, %[sBn2]", f"s_mov_b32 {D_LO2}, %[slo]", f"s_mov_b32 {D_SP2}, %[ssp]",
         f"s_mov_b32 {SKY}, %[sky2]", f"s_sub_u32 {SIN}, %[snper], 2", f"s_lshr_b32 {SCNT}, %[snper], 1", f"s_mov_b32 {SFLAG}, %[sflag]"]
    L += duo_kstep(0, True) + ["s_branch LDLOOP1_%=", "LDLOOP_%=:"]
    for pp in range(6):
        L += duo_kstep(pp)
        if pp == 0:
            L.append("LDLOOP1_%=:")
    L += [f"s_sub_u32 {SCNT}, {SCNT}, 1", f"s_cmp_lg_u32 {SCNT}, 0", "s_cbranch_scc1 LDLOOP_%=", "s_mov_b32 m0, s90", "s_setprio 0"]
    return L


def refill_duo():
    areg, breg = duo_frags(0)
    L = [f"ds_read_b128 v[{areg[i]}:{areg[i] + 3}], %[va0] offset:{i * 1024}" for i in range(8)]
    L += [f"ds_read_b128 v[{breg[j]}:{breg[j] + 3}], %[vb] offset:{j * 1024}" for j in range(4)]
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    return L


# ---- "tall": the lone-wave loop for 128-output-channel layers (conv_row_tall_kernel) ---------------------------------------------------
# Tile 512 pixels x 128 channels: wave w owns pixels 128 w .. + 127 and all 128 channels -- the 128 x 128 wave tile, accumulator layout,
# fragment registers and epilogue of conv_row_lw_kernel.  (256 x 128 with 128 x 64 wave tiles, the duo kernel above, reads 1.5 x the LDS
# fragment bytes per MFMA and sits on the LDS port: 0.39 of peak in its main loop.)  Two 256 x 256 row buffers would not fit for 512
# pixels, so K is staged in 32-channel periods like the duo kernel: two row buffers of 576 rows x 64 B (dil <= 32), B in FOUR 8-KiB slots.
# A period P = the three taps of one (32-channel block, kernel row) = 3 k-steps of 64 MFMAs.  Body k-step b = 0 .. 11 (four periods: the
# slot cycle of 4 and the period length of 3 close after 12; fragment set b & 1, row buffer (b / 3) & 1, B slot b & 3):
#   * the 16 fragment reads of k-step b + 1;
#   * B of k-step b + 4 into slot b & 3 (its fragments were read during b - 1): 2 pieces per wave, source = period P + 1 tap kx + 1
#     (kx = 0, 1) or period P + 2 tap 0 (kx = 2);
#   * at kx = 2 all 9 pieces of this wave of the row buffer of period P + 2 (the buffer of period P, last read during (P, 1));
#   * s_waitcnt vmcnt(13, 4, 13)[kx]: what k-step b + 2 reads has landed (B: issued at b - 2; row buffer: at the previous period's
#     tap 2), then the barrier.  After an epilogue the first two waits leave its stores outstanding (everything they are for was issued
#     before the epilogue, and the kernel waits vmcnt(0) before the first store), the third one is the first that needs them acknowledged.
# Tiles have an even number of periods (Cin % 64 == 0), so a tile starts at b = 0 or b = 6 (operand spar): fragment set 0, row buffer 0,
# B slot 0 or 2.
TALL_ABUF, TALL_BSLOT = 576 * 64, 128 * 64
TALL_BAL = os.environ.get("KDCC_GEN_TALL_BAL", "0") == "1"     # (what a third row buffer could buy at most: tools/tall_bal_probe.sh)
TALL_WAIT = (10, 10, 10) if TALL_BAL else (13, 4, 13)
TALL_FLAGWIN = int(os.environ.get("KDCC_GEN_TALL_FLAGWIN", "2"))   # experiment (timing only, wrong results above 2): k-steps after an epilogue whose waits ignore vmcnt


T_MASK = "s[80:81]"       # lane mask of the row-buffer piece being staged (the 512 x 128 loop has no period P + 1 masks in s80 / s81)


def tall_a_piece(j, buf, tag=""):
    """-> the piece's instructions as groups for consecutive MFMA gaps (one group when KDCC_GEN_LW_SPREAD=0)"""
    if SPREAD:
        return [[f"v_add_u32 {VT}, {16 * j}, %[vr0]", f"v_subrev_u32 {VT}, {D_LO2}, {VT}"],
                [f"v_cmp_gt_u32 {T_MASK}, {D_SP2}, {VT}", f"s_add_u32 m0, %[sldsA], {buf * TALL_ABUF + j * 1024}"],
                [f"s_mov_b64 exec, {T_MASK}", f"global_load_lds_dwordx4 %[voa{j}], {D_SAN2}", "s_mov_b64 exec, -1"],
                [f"s_not_b64 exec, {T_MASK}", f"s_cbranch_execz LTNZ{tag}{j}_%=", f"ds_write_b128 %[vz{buf}], %[vzero] offset:{j * 1024}",
                 f"LTNZ{tag}{j}_%=:", "s_mov_b64 exec, -1"]]
    return [[f"s_add_u32 m0, %[sldsA], {buf * TALL_ABUF + j * 1024}", f"v_add_u32 {VT}, {16 * j}, %[vr0]", f"v_subrev_u32 {VT}, {D_LO2}, {VT}",
             f"v_cmpx_gt_u32 vcc, {D_SP2}, {VT}", f"global_load_lds_dwordx4 %[voa{j}], {D_SAN2}", "s_not_b64 exec, exec",
             f"ds_write_b128 %[vz{buf}], %[vzero] offset:{j * 1024}", "s_mov_b64 exec, -1"]]


def tall_shift(tag):
    """period end: B's period P + 1 := P + 2; the (A, B) period P + 2 moves on: next kernel row / channel block of the tile, then the
    next tile's periods 0 and 1"""
    return [f"s_mov_b64 {D_SBN1}, {D_SBN2}",
            f"s_sub_u32 {SIN}, {SIN}, 1", f"s_cmp_gt_i32 {SIN}, 0", f"s_cbranch_scc0 LTT{tag}_%=",
            f"s_add_u32 {SKY}, {SKY}, 1", f"s_cmp_eq_u32 {SKY}, %[snky]",
            f"s_cselect_b32 {ST0}, %[sdAw], %[sdAs]", f"s_cselect_b32 {ST1}, %[sdBw], %[sdBs]", f"s_cselect_b32 {SKY}, 0, {SKY}",
            f"s_ashr_i32 {ST2}, {ST0}, 31", f"s_add_u32 s92, s92, {ST0}", f"s_addc_u32 s93, s93, {ST2}",
            f"s_ashr_i32 {ST2}, {ST1}, 31", f"s_add_u32 s94, s94, {ST1}", f"s_addc_u32 s95, s95, {ST2}",
            f"s_branch LTU{tag}_%=", f"LTT{tag}_%=:",
            f"s_mov_b32 {D_LO2}, %[sloT]", f"s_mov_b32 {D_SP2}, %[sspT]", f"s_cmp_eq_u32 {SIN}, 0", f"s_cbranch_scc0 LTV{tag}_%=",
            f"s_mov_b64 {D_SAN2}, %[sAnT0]", f"s_mov_b64 {D_SBN2}, %[sBnT0]", f"s_branch LTU{tag}_%=", f"LTV{tag}_%=:",
            f"s_mov_b64 {D_SAN2}, %[sAnT1]", f"s_mov_b64 {D_SBN2}, %[sBnT1]", f"LTU{tag}_%=:"]


def tall_kstep(b, zero=False):
    pb, kx, cur = b // 3, b % 3, b & 1
    per = pb & 1
    areg, breg = frag_regs(cur)
    nareg, nbreg = frag_regs(1 - cur)
    slots = [[] for _ in range(64)]
    nkx = (kx + 1) % 3
    nbuf = per if kx < 2 else 1 - per
    nslot = (b + 1) & 3
    reads = [f"ds_read_b128 v[{nareg[i]}:{nareg[i] + 3}], %[va{nkx}] offset:{nbuf * TALL_ABUF + i * 1024}" for i in range(8)]
    reads += [f"ds_read_b128 v[{nbreg[j]}:{nbreg[j] + 3}], %[vb] offset:{nslot * TALL_BSLOT + j * 1024}" for j in range(8)]
    for r, ins in enumerate(reads):
        slots[1 + 3 * r].append(ins)
    # B of k-step b + 4 into slot b & 3
    if kx == 0:
        slots[2] += ["s_add_u32 s88, s78, %[s2cin]", "s_addc_u32 s89, s79, 0"]        # (P + 1, tap 1)
    elif kx == 1:
        slots[2] += ["s_add_u32 s88, s88, %[s2cin]", "s_addc_u32 s89, s89, 0"]        # (P + 1, tap 2)
    else:
        slots[2].append(f"s_mov_b64 {SB}, {D_SBN2}")                                  # (P + 2, tap 0)
    for j, k in enumerate((5, 20)):
        if SPREAD:
            slots[k - 1].append(f"s_add_u32 m0, %[sldsB], {(b & 3) * TALL_BSLOT + j * 1024}")
            slots[k].append(f"global_load_lds_dwordx4 %[vob{j}], {SB}")
        else:
            slots[k] += [f"s_add_u32 m0, %[sldsB], {(b & 3) * TALL_BSLOT + j * 1024}", "s_nop 0", f"global_load_lds_dwordx4 %[vob{j}], {SB}"]
    if TALL_BAL:       # TIMING experiment (results wrong: the pieces overwrite a buffer that is being read): three row-buffer pieces in every k-step
        for n, j in enumerate(range(3 * kx, 3 * kx + 3)):
            for g, grp in enumerate(tall_a_piece(j, per, f"{b}z" if zero else f"{b}")):
                slots[(8, 29, 48)[n] + g] += grp
    elif kx == 2:
        for n, j in enumerate(range(9)):
            for g, grp in enumerate(tall_a_piece(j, per, f"{b}z" if zero else f"{b}")):
                slots[(8, 14, 23, 29, 35, 41, 48, 54, 60)[n] + g] += grp
    L = []
    k = 0
    for i in range(8):
        for j in range(8):
            acc = 4 * (8 * i + j)
            L.append(f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], v[{breg[j]}:{breg[j] + 3}], v[{areg[i]}:{areg[i] + 3}], " + ("0" if zero else f"a[{acc}:{acc + 3}]"))
            L += slots[k]
            k += 1
    tag = f"{b}z" if zero else f"{b}"
    if kx == 2:
        L += tall_shift(tag)
    n = TALL_WAIT[kx]
    if b % 6 < TALL_FLAGWIN:
        L += [f"s_cmp_eq_u32 {SFLAG}, 0", f"s_cbranch_scc1 LTN{tag}_%=", "s_waitcnt vmcnt(63) lgkmcnt(0)", f"s_branch LTD{tag}_%=",
              f"LTN{tag}_%=:", f"s_waitcnt vmcnt({n}) lgkmcnt(0)", f"LTD{tag}_%=:"]
        if b % 6 == TALL_FLAGWIN - 1:
            L.append(f"s_mov_b32 {SFLAG}, 0")
    else:
        L.append(f"s_waitcnt vmcnt({n}) lgkmcnt(0)")
    L.append("s_barrier")
    return L


def tile_tall():
    L = ["s_mov_b32 s90, m0", f"s_mov_b64 {D_SBN1}, %[sBn1]", f"s_mov_b64 {D_SAN2}, %[sAn2]", f"s_mov_b64 {D_SBN2}, %[sBn2]",
         f"s_mov_b32 {D_LO2}, %[slo]", f"s_mov_b32 {D_SP2}, %[ssp]", f"s_mov_b32 {SKY}, %[sky2]", f"s_sub_u32 {SIN}, %[snper], 2",
         f"s_lshr_b32 {SCNT}, %[snper], 1", f"s_mov_b32 {SFLAG}, %[sflag]", "s_cmp_eq_u32 %[spar], 0", "s_cbranch_scc0 LTODDZ_%="]
    L += tall_kstep(0, True) + ["s_branch LTA1_%=", "LTODDZ_%=:"] + tall_kstep(6, True) + ["s_branch LTB1_%=", "LTLOOP_%=:"]
    for b in range(12):
        L += tall_kstep(b)
        if b == 0:
            L.append("LTA1_%=:")
        if b == 6:
            L.append("LTB1_%=:")
        if b == 5:
            L += [f"s_sub_u32 {SCNT}, {SCNT}, 1", f"s_cmp_eq_u32 {SCNT}, 0", "s_cbranch_scc1 LTEND_%="]
    L += [f"s_sub_u32 {SCNT}, {SCNT}, 1", f"s_cmp_lg_u32 {SCNT}, 0", "s_cbranch_scc1 LTLOOP_%=", "LTEND_%=:", "s_mov_b32 m0, s90"]
    return L


def refill_tall():
    areg, breg = frag_regs(0)
    L = [f"ds_read_b128 v[{areg[i]}:{areg[i] + 3}], %[va0] offset:{i * 1024}" for i in range(8)]
    L += [f"ds_read_b128 v[{breg[j]}:{breg[j] + 3}], %[vbf] offset:{j * 1024}" for j in range(8)]
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    return L


def cstr(lines):
    return " \\\n".join('    "' + l + '\\n\\t"' for l in lines)


def main():
    o = ["// GENERATED by tools/gen_conv_lw.py -- do not edit (python tools/gen_conv_lw.py rewrites it).",
         "// The hand-scheduled main loop of conv_row_lw_kernel; see the generator for the schedule.", ""]
    o += ["#define LW_TILE_ASM \\", cstr(tile()), ""]
    global BROKEN
    BROKEN = True
    o += ["#ifdef KDCC_TUNING", "#define LW_TILE_BROKEN_ASM \\", cstr(tile()), "#endif", ""]
    BROKEN = False
    o += ["#define LW_REFILL_ASM \\", cstr(refill()), ""]
    o += ["#define DUO_TILE_ASM \\", cstr(tile_duo()), ""]
    o += ["#define DUO_REFILL_ASM \\", cstr(refill_duo()), ""]
    for i in range(8):
        o += [f"#define DUO_READ_ACC_{i}_ASM \\", cstr([f"v_accvgpr_read_b32 %{n}, a{16 * i + n}" for n in range(16)]), ""]
    o += ["#define DUO_ZERO_ACC_ASM \\", cstr([f"v_accvgpr_write_b32 a{n}, 0" for n in range(128)]), ""]
    o += ["#define DUO_CLOBBER_FRAG " + ", ".join(f'"v{n}"' for n in range(31, 128)),
          "#define DUO_CLOBBER_S " + ", ".join(f'"{x}"' for x in DUO_CLOBBER_S), ""]
    o += ["#define TALL_TILE_ASM \\", cstr(tile_tall()), ""]
    o += ["#define TALL_REFILL_ASM \\", cstr(refill_tall()), ""]
    o += ["#define LW1_TILE_ASM \\", cstr(tile_1x1()), ""]
    o += ["#define LW1_REFILL_ASM \\", cstr(refill_1x1()), ""]
    for i in range(8):
        for jg in range(2):
            o += [f"#define LW_READ_ACC_{i}_{jg}_ASM \\", cstr(read_acc(i, jg)), ""]
    o += ["#define LW_ZERO_ACC_ASM \\", cstr([f"v_accvgpr_write_b32 a{n}, 0" for n in range(256)]), ""]
    o += ["#define LW_CLOBBER_ACC " + ", ".join(f'"a{n}"' for n in range(256)),
          "#define LW_CLOBBER_FRAG " + ", ".join(f'"v{n}"' for n in range(96 if BREG else 127, 256)),
          "#define LW_CLOBBER_S " + ", ".join(f'"{s}"' for s in CLOBBER_S), ""]
    text = "\n".join(o)
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(0 if open(OUT).read() == text else 1)
    with open(OUT, "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
