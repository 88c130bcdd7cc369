#!/usr/bin/env python3
"""Static check of the hand-scheduled conv loops (tools/gen_conv_lw.py: LW_TILE_ASM of conv_row_lw_kernel, TALL_TILE_ASM of
conv_row_tall_kernel) on the CPU: the generated instruction stream of one wave is INTERPRETED for a few consecutive tiles -- scalar
registers, branches, M0, the in-order vmcnt counter, barriers -- with concrete operand values, and every LDS access is checked:

  * source: each fragment read of k-step h + 1 finds, in the row buffer / B slot it addresses, pieces that were fetched from the
    address the convolution needs there (period = (channel block, kernel row), tap, k-half) -- the staging iterator's arithmetic,
    the running B pointer and the hand-over to the next tile are all exercised;
  * read after write: every LDS-DMA piece a read depends on was covered by an `s_waitcnt vmcnt(N)` that it no longer falls under
    (vmcnt retires in order: after vmcnt(N) all but the youngest N operations have landed) AND a barrier after that wait, before
    the read (the other waves' pieces are published by the same barrier: all waves run this program);
  * write after read: no LDS-DMA piece or zero-fill is issued into a buffer / slot before a barrier that follows the last read of
    what it held;
  * every vmcnt operand fits the 6-bit field.

The epilogue between two tiles is modelled as the kernels run it: wait for everything staged ahead (vmcnt(0)), then `stores`
output stores that nobody waits for.  usage: check_lw_schedule.py  (exit 1 on a finding; tests/test_abi.py runs it)."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_conv_lw as G   # noqa: E402


class Finding(Exception):
    pass


def pair(tok):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    return int(m.group(1)) if m else None


class Wave:
    """one wave's view: scalar state, the in-order VMEM queue, barriers, LDS regions"""

    def __init__(self, ops, region_of_lds, name):
        self.s = {}            # SGPR number -> value
        self.scc = 0
        self.m0 = 0
        self.ops = ops         # operand name -> value (64-bit values as ints)
        self.vm = []           # issued VMEM operations, oldest first: dict(kind, region, tag, landed)
        self.nbar = 0          # barriers executed
        self.region_of_lds = region_of_lds
        self.reg = {}          # region -> dict(pieces: {piece: (tag, vm entry or None)}, last_read_bar)
        self.name = name
        self.findings = []
        self.mfma = 0
        self.reads = []        # (region, operand, offset) of the current k-step (consumed by the driver)

    # ---- operand access
    def val(self, tok):
        tok = tok.strip()
        if tok.startswith("%["):
            return self.ops[tok[2:-1]]
        p = pair(tok)
        if p is not None:
            return self.s.get(p, 0) | (self.s.get(p + 1, 0) << 32)
        if re.fullmatch(r"s\d+", tok):
            return self.s.get(int(tok[1:]), 0)
        if tok == "m0":
            return self.m0
        if tok == "exec":
            return 0
        return int(tok, 0)

    def put(self, tok, v):
        tok = tok.strip()
        p = pair(tok)
        if p is not None:
            self.s[p] = v & 0xffffffff
            self.s[p + 1] = (v >> 32) & 0xffffffff
        elif tok == "m0":
            self.m0 = v & 0xffffffff
        elif tok.startswith("%["):       # "+s" operand (sBp)
            self.ops[tok[2:-1]] = v
        elif re.fullmatch(r"s\d+", tok):
            self.s[int(tok[1:])] = v & 0xffffffff
        elif tok in ("exec", "vcc"):
            pass
        else:
            raise Finding(f"write to {tok}")

    def region(self, lds_addr):
        return self.region_of_lds(lds_addr)

    def note(self, msg):
        self.findings.append(f"{self.name}: {msg}")

    # ---- memory model
    def dma(self, lds_addr, tag):
        r, piece = self.region(lds_addr)
        st = self.reg.setdefault(r, {"pieces": {}, "last_read_bar": -1})
        if st["last_read_bar"] >= self.nbar:
            self.note(f"LDS-DMA into {r} piece {piece} in the barrier interval of a read of its previous content (interval {self.nbar})")
        e = {"kind": "load", "region": r, "landed": False}
        self.vm.append(e)
        st["pieces"][piece] = {"tag": tag, "vm": e, "wait_bar": None}

    def store(self):
        self.vm.append({"kind": "store", "landed": False})

    def waitcnt(self, n):
        if n > 63:
            self.note(f"vmcnt({n}) does not fit the field")
        pending = [e for e in self.vm if not e["landed"]]
        done = pending[:max(0, len(pending) - n)]
        for e in done:
            e["landed"] = True
            e["wait_bar"] = self.nbar      # published by the next barrier
        self.vm = [e for e in self.vm if not e["landed"]][-80:]

    def read(self, lds_addr, want_tag, what):
        r, _ = self.region(lds_addr)
        st = self.reg.get(r)
        if st is None or not st["pieces"]:
            self.note(f"{what}: read of {r}, which nothing was staged into")
            return
        for piece, pc in st["pieces"].items():
            if pc["tag"] != want_tag:
                self.note(f"{what}: {r} piece {piece} holds data from {pc['tag']:#x}, the read needs {want_tag:#x}")
                break
            e = pc["vm"]
            if e is not None:
                if not e["landed"]:
                    self.note(f"{what}: {r} piece {piece} is read while its DMA may still be in flight (no vmcnt wait covers it)")
                    break
                if e["wait_bar"] >= self.nbar:
                    self.note(f"{what}: {r} piece {piece} landed, but no barrier lies between the wait and the read")
                    break
        st["last_read_bar"] = self.nbar


def run(lines, w, on_read, max_steps=400000):
    """interpret one asm statement (list of instruction strings with %[operand] placeholders and local labels)"""
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    pc = 0
    steps = 0
    while pc < len(lines):
        steps += 1
        if steps > max_steps:
            raise Finding("the statement does not terminate")
        ins = lines[pc]
        pc += 1
        if ins.endswith(":"):
            continue
        op, _, rest = ins.partition(" ")
        a = [x.strip() for x in rest.split(",")] if rest else []
        if op == "v_mfma_f32_16x16x32_bf16":
            w.mfma += 1
        elif op == "ds_read_b128":
            addr_op, off = a[1].split(" offset:")
            on_read(w, addr_op.strip()[2:-1], int(off))
        elif op == "global_load_lds_dwordx4":
            w.dma(w.m0, w.val(a[1]))
        elif op == "ds_write_b128":
            pass      # zero fill under the complementary mask of the piece just issued: same region, same interval
        elif op in ("v_add_u32", "v_subrev_u32", "v_cmpx_gt_u32", "v_cmp_gt_u32", "s_nop", "s_setprio", "s_not_b64", "s_cbranch_execz"):
            pass      # (lane masks are not modelled: a piece counts as issued whatever its EXEC, its zero fill as never skipped)
        elif op == "s_mov_b32" or op == "s_mov_b64":
            w.put(a[0], w.val(a[1]))
        elif op == "s_add_u32":
            v = (w.val(a[1]) & 0xffffffff) + (w.val(a[2]) & 0xffffffff)
            w.scc = v >> 32
            w.put(a[0], v & 0xffffffff)
        elif op == "s_addc_u32":
            v = (w.val(a[1]) & 0xffffffff) + (w.val(a[2]) & 0xffffffff) + w.scc
            w.scc = v >> 32
            w.put(a[0], v & 0xffffffff)
        elif op == "s_sub_u32":
            v = (w.val(a[1]) & 0xffffffff) - (w.val(a[2]) & 0xffffffff)
            w.scc = 1 if v < 0 else 0
            w.put(a[0], v & 0xffffffff)
        elif op == "s_lshr_b32":
            w.put(a[0], (w.val(a[1]) & 0xffffffff) >> w.val(a[2]))
        elif op == "s_ashr_i32":
            x = w.val(a[1]) & 0xffffffff
            x = x - (1 << 32) if x >> 31 else x
            w.put(a[0], (x >> w.val(a[2])) & 0xffffffff)
        elif op in ("s_cselect_b32", "s_cselect_b64"):
            w.put(a[0], w.val(a[1]) if w.scc else w.val(a[2]))
        elif op in ("s_cmp_eq_u32", "s_cmp_lg_u32", "s_cmp_gt_i32"):
            x, y = w.val(a[0]) & 0xffffffff, w.val(a[1]) & 0xffffffff
            if op == "s_cmp_gt_i32":
                sx = x - (1 << 32) if x >> 31 else x
                sy = y - (1 << 32) if y >> 31 else y
                w.scc = int(sx > sy)
            else:
                w.scc = int((x == y) == (op == "s_cmp_eq_u32"))
        elif op in ("s_cbranch_scc0", "s_cbranch_scc1"):
            if w.scc == int(op[-1]):
                pc = labels[a[0]]
        elif op == "s_branch":
            pc = labels[a[0]]
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                w.waitcnt(int(m.group(1)))
        elif op == "s_barrier":
            w.nbar += 1
        else:
            raise Finding(f"instruction the checker does not model: {ins}")


# ---- the two kernels -----------------------------------------------------------------------------------------------------------------
LDS_A, LDS_B = 0x10000, 0x80000       # LDS byte addresses of this wave's first row-buffer piece / first B piece (any disjoint values)


def check_lw(cin, nkys, wave=1, stores=32):
    """conv_row_lw_kernel: tiles with nkys[t] kernel rows inside the image, Cin channels"""
    nkc = cin // 64
    dWl2, cin2 = 0x40000, 2 * cin

    def tile_bases(t):
        return 0x100000000 + t * 0x1000000, 0x7000000 + (t % 3) * 0x100000      # abase (row ho), bbase (N tile)

    def a_of(t, cb, kyi, kylo):
        return tile_bases(t)[0] + (kylo + kyi - 1) * dWl2 + cb * 128

    def b_of(t, cb, kyi, kylo):
        return tile_bases(t)[1] + 2 * ((kylo + kyi) * 3 * cin + cb * 64)

    def region_of(lds):
        if LDS_A <= lds < LDS_A + 2 * G.ABUF:
            o = lds - LDS_A
            return f"row buffer {o // G.ABUF}", (o % G.ABUF) // 1024
        o = lds - LDS_B
        if not 0 <= o < 4 * G.BSLOT:
            raise Finding(f"LDS-DMA destination {lds:#x} outside the buffers")
        return f"B slot {o // G.BSLOT}", (o % G.BSLOT) // 1024

    w = Wave({}, region_of, f"conv_row_lw_kernel Cin {cin} kernel rows {nkys}")
    kylo = [1 if n == 2 and t % 2 == 0 else 0 for t, n in enumerate(nkys)]     # (two rows: the top or the bottom one is missing)
    # prologue: row buffer 0 <- period 0; B slots 0..3 <- k-steps 0..3 of period 0
    for j in range(10):
        w.reg.setdefault("row buffer 0", {"pieces": {}, "last_read_bar": -1})["pieces"][j] = {"tag": a_of(0, 0, 0, kylo[0]), "vm": None}
    for q in range(4):
        for j in range(4):
            w.reg.setdefault(f"B slot {q}", {"pieces": {}, "last_read_bar": -1})["pieces"][j] = {"tag": b_of(0, 0, 0, kylo[0]) + (q >> 1) * cin2 + (q & 1) * 64, "vm": None}
    sBp = b_of(0, 0, 0, kylo[0]) + 2 * cin2
    par, flag = 0, 0
    lines = G.tile()
    for t, nky in enumerate(nkys):
        nper = nkc * nky
        nt = t + 1 if t + 1 < len(nkys) else t
        h = [0]          # k-steps of this tile computed so far (the reads of k-step h feed k-step h + 1)

        def expect(k, t=t, nky=nky):
            """sources of k-step k of this tile (k may run into the next tile)"""
            tt, kk, nk, kl = t, k, nky, kylo[t]
            if kk >= 6 * nkc * nky:
                kk -= 6 * nkc * nky
                tt, nk, kl = nt, nkys[nt], kylo[nt]
            q, p = divmod(kk, 6)
            cb, kyi = divmod(q, nk)
            return a_of(tt, cb, kyi, kl), b_of(tt, cb, kyi, kl) + (p >> 1) * cin2 + (p & 1) * 64, q

        def on_read(w, operand, off, t=t):
            k = h[0] + 1
            a_src, b_src, q = expect(k)
            if operand.startswith("va"):
                buf = off // G.ABUF
                w.read(LDS_A + buf * G.ABUF, a_src, f"tile {t} k-step {k} A fragments")
            else:
                w.read(LDS_B + (off // G.BSLOT) * G.BSLOT, b_src, f"tile {t} k-step {k} B fragments")

        w.ops = {"sBp": sBp, "sAn": a_of(t, 0, 1, kylo[t]) if nky > 1 else a_of(t, 1, 0, kylo[t]), "sBn": b_of(t, 0, 1, kylo[t]) if nky > 1 else b_of(t, 1, 0, kylo[t]),
                 "sAnT": a_of(nt, 0, 0, kylo[nt]), "sBnT": b_of(nt, 0, 0, kylo[nt]), "slo": 0, "ssp": 320, "sloT": 0, "sspT": 320,
                 "sdAs": dWl2, "sdAw": (128 - (nky - 1) * dWl2) & 0xffffffff, "sdBs": 6 * cin, "sdBw": (128 - (nky - 1) * 6 * cin) & 0xffffffff,
                 "snky": nky, "snper": nper, "s2c": cin2 - 64, "sflag": flag, "spar": par, "sldsA": LDS_A, "sldsB": LDS_B}
        # count k-steps through the MFMA counter: 64 per k-step
        base_mfma = w.mfma

        def on_read_counted(w, operand, off):
            h[0] = (w.mfma - base_mfma - 1) // 64
            on_read(w, operand, off)

        run(lines, w, on_read_counted)
        if w.mfma - base_mfma != 64 * 6 * nper:
            w.note(f"tile {t}: {w.mfma - base_mfma} MFMAs for {6 * nper} k-steps")
        sBp = w.ops["sBp"]
        par = (par + nper) & 1
        # refill: the next tile's first fragments, then the epilogue
        if t + 1 < len(nkys):
            a_src, b_src, _ = expect(6 * nper)
            w.read(LDS_A + par * G.ABUF, a_src, f"tile {t + 1} refill A")
            w.read(LDS_B + par * 2 * G.BSLOT, b_src, f"tile {t + 1} refill B")
            w.nbar += 1
            w.waitcnt(0)
            for _ in range(stores):
                w.store()
            flag = 1
    return w.findings


def check_tall(cin, nkys, stores=32):
    nkc = cin // 32
    dWl2, cin2 = 0x40000, 2 * cin

    def bases(t):
        return 0x100000000 + t * 0x1000000, 0x7000000

    def a_of(t, q, nky, kylo):
        return bases(t)[0] + (kylo + q % nky - 1) * dWl2 + (q // nky) * 64

    def b_of(t, q, nky, kylo):
        return bases(t)[1] + 2 * ((kylo + q % nky) * 3 * cin + (q // nky) * 32)

    def region_of(lds):
        if LDS_A <= lds < LDS_A + 2 * G.TALL_ABUF:
            o = lds - LDS_A
            return f"row buffer {o // G.TALL_ABUF}", (o % G.TALL_ABUF) // 1024
        o = lds - LDS_B
        if not 0 <= o < 4 * G.TALL_BSLOT:
            raise Finding(f"LDS-DMA destination {lds:#x} outside the buffers")
        return f"B slot {o // G.TALL_BSLOT}", (o % G.TALL_BSLOT) // 1024

    w = Wave({}, region_of, f"conv_row_tall_kernel Cin {cin} kernel rows {nkys}")
    kylo = [1 if n == 2 and t % 2 == 0 else 0 for t, n in enumerate(nkys)]
    for b in range(2):
        for j in range(9):
            w.reg.setdefault(f"row buffer {b}", {"pieces": {}, "last_read_bar": -1})["pieces"][j] = {"tag": a_of(0, b, nkys[0], kylo[0]), "vm": None}
    for q in range(4):
        for j in range(2):
            src = b_of(0, 0, nkys[0], kylo[0]) + q * cin2 if q < 3 else b_of(0, 1, nkys[0], kylo[0])
            w.reg.setdefault(f"B slot {q}", {"pieces": {}, "last_read_bar": -1})["pieces"][j] = {"tag": src, "vm": None}
    par, flag = 0, 0
    lines = G.tile_tall()
    for t, nky in enumerate(nkys):
        nper = nkc * nky
        nt = t + 1 if t + 1 < len(nkys) else t

        def expect(k, t=t, nky=nky):
            tt, kk, nk, kl = t, k, nky, kylo[t]
            if kk >= 3 * nkc * nky:
                kk -= 3 * nkc * nky
                tt, nk, kl = nt, nkys[nt], kylo[nt]
            q, kx = divmod(kk, 3)
            return a_of(tt, q, nk, kl), b_of(tt, q, nk, kl) + kx * cin2

        base_mfma = w.mfma
        phase = par * 6

        def on_read(w, operand, off, t=t):
            k = (w.mfma - base_mfma - 1) // 64 + 1
            a_src, b_src = expect(k)
            if operand.startswith("va"):
                w.read(LDS_A + (off // G.TALL_ABUF) * G.TALL_ABUF, a_src, f"tile {t} k-step {k} A fragments")
            else:
                w.read(LDS_B + (off // G.TALL_BSLOT) * G.TALL_BSLOT, b_src, f"tile {t} k-step {k} B fragments")

        w.ops = {"sBn1": b_of(t, 1, nky, kylo[t]), "sAn2": a_of(t, 2, nky, kylo[t]), "sBn2": b_of(t, 2, nky, kylo[t]),
                 "sAnT0": a_of(nt, 0, nkys[nt], kylo[nt]), "sBnT0": b_of(nt, 0, nkys[nt], kylo[nt]), "sAnT1": a_of(nt, 1, nkys[nt], kylo[nt]),
                 "sBnT1": b_of(nt, 1, nkys[nt], kylo[nt]), "slo": 0, "ssp": 576, "sloT": 0, "sspT": 576, "sdAs": dWl2,
                 "sdAw": (64 - (nky - 1) * dWl2) & 0xffffffff, "sdBs": 6 * cin, "sdBw": (64 - (nky - 1) * 6 * cin) & 0xffffffff, "snky": nky,
                 "sky2": 2 % nky, "snper": nper, "s2cin": cin2, "sflag": flag, "spar": par, "sldsA": LDS_A, "sldsB": LDS_B}
        run(lines, w, on_read)
        if w.mfma - base_mfma != 64 * 3 * nper:
            w.note(f"tile {t}: {w.mfma - base_mfma} MFMAs for {3 * nper} k-steps")
        par = (par + (nper >> 1)) & 1
        if t + 1 < len(nkys):
            a_src, b_src = expect(3 * nper)
            w.read(LDS_A, a_src, f"tile {t + 1} refill A")
            w.read(LDS_B + par * 2 * G.TALL_BSLOT, b_src, f"tile {t + 1} refill B")
            w.nbar += 1
            w.waitcnt(0)
            for _ in range(stores):
                w.store()
            flag = 1
        del phase
    return w.findings


def main():
    findings = []
    for cin, nkys in ((64, [3, 2, 3]), (128, [3, 3, 2, 2, 3]), (320, [3, 2, 3, 3]), (512, [2, 3, 3])):
        for stores in (32, 64, 96):
            findings += check_lw(cin, nkys, stores=stores)
    for cin, nkys in ((64, [3, 2, 3, 3]), (128, [3, 2, 2, 3]), (192, [3, 3, 2, 3])):
        for stores in (32, 64):
            findings += check_tall(cin, nkys, stores=stores)
    for f in findings[:30]:
        print(f)
    print(f"check_lw_schedule: {len(findings)} findings")
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
