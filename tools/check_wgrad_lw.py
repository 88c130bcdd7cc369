#!/usr/bin/env python3
"""Checks of the lone-wave dense weight-gradient kernels (conv_wgrad_lw_kernel, csrc/pw_wgrad.hip + the generated csrc/wgrad_lw_body.inc; and, with
interpret_pw below, conv_wgrad_pw_lw_kernel + csrc/wgrad_pw_lw_body.inc).

1. Schedule interpreter (no hipcc needed).  The generated stream is walked prologue -> the four unrolled stage bodies, twice -> drain as the hardware retires
   it, once with the zero fill of every row-buffer piece issued and once with all of them skipped: LDS operations and vector-memory operations are two IN-ORDER queues, `s_waitcnt lgkmcnt(n) / vmcnt(n)` retires all but the n
   youngest of a queue.  Checked:
   * every MFMA's fragment registers have landed (no transposing read into them is still in the LDS queue), and no read overwrites a
     fragment register an older read is still in flight for;
   * the LDS-DMA pieces are numbered by the stage they belong to (prologue: stages 0, 1, 2 whole and the first five pieces of stage 3;
     stage body `it`: the last four pieces of stage it + 3 in k-step 0, the first five of stage it + 4 in k-step 1).  At the barrier
     of stage body `it` no piece of a stage <= it + 1 may be in flight (k-step 1 reads stage it + 1 right behind it), at the
     prologue's barrier none of stage 0; no LDS operation may be in flight at a barrier;
   * the reads of k-step 0's fragments (ring offset 0 / 1024: stage it + 1) come behind the barrier, those of k-step 1's (offset 8192 /
     9216: stage it) in front of it; the pieces that overwrite stage it's slot (stage it + 4) come behind it;
   * every fragment read and every piece addresses the ring slot (stage mod 4) of the stage it belongs to;
   * one barrier and 96 MFMAs per stage body, nothing in flight at the end.
2. ISA audit (needs hipcc): in the compiled kernel no compiler-generated instruction touches an accumulation register, the 4 x 96 MFMAs
   sit in ONE inline-asm statement, no scratch.

usage: check_wgrad_lw.py [--no-isa]; exit 1 on a finding.  tests/test_abi.py runs it and holds it to account with mutated schedules."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "knowledge-distillation-by-replacing-cheap-conv_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, os.path.join(ROOT, "tools"))

VREG = re.compile(r"\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = []
    for m in VREG.finditer(text):
        out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


STAGE = 40960


def interpret(lines, passes=2):
    """both outcomes of the pieces' zero fill: issued (boundary stages), skipped (s_cbranch_execz taken: interior stages)"""
    skipped = [l for l in lines if not l.startswith("ds_write_b128")]
    return interpret_path(lines, passes) + [f"(zero fill skipped) {f}" for f in interpret_path(skipped, passes)]


def interpret_path(lines, passes):
    findings = []
    top = lines.index("WGL_LOOP_%=:")
    back = max(i for i, l in enumerate(lines) if l.startswith("s_cbranch_scc1 WGL_LOOP"))
    seq = [(l, -1) for l in lines[:top]]
    it = 0
    for p in range(passes):
        for l in lines[top + 1:back + 1]:
            seq.append((l, it))
            if l.startswith("s_cbranch_scc0 WGL_DONE") or l.startswith("s_cbranch_scc1 WGL_LOOP"):
                it += 1
    bodies = it
    seq += [(l, bodies) for l in lines[back + 1:]]
    ds, vm = [], []                 # ds: (dest regs, text); vm: stage of the piece
    n_piece = 0
    stats = {}
    m0 = None

    def stage_of(n):
        if n < 32:
            return n // 9
        k = n - 32
        it, r = divmod(k, 9)
        return it + 3 if r < 4 else it + 4

    barrier_seen = {}
    for ins, it in seq:
        op = ins.split(" ")[0]
        st = stats.setdefault(it, {"mfma": 0, "barrier": 0})
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                k = int(m.group(1))
                ds = ds[len(ds) - k:] if k else []
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                k = int(m.group(1))
                vm = vm[len(vm) - k:] if k else []
            continue
        if op == "s_add_u32" and ins.startswith("s_add_u32 m0, %[sldsw], "):
            m0 = int(ins.rsplit(" ", 1)[1])
            continue
        if op == "s_barrier":
            st["barrier"] += 1
            barrier_seen[it] = True
            if ds:
                findings.append(f"body {it}: s_barrier with LDS operations in flight: {ds[-1][1]}")
            limit = 0 if it < 0 else it + 1
            late = [s for s in vm if s <= limit]
            if late:
                findings.append(f"body {it}: {len(late)} LDS-DMA piece(s) of stage {min(late)} in flight at the barrier behind which stage {limit} is read")
            continue
        if op.startswith("v_mfma"):
            st["mfma"] += 1
            pend = {r: t for d, t in ds for r in d}
            for r in vregs(ins):
                if r in pend:
                    findings.append(f"body {it}: `{ins}` reads v{r} while `{pend[r]}` is in flight")
            continue
        if op.startswith("ds_read"):
            dst = vregs(ins.split(",")[0])
            pend = {r: t for d, t in ds for r in d}
            for r in dst:
                if r in pend:
                    findings.append(f"body {it}: `{ins}` overwrites v{r} while `{pend[r]}` is in flight")
            ds.append((dst, ins))
            off = int(re.search(r"offset:(\d+)", ins).group(1))
            ring = off // STAGE + (2 if re.search(r"%\[w[ab]", ins) else 0)
            ks = 1 if off % STAGE >= 8192 else 0
            if 0 <= it < bodies:
                want = it if ks else it + 1
                if ring != want % 4:
                    findings.append(f"body {it}: a read of stage {want} addresses ring slot {ring}: {ins}")
                if ks == 0 and not barrier_seen.get(it):
                    findings.append(f"body {it}: a read of stage {it + 1} in front of the barrier that publishes it: {ins}")
                if ks == 1 and barrier_seen.get(it):
                    findings.append(f"body {it}: a read of stage {it} behind the barrier that retires it: {ins}")
            continue
        if op.startswith("ds_write"):
            ds.append(([], ins))
            continue
        if op.startswith("global_load_lds"):
            s = stage_of(n_piece)
            n_piece += 1
            vm.append(s)
            if m0 is None or m0 // STAGE != s % 4:
                findings.append(f"body {it}: a piece of stage {s} goes to M0 = {m0} (ring slot {None if m0 is None else m0 // STAGE}): {ins}")
            m0 = None
            if 0 <= it < bodies and s == it + 4 and not barrier_seen.get(it):
                findings.append(f"body {it}: a piece of stage {s} overwrites stage {it}'s slot in front of the barrier that retires it: {ins}")
            continue
    for it, st in stats.items():
        if 0 <= it < bodies and (st["mfma"] != 96 or st["barrier"] != 1):
            findings.append(f"body {it}: {st['mfma']} MFMAs / {st['barrier']} barriers (96 / 1 expected)")
    if ds or vm:
        findings.append(f"operations still in flight at the end: {len(ds)} LDS, {len(vm)} vector memory")
    return findings


def interpret_pw(lines, passes=2):
    """the same walk for conv_wgrad_pw_lw_kernel's loop (tools/gen_wgrad_pw_lw.py): 32-pixel stages of 32 KiB, the prologue stages 0-2, body `it`
    the 8 pieces of stage it + 3 in its first half, ONE barrier in its middle (no piece of a stage <= it + 1 and no LDS operation in
    flight), behind it the 32 reads of stage it + 1 -- every one of them in ring slot (it + 1) mod 4 --, 64 MFMAs per body"""
    findings = []
    top = lines.index("WGP_LOOP_%=:")
    back = max(i for i, l in enumerate(lines) if l.startswith("s_cbranch_scc1 WGP_LOOP"))
    seq = [(l, -1) for l in lines[:top]]
    it = 0
    for p in range(passes):
        for l in lines[top + 1:back + 1]:
            seq.append((l, it))
            if l.startswith("s_cbranch_scc0 WGP_DONE") or l.startswith("s_cbranch_scc1 WGP_LOOP"):
                it += 1
    bodies = it
    seq += [(l, bodies) for l in lines[back + 1:]]
    ds, vm, n_piece, m0, stats, barrier_seen = [], [], 0, None, {}, {}
    for ins, it in seq:
        op = ins.split(" ")[0]
        st = stats.setdefault(it, {"mfma": 0, "barrier": 0})
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                k = int(m.group(1))
                ds = ds[len(ds) - k:] if k else []
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                k = int(m.group(1))
                vm = vm[len(vm) - k:] if k else []
        elif op == "s_add_u32" and ins.startswith("s_add_u32 m0, %[sldsw], "):
            m0 = int(ins.rsplit(" ", 1)[1])
        elif op == "s_barrier":
            st["barrier"] += 1
            barrier_seen[it] = True
            if ds:
                findings.append(f"body {it}: s_barrier with LDS operations in flight: {ds[-1][1]}")
            limit = 0 if it < 0 else it + 1
            late = [x for x in vm if x <= limit]
            if late:
                findings.append(f"body {it}: {len(late)} LDS-DMA piece(s) of stage {min(late)} in flight at the barrier behind which stage {limit} is read")
        elif op.startswith("v_mfma"):
            st["mfma"] += 1
            pend = {r: t for d, t in ds for r in d}
            for r in vregs(ins):
                if r in pend:
                    findings.append(f"body {it}: `{ins}` reads v{r} while `{pend[r]}` is in flight")
        elif op.startswith("ds_read"):
            dst = vregs(ins.split(",")[0])
            pend = {r: t for d, t in ds for r in d}
            for r in dst:
                if r in pend:
                    findings.append(f"body {it}: `{ins}` overwrites v{r} while `{pend[r]}` is in flight")
            ds.append((dst, ins))
            off = int(re.search(r"offset:(\d+)", ins).group(1))
            ring = off // 32768 + (2 if re.search(r"%\[w[ab]", ins) else 0)
            if it < bodies and ring != (it + 1) % 4:
                findings.append(f"body {it}: a read of stage {it + 1} addresses ring slot {ring}: {ins}")
            if it < bodies and not barrier_seen.get(it):
                findings.append(f"body {it}: a read of stage {it + 1} in front of the barrier that publishes it: {ins}")
        elif op.startswith("global_load_lds"):
            s_ = n_piece // 8 if n_piece < 24 else (n_piece - 24) // 8 + 3
            n_piece += 1
            vm.append(s_)
            if m0 is None or m0 // 32768 != s_ % 4:
                findings.append(f"body {it}: a piece of stage {s_} goes to M0 = {m0}: {ins}")
            m0 = None
    for it, st in stats.items():
        if 0 <= it < bodies and (st["mfma"] != 64 or st["barrier"] != 1):
            findings.append(f"body {it}: {st['mfma']} MFMAs / {st['barrier']} barriers (64 / 1 expected)")
    if ds or vm:
        findings.append(f"operations still in flight at the end: {len(ds)} LDS, {len(vm)} vector memory")
    return findings


def audit_isa(text, kernel="conv_wgrad_lw_kernel", n_mfma=384):
    findings = []
    m = re.search(r"^(_ZN\S*" + kernel + r"\S*):\s*;[^\n]*\n(.*?)\.Lfunc_end", text, flags=re.M | re.S)
    if not m:
        return [kernel + " not found in the assembly"]
    name, code = m.group(1), m.group(2)
    in_asm, per_stmt, cur = False, [], 0
    for l in code.split("\n"):
        if "#ASMSTART" in l:
            in_asm, cur = True, 0
            continue
        if "#ASMEND" in l:
            in_asm = False
            per_stmt.append(cur)
            continue
        s = l.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        if in_asm:
            cur += s.count("v_mfma")
        else:
            if "v_mfma" in s:
                findings.append(f"{name}: compiler-generated MFMA")
            if re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", s) or "accvgpr" in s:
                findings.append(f"{name}: compiler instruction touches an accumulation register: {s}")
            if "scratch_" in s:
                findings.append(f"{name}: scratch access: {s}")
    if sorted(per_stmt) != [0, 0, n_mfma]:
        findings.append(f"{name}: expected three inline-asm statements (zero, the four unrolled stage bodies with {n_mfma} MFMAs, store), found MFMA counts {per_stmt}")
    return findings


def main():
    import gen_wgrad_lw as G
    import gen_wgrad_pw_lw as GP
    findings = interpret(G.build()) + [f"(1x1 loop) {f}" for f in interpret_pw(GP.build())]
    n_isa = 0
    if "--no-isa" not in sys.argv and os.path.exists(HIPCC):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "pw_wgrad.s")
            subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-unused-value", "-S",
                                   "--cuda-device-only", os.path.join(CSRC, "pw_wgrad.hip"), "-o", out], stderr=subprocess.DEVNULL)
            findings += audit_isa(open(out).read()) + audit_isa(open(out).read(), "conv_wgrad_pw_lw_kernel", 256)
            n_isa = 1
    for f in findings[:40]:
        print(f)
    print(f"check_wgrad_lw: schedule interpreted, ISA audited: {bool(n_isa)}, {len(findings)} findings")
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
