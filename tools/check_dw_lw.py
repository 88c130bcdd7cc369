#!/usr/bin/env python3
"""Checks of the lone-wave depthwise fan-out kernel (csrc/dwconv_lw.hip + the generated csrc/dw_lw_body.inc).

1. Schedule interpreter (no hipcc needed).  The generated stream is interpreted as the hardware retires it: LDS operations and
   vector-memory operations are two IN-ORDER queues, `s_waitcnt lgkmcnt(n) / vmcnt(n)` retires all but the n youngest of a queue,
   scalar loads retire only at lgkmcnt(0).  Walking prologue -> item body -> item body -> drain, every instruction's source
   registers are looked up: a register that is the destination of an operation still in a queue is a finding (a fragment read by an
   MFMA before its ds_read was waited for, a buffer_store of staging data still in flight, a transposing v_perm of tile data that has
   not landed, an item descriptor used before the s_load returned).  Also checked: each ds_read / buffer_load destination is not
   overwritten while an older operation into it is pending, every item body has exactly 4 barriers each preceded by a lgkmcnt(0),
   and 336 MFMAs per item.
2. ISA audit (needs hipcc): in the compiled kernel no compiler-generated instruction touches an accumulation register after the
   first operand fragment has been written there, every MFMA sits in the one generated statement, there is no scratch, and the
   statement's only vector input is v167 (the one register the statement does not clobber).

usage: check_dw_lw.py [--no-isa]; exit 1 on a finding.  tests/test_abi.py runs it and holds it to account with mutated schedules."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "knowledge-distillation-by-replacing-cheap-conv_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, os.path.join(ROOT, "tools"))

REG = re.compile(r"\b([vs])(\d+)\b|\b([vs])\[(\d+):(\d+)\]")


def regs(operand):
    out = []
    for m in REG.finditer(operand):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out += [(m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out


def split_ops(ins):
    op, _, rest = ins.partition(" ")
    rest = re.sub(r"\boffset\d?:\d+|\boffen\b|\bnt\b", "", rest)
    return op, [o.strip() for o in rest.split(",") if o.strip()]


def interpret(lines, passes=2):
    """lines: the generated instruction list (gen_dw_lw.build()).  Returns findings."""
    findings = []
    top = lines.index("DWLW_ITEM_%=:")
    back = max(i for i, l in enumerate(lines) if l.startswith("s_cbranch_scc0 DWLW_ITEM"))
    seq = [(l, "pro") for l in lines[:top]]
    for p in range(passes):
        seq += [(l, f"item{p}") for l in lines[top + 1:back + 1]]
    seq += [(l, "drain") for l in lines[back + 1:]]
    ds, vm = [], []          # pending: (set of dest regs, text, is_smem)
    stats = {}
    for n, (ins, where) in enumerate(seq):
        op, ops = split_ops(ins)
        st = stats.setdefault(where, {"mfma": 0, "barrier": 0})
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                k = int(m.group(1))
                n_ret = len(ds) if k == 0 else max(0, len(ds) - k)
                ds = [d for d in ds[:n_ret] if d[2] and k] + ds[n_ret:]       # (a scalar load only retires at lgkmcnt(0): it may overtake LDS operations)
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                k = int(m.group(1))
                vm = vm[max(0, len(vm) - k):] if k else []
            st["last_wait0"] = n if re.search(r"lgkmcnt\(0\)", ins) else st.get("last_wait0", -9)
            continue
        if op == "s_barrier":
            st["barrier"] += 1
            if ds:
                findings.append(f"{where}: s_barrier with LDS operations in flight: {ds[-1][1]}")
            continue
        if op.endswith(":") or op.startswith("s_cbranch") or op.startswith("s_branch"):
            continue
        if op.startswith("v_mfma"):
            st["mfma"] += 1
        # destination / source split
        if op.startswith("ds_write") or op.startswith("buffer_store"):
            dst, src = [], [r for o in ops for r in regs(o)]
        elif op.startswith("v_cmp") and ops and ops[0] == "vcc":
            dst, src = [], [r for o in ops[1:] for r in regs(o)]
        else:
            dst, src = (regs(ops[0]) if ops else []), [r for o in ops[1:] for r in regs(o)]
            if op.startswith("v_mfma") and not ins.rstrip().endswith(", 0"):
                pass     # (C operand is listed among the sources already)
        pend = {}
        for q in (ds, vm):
            for d, text, _ in q:
                for r in d:
                    pend[r] = text
        for r in src:
            if r in pend:
                findings.append(f"{where}: `{ins}` reads {r[0]}{r[1]} while `{pend[r]}` is in flight")
        for r in dst:
            if r in pend:
                findings.append(f"{where}: `{ins}` overwrites {r[0]}{r[1]} while `{pend[r]}` is in flight")
        if op.startswith("ds_read"):
            ds.append((set(dst), ins, False))
        elif op.startswith("ds_write"):
            ds.append((set(), ins, False))
        elif op.startswith("s_load"):
            ds.append((set(dst), ins, True))
        elif op.startswith("buffer_load"):
            vm.append((set(dst), ins, False))
        elif op.startswith("buffer_store"):
            vm.append((set(), ins, False))
    for where, st in stats.items():
        if where.startswith("item") and (st["mfma"] != 336 or st["barrier"] != 4):
            findings.append(f"{where}: {st['mfma']} MFMAs / {st['barrier']} barriers (336 / 4 expected)")
    if ds or vm:
        findings.append(f"operations still in flight at the end: {len(ds)} LDS, {len(vm)} vector memory")
    return findings


def audit_isa(text):
    findings = []
    m = re.search(r"^(_ZN\S*dw_lw_fan3_kernel\S*):\s*;[^\n]*\n(.*?)\.Lfunc_end", text, flags=re.M | re.S)
    if not m:
        return ["dw_lw_fan3_kernel not found in the assembly"]
    name, code = m.group(1), m.group(2)
    lines = code.split("\n")
    in_asm, seen_acc_write, mf_stmt, cur = False, False, [], 0
    vin = None
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm, cur = True, 0
            continue
        if "#ASMEND" in l:
            in_asm = False
            if cur:
                mf_stmt.append(cur)
            continue
        s = l.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        if in_asm:
            if "v_accvgpr_write" in s:
                seen_acc_write = True
            cur += s.count("v_mfma")
            mm = re.match(r"ds_read_b128 v\[152:155\], (v\d+)", s)
            if mm:
                vin = mm.group(1)
        else:
            if "v_mfma" in s:
                findings.append(f"{name}: compiler-generated MFMA")
            if seen_acc_write and (re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", s) or "accvgpr" in s):
                findings.append(f"{name}: compiler instruction touches an accumulation register after the operands were placed: {s}")
            if "scratch_" in s:
                findings.append(f"{name}: scratch access: {s}")
    if mf_stmt != [336]:
        findings.append(f"{name}: expected one inline-asm statement with 336 MFMAs, found {mf_stmt}")
    if vin != "v167":
        findings.append(f"{name}: the generated statement's vector input is {vin}, not v167 (the one register it does not clobber)")
    return findings


def main():
    import gen_dw_lw as G
    findings = interpret(G.build())
    n_isa = 0
    if "--no-isa" not in sys.argv and os.path.exists(HIPCC):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "dwconv_lw.s")
            subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-unused-value", "-S",
                                   "--cuda-device-only", os.path.join(CSRC, "dwconv_lw.hip"), "-o", out], stderr=subprocess.DEVNULL)
            findings += audit_isa(open(out).read())
            n_isa = 1
    for f in findings[:40]:
        print(f)
    print(f"check_dw_lw: schedule interpreted, ISA audited: {bool(n_isa)}, {len(findings)} findings")
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
