#!/usr/bin/env python3
"""Audit of conv_lw.hip's device code (cdna_hip_programming.md 5.7 item 4).  The hand-written main loop keeps its accumulators
in a[0:255] across compiler-generated code (the epilogue reads them back with v_accvgpr_read in inline asm), so:
  * no compiler-generated instruction anywhere in the kernel may touch an accumulator register -- a spill into that range
    would be silent corruption;
  * every MFMA of a kernel sits in ONE inline-asm statement (all periods of a tile: the fragment registers v[128:255] never
    live across compiler code): 896 in the 3x3 kernel (two period bodies + two zero-C first k-steps), 320 in the 1x1 kernel;
  * no scratch (private segment) in the instantiation without epilogue operands.
usage: check_lw_asm.py [conv_lw.s]   (without an argument compiles csrc/conv_lw.hip to assembly first); exit 1 on a finding."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "knowledge-distillation-by-replacing-cheap-conv_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+(\S+)")
LABEL = re.compile(r"^(\.?[A-Za-z_][\w.$]*):")


def audit(text):
    findings = []
    bodies = re.split(r"^(_ZN\S*conv_(?:row_lw|pw_lw|row_duo|row_tall)_kernel\S*):\s*;[^\n]*\n", text, flags=re.M)
    n_kernels = 0
    for k in range(1, len(bodies), 2):
        name, code = bodies[k], bodies[k + 1].split(".Lfunc_end")[0]
        n_kernels += 1
        lines = code.split("\n")
        # instructions with their kind: ('asm', tag) for a whole inline-asm statement, ('ins', text), ('label', name)
        items, i = [], 0
        while i < len(lines):
            l = lines[i]
            if "#ASMSTART" in l:
                j = i + 1
                blk = []
                while "#ASMEND" not in lines[j]:
                    blk.append(lines[j])
                    j += 1
                body = "\n".join(blk)
                items.append(("asm", "", body))
                i = j + 1
                continue
            s = l.split(";")[0].rstrip()
            m = LABEL.match(s)
            if m:
                items.append(("label", m.group(1), ""))
            elif s.strip() and not s.strip().startswith("."):
                items.append(("ins", s, ""))
            i += 1
        # 1. accumulator registers outside inline asm
        for it in items:
            if it[0] == "ins" and (re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", it[1]) or "accvgpr" in it[1]):
                findings.append(f"{name}: compiler instruction touches an accumulator register: {it[1].strip()}")
        # 2. all MFMAs in one statement
        mf = [it[2].count("v_mfma") for it in items if it[0] == "asm" and "v_mfma" in it[2]]
        if "conv_row_lw_kernelILi16E" in name:
            # the classifier epilogue (conv_common.h lw_epilogue_cls16): 2 halves x 8 pixel tiles statements of four MFMAs (2 k-steps x 2 class
            # tiles) whose operands are all VGPRs -- none of them may name an accumulation register -- each closed by its own wait states
            extra = [it[2] for it in items if it[0] == "asm" and it[2].count("v_mfma") == 4]
            if len(extra) != 16 or any(re.search(r"\ba\[?\d", b) or "s_nop 15" not in b for b in extra):
                findings.append(f"{name}: expected 16 four-MFMA statements on vector registers (with their wait states) in the classifier epilogue, found {len(extra)}")
            mf = [n for n in mf if n != 4]
        want = [320] if "conv_pw_lw" in name else [224] if "conv_row_duo" in name else [896]      # 1x1: one pass over the four slots; 3x3: two period bodies; + the zero-C first k-step(s)
        if mf != want:
            findings.append(f"{name}: expected one inline-asm statement with {want[0]} MFMAs, found {mf}")
        if any(it[0] == "ins" and "v_mfma" in it[1] for it in items):
            findings.append(f"{name}: compiler-generated MFMA")
        if "ILi0E" in name and "conv_row_duo" not in name:
            m = re.search(re.escape(name) + r"\.private_seg_size, (\d+)", text)
            if m and int(m.group(1)) != 0:
                findings.append(f"{name}: scratch in the no-operand instantiation ({m.group(1)} B)")
    if n_kernels == 0:
        findings.append("no conv_row_lw_kernel / conv_pw_lw_kernel found in the assembly")
    return findings, n_kernels


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "conv_lw.s")
            subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-unused-value", "-S",
                                   "--cuda-device-only", os.path.join(CSRC, "conv_lw.hip"), "-o", out], stderr=subprocess.DEVNULL)
            text = open(out).read()
    findings, n = audit(text)
    for f in findings[:40]:
        print(f)
    print(f"check_lw_asm: {n} kernels, {len(findings)} findings")
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
