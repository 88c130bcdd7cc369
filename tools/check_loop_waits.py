#!/usr/bin/env python3
"""Guard for the hand-scheduled conv main loops: hipcc's wait-count insertion must not put an s_waitcnt vmcnt(..) of its own
inside them (it does when it believes a global load issued by the epilogue may still be pending at the loop header and the
loop overwrites that load's destination registers -- the wait then drains the LDS-DMA every stage; 1x1 layers ran 1.5x
slower when this happened).  Compiles csrc/conv_igemm.hip to assembly and lists every s_waitcnt vmcnt outside inline-asm
blocks that sits in a depth-2 loop of a persistent kernel.  usage: python tools/check_loop_waits.py  (exit 1 if any)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "knowledge-distillation-by-replacing-cheap-conv_amd", "csrc", "conv_igemm.hip")
out = os.path.join(tempfile.gettempdir(), "conv_igemm_check.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", out],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
starts = [(i, l) for i, l in enumerate(lines) if re.match(r"^_ZN\S*:", l)]
def production(name):
    """the instantiations kd_conv2d_fwd dispatches by default: ping-pong, no debug stamps"""
    m = re.search(r"conv_row_persist_kernel.*EELi\dELb(\d)ELb(\d)E", name)
    if m:
        return m.group(1) == "0" and m.group(2) == "1"
    m = re.search(r"conv_igemm_persist_kernel.*EELi\dELb(\d)E(?:Lb(\d)E)?", name)
    return bool(m) and m.group(1) == "1" and (m.group(2) or "0") == "0"


bad = 0
for idx, (i, l) in enumerate(starts):
    if "persist_kernel" not in l and "pp128_kernel" not in l:
        continue
    prod = production(l) or "pp128_kernel" in l
    end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
    depth2, inasm, cur = False, False, ""
    for k in range(i, end):
        t = lines[k].strip()
        if t.startswith(";;#ASMSTART"): inasm = True
        if t.startswith(";;#ASMEND"): inasm = False
        m = re.match(r"(\.LBB\d+_\d+):(.*)", t)
        if m:
            depth2, cur = "Depth=2" in m.group(2), m.group(1)
        if depth2 and not inasm and t.startswith("s_waitcnt") and "vmcnt" in t:
            print(f"{'PRODUCTION' if prod else 'a/b-only  '} {l[:110]}  {cur}: {t}")
            bad += 1 if prod else 0
print("compiler-inserted vmcnt waits inside the main loops of the default kernels:", bad)
sys.exit(1 if bad else 0)
