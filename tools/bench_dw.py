#!/usr/bin/env python3
"""Standalone microbenchmark of the depthwise kernels at the student's shapes (GPU box only)."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kdcc_amd
# KDCC_LIB=/path/to/other.so: A/B against another build of the library (same ABI); KDCC_LIB=tuning + KDCC_DW_DBG=bits: the
# phase ablations of the diagnostics build (kdcc_amd/_lib.py)
if os.environ.get("KDCC_LIB") == "tuning":
    kdcc_amd._lib.build_tuning()
from kdcc_amd import ops

def t(fn, it=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

NB = int(os.environ.get("KDCC_BENCH_BATCH", "2"))
for C in (512, 1024, 2048, 4096):
    H, W, k, p, d = 128, 256, 9, 20, 5
    x = torch.randn(NB, H, W, C, device="cuda").bfloat16()
    g = torch.randn(NB, H, W, C, device="cuda").bfloat16()
    w = torch.randn(C, 1, k, k, device="cuda") / k
    wt = ops.pack_dw_weight(w)
    y = torch.empty_like(x)
    dw = torch.empty_like(w)
    ws = torch.empty(ops._lib.lib().kd_dwconv_wgrad_workspace(__import__("ctypes").byref(ops._dw_desc(x, k, p, d))), dtype=torch.uint8, device="cuda")
    fl = 2.0 * k * k * C * H * W * NB
    by = 2.0 * 2 * C * H * W * NB
    ms = t(lambda: ops.dwconv(x, wt, k, p, d, out=y))
    print(f"dw fwd   C={C:5d}: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s  {by/ms/1e6:7.1f} GB/s (algorithmic in+out)")
    ms = t(lambda: ops.dwconv(x, wt, k, p, d, out=y, mask=g, mask_scale=torch.ones(C, device='cuda'), res_post=g))
    print(f"dw dgrad(mask+res) C={C:5d}: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s")
    if C == 4096:   # the ASPP input gradient: three branches summed in one launch vs chained through res_post
        g2, g3 = torch.randn_like(g), torch.randn_like(g)
        ms = t(lambda: ops.dwconv_sum([x, g2, g3], [wt, wt, wt], k, p, d, out=y))
        print(f"dw sum3 (one launch) C={C:5d}: {ms:7.3f} ms")
        def chain():
            r = ops.dwconv(x, wt, k, p, d, out=y)
            r = ops.dwconv(g2, wt, k, p, d, out=y, res_post=r)
            ops.dwconv(g3, wt, k, p, d, out=y, res_post=r)
        ms = t(chain)
        print(f"dw sum3 (chained)    C={C:5d}: {ms:7.3f} ms")
        ys = [torch.empty_like(x) for _ in range(3)]
        ms = t(lambda: ops.dwconv_fanout(x, [wt, wt, wt], k, p, d, outs=ys))
        print(f"dw fan-out 3 (one launch) C={C:5d}: {ms:7.3f} ms")
        ms = t(lambda: [ops.dwconv(x, wt, k, p, d, out=o) for o in ys])
        print(f"dw fan-out 3 (3 launches) C={C:5d}: {ms:7.3f} ms")
        dws = [torch.empty_like(w) for _ in range(3)]
        ms = t(lambda: ops.dwconv_wgrad_multi(x, [g, g2, g3], dws, k, p, d))
        print(f"dw wgrad of 3 (one launch)  C={C:5d}: {ms:7.3f} ms  {3 * fl/ms/1e9:7.1f} TFLOP/s")
        ms = t(lambda: [ops.dwconv_wgrad(x, gg, o, k, p, d, workspace=ws) for gg, o in zip((g, g2, g3), dws)])
        print(f"dw wgrad of 3 (3 launches)  C={C:5d}: {ms:7.3f} ms")
        del g2, g3, ys, dws
    ms = t(lambda: ops.dwconv_wgrad(x, g, dw, k, p, d, workspace=ws))
    print(f"dw wgrad C={C:5d}: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s")

# pointwise weight gradient at the student's trainable 1x1 shapes
for (Cin, Cout) in ((512, 512), (1024, 2048), (4096, 256)):
    H, W = 128, 256
    a = torch.randn(1, H, W, Cin, device="cuda").bfloat16()
    g = torch.randn(1, H, W, Cout, device="cuda").bfloat16()
    dw = torch.empty(Cout, Cin, 1, 1, device="cuda")
    ws = torch.empty(ops._lib.lib().kd_pw_wgrad_workspace(H * W, Cin, Cout), dtype=torch.uint8, device="cuda")
    ms = t(lambda: ops.pw_wgrad(a, g, dw, workspace=ws))
    print(f"pw wgrad {Cin}->{Cout}: {ms:7.3f} ms  {2.0 * H * W * Cin * Cout / ms / 1e9:7.1f} TFLOP/s")
