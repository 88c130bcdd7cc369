#!/usr/bin/env python3
"""A/B timing of kd_conv2d_wgrad / kd_pw_wgrad shapes in ONE process on one device (interleaved rounds).
usage: KDCC_WGRAD_NST=2|3|4 python tools/bench_wgrad.py   (compare runs only within one gpurun call)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdcc_amd  # noqa: E402
if os.environ.get("KDCC_LIB") == "tuning":   # A/B switches that exist only in the diagnostics build (e.g. KDCC_WGRAD_IL=1)
    kdcc_amd._lib.build_tuning()
from kdcc_amd import ops  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout, k, dil   (P92 mode-B layer classes at 4 images)
    (4, 512, 1024, 128, 128, 3, 1), (4, 256, 512, 256, 256, 3, 1), (4, 128, 256, 512, 512, 3, 1), (4, 128, 256, 512, 1024, 3, 2),
    (4, 128, 256, 2048, 4096, 1, 1), (4, 128, 256, 4096, 256, 1, 1), (4, 512, 1024, 304, 256, 3, 1), (4, 128, 256, 512, 512, 1, 1),
    (4, 512, 1024, 64, 128, 3, 1), (4, 256, 512, 128, 256, 3, 1), (4, 512, 1024, 256, 256, 3, 1), (4, 128, 256, 1024, 512, 3, 2)]
g = torch.Generator(device="cuda").manual_seed(0)
for (N, H, W, Ci, Co, k, d) in SHAPES:
    x = torch.randn((N, H, W, Ci), device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn((N, H, W, Co), device="cuda", generator=g).to(torch.bfloat16)
    dw = torch.empty((Co, Ci, k, k), device="cuda")
    pad = d if k == 3 else 0
    for _ in range(2):
        ops.conv2d_wgrad(x, dy, dw, 1, pad, d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv2d_wgrad(x, dy, dw, 1, pad, d)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * N * H * W * Ci * Co * k * k
    print(f"wgrad {Ci:5d}->{Co:5d} k{k} d{d} @{H}x{W}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
