#!/usr/bin/env python3
"""Time kd_conv3x3_small at the Gated-SCNN shape stream's sizes (8 x 1024 x 2048; C = 64 res1, 32 res2, 16 res3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kdcc_amd
from kdcc_amd import ops
for C in (64, 32, 16):
    x = torch.randn(8, 1024, 2048, C, device='cuda').bfloat16()
    res = torch.randn_like(x)
    w = (torch.randn(C, 3, 3, C, device='cuda') * 0.1).bfloat16()
    b = torch.randn(C, device='cuda')
    y = torch.empty_like(x)
    for _ in range(2): ops.conv3x3_small(x, w, b, res, True, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv3x3_small(x, w, b, res, True, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"conv3x3_small C={C}: {ms:.3f} ms  {3 * x.numel() * 2 / ms / 1e6:.0f} GB/s (x + res + y)")
