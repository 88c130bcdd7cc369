#!/usr/bin/env python3
"""Time kd_stem_conv_pool at the bench size (8 x 1024 x 2048, frozen stem: activated output only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kdcc_amd
from kdcc_amd import ops
x = torch.randn(8, 3, 1024, 2048, device='cuda')
w = torch.randn(64, 3, 3, 3, device='cuda') * 0.2
sc, sh = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda') * 0.1
for _ in range(2): ops.stem_conv_pool(x, w, sc, sh, want_raw=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.stem_conv_pool(x, w, sc, sh, want_raw=False)
e1.record(); torch.cuda.synchronize()
print('stem_conv_pool ms', e0.elapsed_time(e1) / 5)
