#!/usr/bin/env python3
"""Phase picture of conv_row_duo_kernel (tuning build, GPU box): per workgroup and tile the 100-MHz stamps tile start / main loop
end / epilogue end (KDCC_CONV_TUNE=1024).  Prints, for a few CUs, the two co-resident workgroups (block b and b + 256) side by side
and how much of each epilogue overlapped the partner's main loop.
usage: KDCC_LIB=tuning KDCC_CONV_DUO=2 KDCC_CONV_TUNE=1024 python tools/duo_timeline.py
       KDCC_LIB=tuning KDCC_CONV_TUNE=1024 python tools/duo_timeline.py --tall     (conv_row_tall_kernel: one workgroup per CU)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdcc_amd
from kdcc_amd import _lib, ops

N, H, W, Cin, Cout = 8, 512, 1024, 128, 128
x = torch.randn(N, H, W, Cin, device="cuda").relu().bfloat16()
w = (torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.03).bfloat16()
out = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16)
sc, sh = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
for _ in range(3):
    with _lib.kernel_log() as log:
        ops.conv2d(x, w, 1, 1, 1, out_act=out, act_scale=sc, act_shift=sh, act_relu=True)
torch.cuda.synchronize()
print("kernel:", dict(log.counts))
buf = np.zeros(512 * 32 * 8, dtype=np.uint64)
lib = _lib.lib()
lib.kd_debug_conv_tlog.argtypes = [C.c_void_p, C.c_size_t]
lib.kd_debug_conv_tlog(buf.ctypes.data, buf.nbytes)
t = buf.reshape(512, 32, 8).astype(np.int64)
t0 = t[:, 0, 0].min()
us = lambda v: (v - t0) / 100.0
if "--tall" in sys.argv:
    for b in (0, 9, 100, 255):
        a = t[b]
        print(f"--- block {b}: tile start, main-loop end, epilogue end [us]; main loop, boundary (epilogue + refill)")
        for k in range(8):
            print(f"  tile {k}: {us(a[k,0]):8.1f} {us(a[k,1]):8.1f} {us(a[k,2]):8.1f}   {(a[k,1]-a[k,0])/100:6.1f} {(a[k+1,0]-a[k,1])/100:6.1f}")
    ml = (t[:256, 1:30, 1] - t[:256, 1:30, 0]) / 100.0
    bd = (t[:256, 2:30, 0] - t[:256, 1:29, 1]) / 100.0
    print(f"main loop per tile {ml.mean():.2f} us (sd {ml.std():.2f}; min {ml.min():.2f}, max {ml.max():.2f}), boundary {bd.mean():.2f} us (sd {bd.std():.2f})")
    seg = lambda a, b: ((t[:256, 1:30, a] - t[:256, 1:30, b]) / 100.0).mean()
    print(f"epilogue phases [us]: accumulators read + packed (half 0) {seg(4, 1):.2f}, staged loads landed {seg(5, 4):.2f}, half 0 transposed + stored {seg(6, 5):.2f}, "
          f"accumulators (half 1) {seg(7, 6):.2f}, half 1 {seg(2, 7):.2f}")
    sys.exit(0)
hidden = []
for b in (0, 1, 9, 100, 255):
    a, c = t[b], t[b + 256]
    print(f"--- blocks {b} / {b + 256} (one CU): tile start, main-loop end, epilogue end [us]")
    for k in range(6):
        print(f"  tile {k}: A {us(a[k,0]):8.1f} {us(a[k,1]):8.1f} {us(a[k,2]):8.1f}   B {us(c[k,0]):8.1f} {us(c[k,1]):8.1f} {us(c[k,2]):8.1f}")
for b in range(256):
    for me, other in ((t[b], t[b + 256]), (t[b + 256], t[b])):
        for k in range(2, 30):
            e0, e1 = me[k, 1], me[k, 2]          # my epilogue
            ov = 0
            for j in range(32):
                m0, m1 = other[j, 0], other[j, 1]    # partner's main loop
                ov += max(0, min(e1, m1) - max(e0, m0))
            hidden.append(ov / max(e1 - e0, 1))
hidden = np.array(hidden)
ep = (t[:, 2:30, 2] - t[:, 2:30, 1]) / 100.0
ml = (t[:, 2:30, 1] - t[:, 2:30, 0]) / 100.0
print(f"main loop per tile {ml.mean():.1f} us (sd {ml.std():.1f}), epilogue {ep.mean():.1f} us (sd {ep.std():.1f}); "
      f"fraction of an epilogue under the partner's main loop: mean {hidden.mean():.2f}, median {np.median(hidden):.2f}")
