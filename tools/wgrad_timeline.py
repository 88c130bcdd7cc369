#!/usr/bin/env python3
"""Phase clocks of conv_wgrad_row_kernel (KDCC_WGRAD_DBG=1; debug, GPU box): cycles per K stage spent waiting for the DMA,
at the barrier, issuing the next stage's DMA, and in the fragment reads + MFMAs."""
import os, sys, ctypes as C
os.environ.setdefault("KDCC_WGRAD_DBG", "1")   # | 2 no MFMAs | 4 no fragment reads | 8 no DMA (timing ablations)
os.environ["KDCC_LIB"] = "tuning"   # the timestamps exist only in the diagnostics build (make -C csrc TUNING=1)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdcc_amd
from kdcc_amd import _lib
_lib.build_tuning()
from kdcc_amd import ops

for (N, H, W, Ci, Co, d) in [(4, 512, 1024, 128, 128, 1), (4, 128, 256, 512, 512, 1), (4, 256, 512, 256, 256, 1)]:
    x = torch.randn((N, H, W, Ci), device="cuda").to(torch.bfloat16)
    dy = torch.randn((N, H, W, Co), device="cuda").to(torch.bfloat16)
    dw = torch.empty((Co, Ci, 3, 3), device="cuda")
    for _ in range(2):
        ops.conv2d_wgrad(x, dy, dw, 1, d, d)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    lib = _lib.lib()
    lib.kd_debug_wgrad_tlog.argtypes = [C.c_void_p, C.c_size_t]
    lib.kd_debug_wgrad_tlog(buf.ctypes.data, buf.nbytes)
    t = buf.reshape(256, 8, 8).astype(np.float64)
    nst = np.maximum(t[:, :, 4], 1.0)
    per = (t[:, :, :4] / nst[:, :, None]).mean(0)
    print(f"wgrad {Ci}->{Co} @{H}x{W}: stages per workgroup {nst.mean():.0f}; cycles per stage (whole main loop; the phase columns are gone with the software pipeline):")
    for wv in (0, 4):
        print(f"   wave {wv}: " + " ".join(f"{v:7.0f}" for v in per[wv]), f" sum {per[wv].sum():7.0f}")
