#!/usr/bin/env python3
"""Per-tile timeline of the persistent conv kernels (KDCC_CONV_TUNE=512 timestamps, debug only; GPU box)."""
import os, sys, ctypes as C
os.environ["KDCC_CONV_TUNE"] = os.environ.get("KDCC_CONV_TUNE", "512")
os.environ["KDCC_LIB"] = "tuning"   # the timestamps exist only in the diagnostics build (make -C csrc TUNING=1)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kdcc_amd
from kdcc_amd import _lib
_lib.build_tuning()
from kdcc_amd import ops

def run(name, H, W, Cin, Cout, k, d, NB=4, res=False):
    dt = torch.bfloat16
    x = torch.randn(NB, H, W, Cin, device="cuda").to(dt)
    w = (torch.randn(Cout, k, k, Cin, device="cuda") * 0.05).to(dt)
    pad = d * (k - 1) // 2
    out = torch.empty(NB, H, W, Cout, device="cuda", dtype=dt)
    raw = torch.empty_like(out)
    r = torch.randn_like(out) if res else None
    sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
    for _ in range(3):
        ops.conv2d(x, w, 1, pad, d, res_pre=r, out_raw=raw if res else None, out_act=out, act_scale=sc, act_shift=sh, act_relu=True)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 32 * 8 + 256 * 64, dtype=np.uint64)
    lib = _lib.lib()
    lib.kd_debug_conv_tlog.argtypes = [C.c_void_p, C.c_size_t]
    lib.kd_debug_conv_tlog(buf.ctypes.data, buf.nbytes)
    pw = buf[256 * 32 * 8:].reshape(256, 8, 8).astype(np.float64).mean(0)
    t = buf[:256 * 32 * 8].reshape(256, 32, 8).astype(np.float64) / 100.0   # us
    ntile = int((t[0, :, 0] > 0).sum())
    print(f"== {name}: tiles per workgroup {ntile}")
    nst = H * 0 + (9 if k == 3 else 1) * Cin // 64
    print("   phase cycles per stage -- waves 0-3: reads | barrier | MFMAs | DMA wait | barrier | DMA issue; waves 4-7: DMA wait | barrier | DMA issue | reads | barrier | MFMAs")
    for wv in range(8):
        print(f"     wave {wv}: " + " ".join(f"{v / nst:7.0f}" for v in pw[wv, :6]), f"  sum {pw[wv, :6].sum() / nst:7.0f}")
    for i in range(min(ntile, 6)):
        a = t[:, i, :]
        top = (a[:, 1] - a[:, 0]).mean()
        main = (a[:, 4] - a[:, 1]).mean(); pro = (a[:, 5] - a[:, 4]).mean()
        setup = (a[:, 7] - a[:, 4]).mean() if i + 1 < ntile else 0.0; epi = (a[:, 6] - a[:, 5]).mean()
        nxt = (t[:, i + 1, 0] - a[:, 6]).mean() if i + 1 < ntile else float("nan")
        print(f"  tile {i}: top wait {top:6.2f}  main loop {main:7.2f}  setup+prologue {pro:5.2f} (setup {setup:5.2f})  epilogue {epi:5.2f}  -> next {nxt:5.2f} us")

run("3x3 512->512 [a]", 128, 256, 512, 512, 3, 1)
run("3x3 512->512 [pra]", 128, 256, 512, 512, 3, 1, res=True)
run("1x1 1024->2048 [a]", 128, 256, 1024, 2048, 1, 1)
run("1x1 1024->2048 [pra]", 128, 256, 1024, 2048, 1, 1, res=True)


def run_pp128(name, H, W, Cin, NB=4):
    """phase clocks of the 512 x 128 ping-pong kernel (no-operand epilogue)"""
    dt = torch.bfloat16
    x = torch.randn(NB, H, W, Cin, device="cuda").to(dt)
    w = (torch.randn(128, 3, 3, Cin, device="cuda") * 0.05).to(dt)
    out = torch.empty(NB, H, W, 128, device="cuda", dtype=dt)
    sc = torch.ones(128, device="cuda"); sh = torch.zeros(128, device="cuda")
    for _ in range(3):
        ops.conv2d(x, w, 1, 1, 1, out_act=out, act_scale=sc, act_shift=sh, act_relu=True)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 32 * 8 + 256 * 64, dtype=np.uint64)
    lib = _lib.lib()
    lib.kd_debug_conv_tlog.argtypes = [C.c_void_p, C.c_size_t]
    lib.kd_debug_conv_tlog(buf.ctypes.data, buf.nbytes)
    pw = buf[256 * 32 * 8:].reshape(256, 8, 8).astype(np.float64).mean(0)
    print(f"== {name}: phase cycles per 64-B stage -- waves 0-3: reads | barrier | MFMAs | DMA wait | barrier | DMA issue; waves 4-7: DMA wait | barrier | DMA issue | reads | barrier | MFMAs")
    for wv in range(8):
        nst = max(pw[wv, 6], 1.0)
        print(f"     wave {wv}: " + " ".join(f"{v / nst:7.0f}" for v in pw[wv, :6]), f"  sum {pw[wv, :6].sum() / nst:7.0f}")

run_pp128("3x3 128->128 @512x1024 (pp128)", 512, 1024, 128)
