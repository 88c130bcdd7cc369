#!/usr/bin/env python3
"""conv_wgrad_lw_kernel (one wave per SIMD, generated stage loop: tools/gen_wgrad_lw.py) against conv_wgrad_row_kernel (8 waves), the
kernel it replaces on the 3x3 layers with Cout % 128 == 0: the same launches in two FRESH child processes
(KDCC_WGRAD_LW = KDCC_WGRAD_PW_LW = 1 / 0: the switches are read once per process; the 1x1 cases compare conv_wgrad_pw_lw_kernel with
conv_wgrad_wide_kernel<true>), the fp32 weight gradients compared BIT FOR BIT (same decomposition, same
LDS images, same k order into the same fp32 chains: any difference is a defect), and the time of each.
usage: python tools/wgrad_lw_check.py [--batch N] [--iters K] [--only substr,substr]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name, N (None: --batch), H, W, Cin, Cout, dil      (mode B's 3x3 layers, models/encoders/wider_resnet.py:124-167 of the reference)
CASES = [
    ("mod2 128->128", None, 512, 1024, 128, 128, 1),
    ("mod3 128->256", None, 256, 512, 128, 256, 1),
    ("mod3 256->256", None, 256, 512, 256, 256, 1),
    ("mod4 256->512", None, 128, 256, 256, 512, 1),
    ("mod4 512->512", None, 128, 256, 512, 512, 1),
    ("mod5 d2 512->1024", None, 128, 256, 512, 1024, 2),
    ("mod5 d2 1024->512", None, 128, 256, 1024, 512, 2),
    ("mod6 d4 512->1024", None, 128, 256, 512, 1024, 4),
    ("mod7 d4 1024->2048", None, 128, 256, 1024, 2048, 4),
    ("final 256->256", None, 512, 1024, 256, 256, 1),
    ("final 304->256", None, 512, 1024, 304, 256, 1),      # the decoder's first 3x3: a ragged third Cin tile (48 of 128 channels)
    ("mod2 64->128", None, 512, 1024, 64, 128, 1),         # half a Cin tile
    # small / edge shapes: fewer stages than the ring holds, one stage per image row (W = 64: first and last tile at once), the
    # largest dilation of the kernel, kernel rows that leave the image for most of the rows, splits that end inside an image
    ("edge 3 rows W64", 1, 3, 64, 128, 128, 1),
    ("edge 1 row W64 d8", 1, 1, 64, 128, 128, 8),
    ("edge d8 W128", 2, 9, 128, 128, 256, 8),
    ("edge d5 W192", 3, 7, 192, 256, 128, 5),
    ("edge d2 20 rows", 1, 20, 128, 128, 128, 2),
    ("edge 2 imgs W64", 2, 9, 64, 128, 128, 1),
    ("edge d3 13 imgs", 13, 5, 64, 128, 384, 3),
    ("edge ragged 304", 1, 12, 64, 304, 256, 1),
    ("edge ragged 72 d4", 2, 8, 64, 72, 128, 4),
    ("edge ragged 200 d2", 1, 5, 128, 200, 128, 2),
    ("edge ragged 8 d8", 1, 4, 64, 8, 128, 8),
    # 1x1 / stride 1 (dil 0 marks them): conv_wgrad_pw_lw_kernel against conv_wgrad_wide_kernel<true> (KDCC_WGRAD_PW_LW)
    ("pw 2048->4096", None, 128, 256, 2048, 4096, 0),
    ("pw 1024->2048", None, 128, 256, 1024, 2048, 0),
    ("pw 2048->1024", None, 128, 256, 2048, 1024, 0),
    ("pw 4096->256", None, 128, 256, 4096, 256, 0),
    ("pw 512->1024", None, 128, 256, 512, 1024, 0),
    ("pw 1280->256", None, 128, 256, 1280, 256, 0),
    ("pw 512->512", None, 128, 256, 512, 512, 0),
    ("pw edge 2 stages", 1, 1, 64, 256, 256, 0),          # fewer stages than the prologue stages
    ("pw edge 8 stages", 1, 4, 64, 512, 256, 0),
    ("pw edge 30 stages", 3, 5, 64, 256, 512, 0),
    ("pw edge 1 img 128x128", 1, 128, 128, 256, 256, 0),
]


def child(a):
    import torch
    import kdcc_amd  # noqa: F401
    from kdcc_amd import _lib, ops
    res = {}
    for name, N, H, W, Cin, Cout, d in CASES:
        if a.only and not any(o and o in name for o in a.only.split(",")):
            continue
        N = N or a.batch
        g = torch.Generator(device="cuda").manual_seed(zlib.crc32(name.encode()) & 0xffff)
        x = torch.randn((N, H, W, Cin + 16), device="cuda", generator=g).relu().bfloat16()[..., :Cin]      # (a pixel stride that is not the channel count)
        dy = (torch.randn((N, H, W, Cout), device="cuda", generator=g) * 0.1).bfloat16()
        k = 3 if d else 1
        dw = torch.full((Cout, Cin, k, k), 7.0, device="cuda")
        run = (lambda: ops.conv2d_wgrad(x, dy, dw, 1, d, max(d, 1))) if name.find("pwapi") < 0 else (lambda: ops.pw_wgrad(x, dy, dw))
        run()
        kern = _lib.last_kernel()
        kern = kern if isinstance(kern, str) else ",".join(kern)
        torch.cuda.synchronize()
        first = hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        again = hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest()
        res[name] = {"kernel": kern, "digest": first, "stable": first == again, "finite": bool(torch.isfinite(dw).all()),
                     "absmax": float(dw.abs().max()), "ms": e0.elapsed_time(e1) / a.iters,
                     "tflops": 2.0 * N * H * W * Cin * Cout * k * k / (e0.elapsed_time(e1) / a.iters) / 1e9}
        print(f"{name}: {kern} {res[name]['ms']:.3f} ms {res[name]['tflops']:.0f} TFLOP/s", file=sys.stderr, flush=True)
    print("RESULT " + json.dumps(res), flush=True)


def arm(lw, a, **extra):
    env = dict(os.environ, KDCC_WGRAD_LW=lw, KDCC_WGRAD_PW_LW=lw)
    env.update(extra)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--batch", str(a.batch), "--iters", str(a.iters), "--only", a.only],
                       env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if r.returncode or not line:
        raise SystemExit(f"child KDCC_WGRAD_LW={lw} failed rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
    return json.loads(line[0][7:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if a.child:
        return child(a)
    lw, row = arm("1", a), arm("0", a)
    bad = 0
    for name in lw:
        same = lw[name]["digest"] == row[name]["digest"]
        bad += not (same and lw[name]["stable"] and lw[name]["finite"])
        print(f"{name:24s} {lw[name]['kernel']:22s} {lw[name]['ms']:8.3f} ms {lw[name]['tflops']:6.0f} TF | {row[name]['kernel']:22s} {row[name]['ms']:8.3f} ms "
              f"{row[name]['tflops']:6.0f} TF | x{row[name]['ms'] / lw[name]['ms']:.2f} {'bit-identical' if same else 'DIFFERENT'}"
              f"{'' if lw[name]['stable'] else ' UNSTABLE'} absmax {lw[name]['absmax']:.3g}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
