#!/usr/bin/env python3
"""Per-shape microbenchmark of kd_conv2d_fwd on the student's conv shapes at 1024x2048 (GPU box only)."""
import sys, os, argparse
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kdcc_amd
from kdcc_amd import ops

# (name, H, W, Cin, Cout, k, stride, dil, count per step [fwd student P92])
SHAPES = [
    ("mod2 3x3 64->128", 512, 1024, 64, 128, 3, 1, 1, 1),
    ("mod2 3x3 128->128", 512, 1024, 128, 128, 3, 1, 1, 5),
    ("mod3 3x3 128->256", 256, 512, 128, 256, 3, 1, 1, 1),
    ("mod3 3x3 256->256", 256, 512, 256, 256, 3, 1, 1, 5),
    ("mod4 3x3 s2 256->512", 256, 512, 256, 512, 3, 2, 1, 1),
    ("mod4 3x3 512->512", 128, 256, 512, 512, 3, 1, 1, 9),
    ("mod5 3x3 d2 512->1024", 128, 256, 512, 1024, 3, 1, 2, 3),
    ("mod5 3x3 d2 1024->512", 128, 256, 1024, 512, 3, 1, 2, 2),
    ("mod6 3x3 d4 512->1024", 128, 256, 512, 1024, 3, 1, 4, 1),
    ("pw 512->512", 128, 256, 512, 512, 1, 1, 1, 3),
    ("1x1 1024->2048", 128, 256, 1024, 2048, 1, 1, 1, 3),
    ("1x1 2048->4096", 128, 256, 2048, 4096, 1, 1, 1, 2),
    ("1x1 4096->256", 128, 256, 4096, 256, 1, 1, 1, 4),
    ("1x1 256->4096", 128, 256, 256, 4096, 1, 1, 1, 3),
    ("1x1 4096->2048", 128, 256, 4096, 2048, 1, 1, 1, 2),
    ("1x1 2048->1024", 128, 256, 2048, 1024, 1, 1, 1, 5),
    ("mod7 3x3 d4 1024->2048", 128, 256, 1024, 2048, 3, 1, 4, 1),
    ("1x1 1280->256", 128, 256, 1280, 256, 1, 1, 1, 1),
    ("final 3x3 320->256", 512, 1024, 320, 256, 3, 1, 1, 1),
    ("final 3x3 256->256", 512, 1024, 256, 256, 3, 1, 1, 1),
    ("cls 1x1 256->19", 512, 1024, 256, 19, 1, 1, 1, 1),
    ("bot_fine 1x1 128->48", 512, 1024, 128, 48, 1, 1, 1, 1),
]

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--iters", type=int, default=10); ap.add_argument("--only", default="")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    tot_f, tot_t = 0.0, 0.0
    for name, H, W, Cin, Cout, k, s, d, cnt in SHAPES:
        if a.only and not any(t in name for t in a.only.split(",")): continue
        NB = int(os.environ.get("KDCC_BENCH_BATCH", "2")); x = torch.randn(NB, H, W, Cin, device="cuda").to(dt)
        w = (torch.randn(Cout, k, k, Cin, device="cuda") * 0.05).to(dt)
        pad = d * (k - 1) // 2
        Ho, Wo = ops.conv_out_size(H, k, s, pad, d), ops.conv_out_size(W, k, s, pad, d)
        out = torch.empty(NB, Ho, Wo, Cout, device="cuda", dtype=dt)
        sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
        for _ in range(2): ops.conv2d(x, w, s, pad, d, out_act=out, act_scale=sc, act_shift=sh, act_relu=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(a.iters): ops.conv2d(x, w, s, pad, d, out_act=out, act_scale=sc, act_shift=sh, act_relu=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        fl = 2.0 * NB * Ho * Wo * Cout * k * k * Cin
        tot_f += fl * cnt; tot_t += ms * cnt
        print(f"{name:26s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s   x{cnt}")
    print(f"weighted total {tot_t:.2f} ms for {tot_f / 1e12:.2f} TFLOP -> {tot_f / tot_t / 1e9:.1f} TFLOP/s")

if __name__ == "__main__":
    main()
