#!/usr/bin/env python3
"""The logged logit losses at the bench's size (8 x 512 x 1024 x 19 logits -> 1024 x 2048): up-sample + ce2d x2 + kldiv on the
materialised tensors against ce2d_up x2 + kldiv_up on the half-resolution ones (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kdcc_amd
from kdcc_amd import ops

N, h, w, C, H, W = 8, 512, 1024, 19, 1024, 2048
g = torch.Generator(device="cuda").manual_seed(0)
s_lo, t_lo = torch.randn(N, h, w, C, device="cuda", generator=g), torch.randn(N, h, w, C, device="cuda", generator=g)
tgt = torch.randint(0, C, (N, H, W), device="cuda", generator=g)
tgt[:, :32] = 255


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r


def eager():
    s = ops.upsample_bilinear_ac(s_lo, (H, W), out_dtype=torch.float32).permute(0, 3, 1, 2)
    t = ops.upsample_bilinear_ac(t_lo, (H, W), out_dtype=torch.float32).permute(0, 3, 1, 2)
    return ops.ce2d(s, tgt), ops.kldiv(s, t, 1.0, want_grad=False)[0], ops.ce2d(t, tgt)


def lazy():
    return ops.ce2d_up(s_lo, tgt, (H, W)), ops.kldiv_up(s_lo, t_lo, (H, W), 1.0), ops.ce2d_up(t_lo, tgt, (H, W))


for name, f in (("materialised", eager), ("from low resolution", lazy), ("ce2d_up alone", lambda: ops.ce2d_up(s_lo, tgt, (H, W))),
                ("kldiv_up alone", lambda: ops.kldiv_up(s_lo, t_lo, (H, W), 1.0))):
    ms, r = timed(f)
    print(f"{name:22s} {ms:7.3f} ms  ", [float(x) for x in (r if isinstance(r, tuple) else (r,))])
