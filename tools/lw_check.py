#!/usr/bin/env python3
"""conv_row_lw_kernel against conv_row_persist_kernel<pp> (GPU box): the same launches in two child processes (KDCC_CONV_LW=1 / 0,
the switch is read once per process), outputs compared BIT FOR BIT (same k order into the same fp32 chains, same epilogue), and the
time of each.  usage: python tools/lw_check.py [--batch N] [--iters K] [--only substr]"""
import argparse
import hashlib
import zlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name, H, W, Cin, Cout, dil, operands, outputs
CASES = [
    ("mod4 512->512", 128, 256, 512, 512, 1, (), ("act",)),
    ("mod4 512->512 +res", 128, 256, 512, 512, 1, ("pre",), ("raw", "act")),
    ("mod3 256->256", 256, 512, 256, 256, 1, (), ("act",)),
    ("mod3 128->256", 256, 512, 128, 256, 1, (), ("raw", "act")),
    ("mod5 d2 512->1024", 128, 256, 512, 1024, 2, (), ("act",)),
    ("mod5 d2 1024->512", 128, 256, 1024, 512, 2, ("mask", "post"), ("raw",)),
    ("mod6 d4 512->1024", 128, 256, 512, 1024, 4, ("mask",), ("raw",)),
    ("mod7 d4 1024->2048", 128, 256, 1024, 2048, 4, (), ("act",)),
    ("aspp d12 4096->256", 128, 256, 4096, 256, 12, (), ("act",)),
    ("aspp d24 4096->256", 128, 256, 4096, 256, 24, (), ("act",)),
    ("final 320->256", 512, 1024, 320, 256, 1, (), ("act",)),
    ("final 256->256", 512, 1024, 256, 256, 1, ("pre", "mask", "post"), ("raw", "act")),
    ("small 64->512 d2", 80, 512, 64, 512, 2, ("pre", "post"), ("act",)),
    ("mod2 64->128", 512, 1024, 64, 128, 1, (), ("act",)),
    ("mod2 128->128", 512, 1024, 128, 128, 1, (), ("act",)),
    ("mod2 128->128 +res", 512, 1024, 128, 128, 1, ("pre",), ("raw", "act")),
    ("mod2 128->128 d3 W512", 96, 512, 128, 128, 3, ("mask", "post"), ("raw",)),
    ("mod2 192->256 d32 H40", 40, 512, 192, 128, 32, ("pre",), ("act",)),
    # 1x1 (dil 0 marks them): conv_pw_lw_kernel against conv_igemm_persist_kernel<pp>
    ("pw 512->512", 128, 256, 512, 512, 0, (), ("act",)),
    ("pw 1024->2048", 128, 256, 1024, 2048, 0, (), ("raw", "act")),
    ("pw 2048->4096", 128, 256, 2048, 4096, 0, ("pre",), ("raw", "act")),
    ("pw 4096->2048", 128, 256, 4096, 2048, 0, ("mask", "post"), ("raw",)),
    ("pw 2048->1024", 128, 256, 2048, 1024, 0, ("mask",), ("raw",)),
    ("pw 256->4096", 128, 256, 256, 4096, 0, (), ("act",)),
    ("pw 1280->256", 128, 256, 1280, 256, 0, (), ("act",)),
    ("pw 128->256 x3", 256, 512, 128, 256, 0, ("pre", "mask", "post"), ("raw", "act")),
]


def child(a):
    import torch
    import kdcc_amd
    from kdcc_amd import _lib, ops
    res = {}
    for name, H, W, Cin, Cout, d, opnds, outs in CASES:
        if a.only and not any(o and o in name for o in a.only.split(",")):
            continue
        g = torch.Generator(device="cuda").manual_seed(zlib.crc32(name.encode()) & 0xffff)
        rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
        x = rn(a.batch, H, W, Cin).relu().bfloat16()
        k = 3 if d else 1
        w = (rn(Cout, k, k, Cin) * (2.0 / (k * k * Cin)) ** 0.5).bfloat16()
        kw = {}
        if "pre" in opnds:
            kw["res_pre"] = rn(a.batch, H, W, Cout).bfloat16()
        if "mask" in opnds:
            kw["mask"] = rn(a.batch, H, W, Cout).relu().bfloat16()
            kw["mask_scale"] = torch.rand(Cout, device="cuda", generator=g) + 0.5
        if "post" in opnds:
            kw["res_post"] = rn(a.batch, H, W, Cout).bfloat16()
        raw = torch.zeros(a.batch, H, W, Cout, device="cuda", dtype=torch.bfloat16) if "raw" in outs else None
        act = torch.zeros(a.batch, H, W, Cout, device="cuda", dtype=torch.bfloat16) if "act" in outs else None
        if act is not None:
            kw.update(act_scale=torch.rand(Cout, device="cuda", generator=g) + 0.5, act_shift=rn(Cout) * 0.1, act_relu=True)
        with _lib.kernel_log() as log:
            ops.conv2d(x, w, 1, d, max(d, 1), out_raw=raw, out_act=act, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.conv2d(x, w, 1, d, max(d, 1), out_raw=raw, out_act=act, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        dig = [hashlib.sha256(t.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16] for t in (raw, act) if t is not None]
        if a.dump:
            torch.save([t.cpu() for t in (raw, act) if t is not None], os.path.join(a.dump, name.replace(" ", "_").replace(">", "") + ".pt"))
        fin = all(bool(torch.isfinite(t.float()).all()) for t in (raw, act) if t is not None)
        res[name] = {"ms": ms, "tflops": 2.0 * a.batch * H * W * Cout * k * k * Cin / ms / 1e9, "digest": dig, "finite": fin,
                     "kernel": [k for k, v in log.counts.items() if v], "absmean": float((act if act is not None else raw).float().abs().mean())}
    print("RESULT " + json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--dump", default="")
    ap.add_argument("--duo", default="0", help="KDCC_CONV_DUO of the first arm (0 off, 1 Cout = 128 layers, 2 every row layer)")
    a = ap.parse_args()
    if a.child:
        return child(a)
    import tempfile
    import torch
    out, dumps = {}, {}
    for lw in ("1", "0"):
        env = dict(os.environ, KDCC_CONV_LW=lw, KDCC_CONV_LW_PW="1", KDCC_CONV_DUO=a.duo if lw == "1" else "0")
        dumps[lw] = tempfile.mkdtemp()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--batch", str(a.batch), "--iters", str(a.iters),
                            "--only", a.only, "--dump", dumps[lw]], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if r.returncode != 0 or not line:
            print(f"child KDCC_CONV_LW={lw} failed rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
            sys.exit(1)
        out[lw] = json.loads(line[0][7:])
    bad = 0
    print(f"{'case':26s} {'lw ms':>8s} {'pp ms':>8s} {'lw TF/s':>8s} {'pp TF/s':>8s}  bitwise")
    tl = tp = 0.0
    for name in out["1"]:
        l, p = out["1"][name], out["0"][name]
        same = l["digest"] == p["digest"] and l["finite"]
        fn = name.replace(" ", "_").replace(">", "") + ".pt"
        ta, tb = torch.load(os.path.join(dumps["1"], fn)), torch.load(os.path.join(dumps["0"], fn))
        rel = max(float((x.float() - y.float()).norm() / y.float().norm().clamp_min(1e-30)) for x, y in zip(ta, tb))
        mx = max(float((x.float() - y.float()).abs().max() / y.float().abs().max()) for x, y in zip(ta, tb))
        ok = l["finite"] and (same or (rel < 2e-3 and mx < 2e-2))        # (another k order: one-ulp bf16 flips)
        if not ok:
            for k, (x, y) in enumerate(zip(ta, tb)):
                ne = (x.view(torch.int16) != y.view(torch.int16))
                idx = ne.nonzero()
                print(f"    output {k}: {int(ne.sum())} of {ne.numel()} differ; non-finite lw {int((~torch.isfinite(x.float())).sum())} pp {int((~torch.isfinite(y.float())).sum())}; "
                      f"first {idx[:3].tolist()} last {idx[-2:].tolist()}; lw {x[tuple(idx[0])].item() if len(idx) else None} pp {y[tuple(idx[0])].item() if len(idx) else None}")
        bad += not ok
        tl += l["ms"]; tp += p["ms"]
        print(f"{name:26s} {l['ms']:8.3f} {p['ms']:8.3f} {l['tflops']:8.0f} {p['tflops']:8.0f}  {'identical' if same else ('rel L2 %.1e max %.1e' % (rel, mx)) + ('' if ok else ' DIFFERENT')}"
              f"  {l['kernel']} vs {p['kernel']} |y| {l['absmean']:.4f} / {p['absmean']:.4f}")
    print(f"total {tl:.3f} ms (lw) vs {tp:.3f} ms (pp): {tp / tl:.3f}x")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
