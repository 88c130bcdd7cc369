"""The three depthwise launches of the replaced ASPP branches at the bench's size (8 x 128 x 256 x 4096, 9x9 / dilation 5, three
branches): fan-out forward, summed input gradient, three-way weight gradient -- NHWC intermediates against lattice-planar ones
(include/kdcc.h), alternating, HIP events.  GB/s over the algorithmic bytes (every tensor read or written once = 4 x 2.15 GB)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kdcc_amd
if os.environ.get("KDCC_LIB") == "tuning": kdcc_amd._lib.build_tuning()
from kdcc_amd import ops
N, H, W, C, k, p, d = int(os.environ.get("N", 8)), 128, 256, 4096, 9, 20, 5
x = torch.randn(N, H, W, C, device="cuda").bfloat16()
w = [ops.pack_dw_weight(torch.randn(C, 1, k, k, device="cuda") / k) for _ in range(3)]
wt = [ops.pack_dw_weight(torch.randn(C, 1, k, k, device="cuda") / k, flip=True) for _ in range(3)]
ys = [torch.empty_like(x) for _ in range(3)]
ls = [ops.Lattice(N, H, W, C, d) for _ in range(3)]
gx, dws = torch.empty_like(x), [torch.empty(C, 1, k, k, device="cuda") for _ in range(3)]
runs = {
    "fanout_nhwc": lambda: ops.dwconv_fanout(x, w, k, p, d, outs=ys),
    "fanout_lattice": lambda: ops.dwconv_fanout_lattice(x, w, k, p, d, outs=ls),
    "sum_nhwc": lambda: ops.dwconv_sum(ys, wt, k, d * (k - 1) - p, d, out=gx),
    "sum_lattice": lambda: ops.dwconv_sum_lattice(ls, wt, k, d * (k - 1) - p, d, out=gx),
    "wgrad_nhwc": lambda: ops.dwconv_wgrad_multi(x, ys, dws, k, p, d),
    "wgrad_lattice": lambda: ops.dwconv_wgrad_multi_lattice(x, ls, dws, k, p, d),
}
only = os.environ.get("ONLY")
if only == "fs":
    runs = {k: v for k, v in runs.items() if not k.startswith("wgrad")}
for f in runs.values():
    f(); f()
torch.cuda.synchronize()
res = {n: [] for n in runs}
for rep in range(3):
    for name, f in runs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10)
gb = 4 * x.numel() * 2 / 1e9
out = {n: {"ms": [round(v, 3) for v in r], "GBps": round(gb / (min(r) * 1e-3)), "frac_of_8TBps": round(gb / (min(r) * 1e-3) / 8000, 3)} for n, r in res.items()}
out["bytes_per_launch_GB"] = round(gb, 3)
out["images"] = N
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
