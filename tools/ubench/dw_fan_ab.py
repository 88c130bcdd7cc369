"""The ASPP fan-out launch (8 x 128 x 256 x 4096, 9x9 / dilation 5, three branches) on the lone-wave kernel (dwconv_lw.hip) and on the
8-wave kernel it replaces (KDCC_DW_LW=0 in a second process), HIP events; also checks the two against each other."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kdcc_amd
from kdcc_amd import ops, _lib
N, H, W, C, k, p, d = int(os.environ.get("N", 8)), 128, 256, int(os.environ.get("C", 4096)), 9, 20, 5
torch.manual_seed(1)
x = torch.randn(N, H, W, C, device="cuda").bfloat16()
w = [ops.pack_dw_weight(torch.randn(C, 1, k, k, device="cuda") / k) for _ in range(3)]
ys = [torch.empty_like(x) for _ in range(3)]
f = lambda: ops.dwconv_fanout(x, w, k, p, d, outs=ys)
f(); torch.cuda.synchronize()
print("kernel:", _lib.last_kernel(), flush=True)
ref = [ops.dwconv(x, w[i], k, p, d) for i in range(3)]
for i in range(3):
    diff = (ys[i].float() - ref[i].float()).abs()
    print(f"branch {i}: max |diff| vs single launch {float(diff.max()):.4g} (max |ref| {float(ref[i].float().abs().max()):.4g}), differing {float((diff > 0).float().mean()):.4%}", flush=True)
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
print(json.dumps({"kernel": _lib.last_kernel(), "ms": [round(t, 3) for t in ts], "images": N, "channels": C}))
