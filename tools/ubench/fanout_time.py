import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kdcc_amd
if os.environ.get("KDCC_LIB") == "tuning": kdcc_amd._lib.build_tuning()
from kdcc_amd import ops
N, H, W, C, k, p, d = 8, 128, 256, 4096, 9, 20, 5
x = torch.randn(N, H, W, C, device="cuda").bfloat16()
w = [ops.pack_dw_weight(torch.randn(C, 1, k, k, device="cuda") / k) for _ in range(3)]
ys = [torch.empty_like(x) for _ in range(3)]
for _ in range(3): ops.dwconv_fanout(x, w, k, p, d, outs=ys)
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.dwconv_fanout(x, w, k, p, d, outs=ys)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 10)
print("fan-out 8 images ms:", " ".join(f"{v:.3f}" for v in res), "dbg", os.environ.get("KDCC_DW_DBG"))
