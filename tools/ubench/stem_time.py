"""kd_stem_conv_pool (mod1 -> pool2 -> bn1 / ReLU of mod2.block1, frozen stem) at the bench's size, 8 x 3 x 1024 x 2048 fp32 NCHW in, HIP events."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kdcc_amd
from kdcc_amd import ops
N, H, W = int(os.environ.get("N", 8)), 1024, 2048
torch.manual_seed(0)
x = torch.randn(N, 3, H, W, device="cuda")
w = (torch.randn(64, 3, 3, 3, device="cuda") * 0.2)
sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1
f = lambda: ops.stem_conv_pool(x, w, sc, sh, want_raw=False)
f(); torch.cuda.synchronize()
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
_, act = f()
print(json.dumps({"ms": [round(t, 3) for t in ts], "checksum": float(act.float().sum()), "absmax": float(act.float().abs().max())}))
