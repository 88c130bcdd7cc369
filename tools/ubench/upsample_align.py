import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kdcc_amd
from kdcc_amd import ops
x = torch.randn(4, 128, 256, 256, device="cuda").to(torch.bfloat16)
def t(y, name):
    for _ in range(3): ops.upsample_bilinear_ac(x, (512, 1024), out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.upsample_bilinear_ac(x, (512, 1024), out=y)
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1) / 10, "ms")
buf = torch.empty(4, 512, 1024, 320, device="cuda", dtype=torch.bfloat16)
t(torch.empty(4, 512, 1024, 256, device="cuda", dtype=torch.bfloat16), "contiguous 256")
t(buf[..., 48:304], "slice 48:304 of 320")
t(buf[..., 0:256], "slice 0:256 of 320")
t(buf[..., 64:320], "slice 64:320 of 320")
