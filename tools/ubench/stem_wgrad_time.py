import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kdcc_amd
from kdcc_amd import ops
x = torch.randn(8, 3, 1024, 2048, device='cuda')
dy = torch.randn(8, 1024, 2048, 64, device='cuda').to(torch.bfloat16)
dw = torch.empty(64, 3, 3, 3, device='cuda')
for _ in range(2): ops.stem_wgrad(x, dy, dw)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.stem_wgrad(x, dy, dw)
e1.record(); torch.cuda.synchronize()
print('stem_wgrad ms', e0.elapsed_time(e1) / 5)
