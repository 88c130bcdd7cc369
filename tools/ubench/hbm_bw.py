import torch, time
for mb in (64, 636, 2048):
    x = torch.empty(mb * 1024 * 1024 // 4, device="cuda", dtype=torch.float32)
    y = torch.empty_like(x)
    for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: y.copy_(x)), ("mul", lambda: torch.mul(x, 2.0, out=y))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        bytes_ = mb * 1024 * 1024 * (1 if name == "fill" else 2)
        print(f"{name:5s} {mb:5d} MB: {ms:7.3f} ms  {bytes_ / ms / 1e9:7.2f} TB/s")
