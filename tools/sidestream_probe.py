#!/usr/bin/env python3
"""Does a side-stream kernel run UNDER the persistent conv launches of a mode-A backward, or only between them?  (GPU box, one GPU.)

The headline step (bench.build / kd_step: P92, mode A, 8 x 1024 x 2048) with parallel.GradReducer's buckets; where the reducer would
launch the RCCL all-reduce of a finished bucket -- on its side stream, from inside backward -- a stand-in runs instead: a copy of the
bucket into a scratch buffer (one kernel, ~35-70 us of a few CUs' time: what one step of a ring all-reduce over an 8-MB bucket asks
of the chip), bracketed by events.  Per bucket: `wait` = stand-in start minus the moment the bucket's last gradient kernel finished
(how long the exchange sat behind conv kernels that hold every CU), `run` = stand-in duration.  For kd_conv_set_persist_cus(n) in
256 / 248 / 240 / 224: step time and those latencies.  usage: python tools/sidestream_probe.py [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import kdcc_amd  # noqa: E402
from kdcc_amd import _lib, parallel  # noqa: E402


class ProbeReducer(parallel.GradReducer):
    """GradReducer whose exchange is a stand-in copy on the side stream, with events (single process: no process group)."""

    def __init__(self, params, **kw):
        super().__init__(params, **kw)
        self.records = []
        self.scratch = {id(b): torch.empty_like(b["flat"]) for b in self.buckets}

    def _launch(self, b):
        main = torch.cuda.current_stream()
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=b["flat"].device)
        ready = torch.cuda.Event(enable_timing=True)
        ready.record(main)                                  # the bucket's last gradient kernel has been enqueued on main
        self._stream.wait_event(ready)
        with torch.cuda.stream(self._stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.scratch[id(b)].copy_(b["flat"])
            e1.record()
        self.records.append((ready, e0, e1, b["flat"].numel() * 4))

    def finish(self):
        for b in self.buckets:
            b["pending"] = b["total"]
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    plan = bench.PLANS["P92"]
    model, crit, opt, _ = bench.build(plan, torch.bfloat16, dev)
    model.teacher_backend = "hip"
    eng = model._student_engine()
    red = ProbeReducer(eng.grad_production_order())
    eng.reducer = red
    g = torch.Generator().manual_seed(1000)
    data = torch.randn((8, 3, 1024, 2048), generator=g).to(dev)
    target = torch.randint(0, 19, (8, 1024, 2048), generator=g)
    target[:, :32] = 255
    target = target.to(dev)
    out = {"workload": "bench.kd_step: P92, mode A, 8 x 1024 x 2048, bf16; stand-in collective = copy of each gradient bucket on the reducer's side stream",
           "buckets_mb": [round(b["flat"].numel() * 4 / 2 ** 20, 1) for b in red.buckets], "settings": []}
    for cus in (256, 248, 240, 224, 256):
        _lib.check(_lib.lib().kd_conv_set_persist_cus(0 if cus == 256 else cus), "kd_conv_set_persist_cus")
        for _ in range(3):
            bench.kd_step(model, crit, opt, data, target, "A")
        torch.cuda.synchronize()
        red.records.clear()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        steps = 10
        t0.record()
        for _ in range(steps):
            bench.kd_step(model, crit, opt, data, target, "A")
        t1.record()
        torch.cuda.synchronize()
        wait = [r.elapsed_time(a) for r, a, _, _ in red.records]
        run = [a.elapsed_time(b) for _, a, b, _ in red.records]
        row = {"persist_workgroups": cus, "ms_per_step": t0.elapsed_time(t1) / steps, "buckets_per_step": len(red.records) / steps,
               "wait_ms_mean": sum(wait) / len(wait), "wait_ms_max": max(wait), "wait_ms_per_bucket_of_a_step": [round(w, 3) for w in wait[:len(wait) // steps]],
               "run_ms_mean": sum(run) / len(run), "run_ms_max": max(run)}
        out["settings"].append(row)
        print(json.dumps(row), flush=True)
    _lib.lib().kd_conv_set_persist_cus(0)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
