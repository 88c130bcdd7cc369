"""Exercises the data-parallel plumbing of the fused student on ONE GPU: a 1-rank RCCL ("nccl") process group with the
collective forced on, so the bucket buffers, the side-stream all-reduce launched from inside backward and the stream
hand-off back to the optimizer all run exactly as on the 8-GPU node (where each rank does the same with world_size 8)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from _seeded import seeded_fill_, seeded_input  # noqa: E402


def test_reducer_inside_backward_single_rank_nccl():
    import kdcc_amd
    from kdcc_amd import losses, parallel
    from kdcc_amd.models import DeepWV3Plus
    from kdcc_amd.models.students import DepthwiseStudent
    from kdcc_amd.utils.optim import RAdam
    plan = ["mod4.block2.convs.conv2", "mod7.block1.convs.conv2", "aspp.features.2.0"]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 1000), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        def run(with_reducer):
            teacher = DeepWV3Plus(19)
            seeded_fill_(teacher, "teacher.")
            teacher.eval()
            model = DepthwiseStudent(teacher, None, dtype=torch.bfloat16)
            model.replace([{"name": n, "epoch": 1} for n in plan], kernel_size=9, padding=20, dilation=5)
            model.register_hint_layers(plan)
            model.unfreeze(plan)
            for n in plan:
                seeded_fill_(model.get_block(n, model.student), f"student.{n}.")
            model.cuda()
            opt = RAdam([p for p in model.student.parameters() if p.requires_grad], lr=1e-3)
            if with_reducer:
                eng = model._student_engine()
                red = parallel.GradReducer(eng.grad_production_order(), bucket_bytes=4 << 20)
                red.force_collective = True
                eng.reducer = red
                assert len(red.buckets) >= 2
            crit = losses.MSELoss(num_classes=1000)
            x = seeded_input("ddp.x", (1, 3, 64, 128)).cuda()
            vals = []
            for _ in range(3):   # several steps: buckets re-arm, versions bump, packs refresh
                model(x)
                loss = 0
                for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
                    loss = loss + crit(s, t)
                loss.backward()
                opt.step()
                opt.zero_grad()
                vals.append(loss.item())
            torch.cuda.synchronize()
            return vals, {n: p.detach().float().cpu().numpy() for n, p in model.student.named_parameters() if p.requires_grad}
        v0, p0 = run(False)
        v1, p1 = run(True)
        np.testing.assert_allclose(v1, v0, rtol=1e-5)           # mean over one rank == identity
        for n in p0:
            np.testing.assert_allclose(p1[n], p0[n], rtol=1e-5, atol=1e-7, err_msg=n)
        assert v0[-1] < v0[0]                                    # and it trains
    finally:
        dist.destroy_process_group()


def test_reducer_mode_b_buckets_complete_before_the_optimizer_single_rank_nccl():
    """Mode B (every student parameter trainable, loss = KLDiv + hints): 370-440 MB of fp32 gradients in 29-46 buckets of >= 8 MB,
    each all-reduced on the side stream from inside backward (1-rank RCCL group, collective forced on).  When backward returns
    every bucket has been produced and handed back (reducer.finish ran: no pending work, counters re-armed), the main stream
    has been made to wait for the side stream, and the step equals the one without a reducer."""
    import kdcc_amd
    from kdcc_amd import losses, parallel
    from kdcc_amd.models import DeepWV3Plus
    from kdcc_amd.models.students import DepthwiseStudent
    from kdcc_amd.utils.optim import RAdam
    plan = ["mod4.block2.convs.conv2", "mod7.block1.convs.conv2", "aspp.features.2.0"]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + (os.getpid() + 7) % 1000), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        def run(with_reducer):
            teacher = DeepWV3Plus(19)
            seeded_fill_(teacher, "teacher.")
            teacher.eval()
            model = DepthwiseStudent(teacher, None, dtype=torch.bfloat16)
            model.replace([{"name": n, "epoch": 1} for n in plan], kernel_size=9, padding=20, dilation=5)
            model.register_hint_layers(plan)
            for n in plan:
                seeded_fill_(model.get_block(n, model.student), f"student.{n}.")
            for p in model.student.parameters():
                p.requires_grad = True
            model.logits_need_grad = True
            model.cuda()
            params = [p for p in model.student.parameters() if p.requires_grad]
            opt = RAdam(params, lr=1e-4)
            red = None
            if with_reducer:
                eng = model._student_engine()
                red = parallel.GradReducer(eng.grad_production_order())
                red.force_collective = True
                eng.reducer = red
                nbytes = sum(b["flat"].numel() * 4 for b in red.buckets)
                assert len(red.buckets) >= 20 and nbytes > 300e6, (len(red.buckets), nbytes)       # 29 buckets, 441 MB with this plan
                assert sum(len(b["views"]) for b in red.buckets) == len(params)
            mse, kld = losses.MSELoss(num_classes=1000), losses.KLDivergenceLoss(1)
            x = seeded_input("ddp.b.x", (1, 3, 64, 128)).cuda()
            vals = []
            for _ in range(2):
                out_st, out_tc = model(x)
                loss = kld(out_st, out_tc)
                for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
                    loss = loss + mse(s, t)
                loss.backward()
                if red is not None:
                    # every bucket was produced, exchanged and handed back before the optimizer reads a gradient
                    assert all(b["work"] is None and b["pending"] == b["total"] for b in red.buckets)
                    torch.cuda.current_stream().synchronize()          # (the main stream waits for the side stream: finish())
                    assert red._stream is not None and red._stream.query()
                    for p in params:
                        assert p.grad is not None and torch.equal(p.grad, red.grad_buffer(p))      # what the optimizer reads is the exchanged bucket
                opt.step()
                opt.zero_grad()
                vals.append(loss.item())
            torch.cuda.synchronize()
            return vals, {n: p.detach().float().cpu().numpy() for n, p in model.student.named_parameters()}
        v0, p0 = run(False)
        v1, p1 = run(True)
        np.testing.assert_allclose(v1, v0, rtol=1e-5)
        for n in p0:
            np.testing.assert_allclose(p1[n], p0[n], rtol=1e-5, atol=1e-7, err_msg=n)
    finally:
        dist.destroy_process_group()


def test_persistent_conv_kernels_on_fewer_cus_are_bit_identical():
    """KDCC_PERSIST_CUS=248 (the persistent conv kernels leave 8 CUs to a concurrent RCCL kernel): a tile's arithmetic does not
    depend on which workgroup computes it -- the outputs of the 3x3 lone-wave kernel, the 1x1 ping-pong kernel and the 512 x 128
    lone-wave kernel are bit-identical to the 256-workgroup launch (child processes: the switch is read once)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import hashlib, json, sys, torch
sys.path.insert(0, %r)
import kdcc_amd
from kdcc_amd import _lib, ops
out = {}
for name, H, W, Cin, Cout, k, d in (("row", 96, 512, 128, 512, 3, 2), ("pw", 160, 512, 128, 512, 1, 1), ("pp128", 64, 1024, 64, 128, 3, 1)):
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(2, H, W, Cin, device="cuda", generator=g).bfloat16()
    w = (torch.randn(Cout, k, k, Cin, device="cuda", generator=g) * 0.05).bfloat16()
    y = torch.zeros(2, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    with _lib.kernel_log() as log:
        ops.conv2d(x, w, 1, d * (k // 2), d, out_raw=y)
    torch.cuda.synchronize()
    out[name] = [hashlib.sha256(y.cpu().view(torch.int16).numpy().tobytes()).hexdigest(), sorted(log.counts)]
print("RESULT " + json.dumps(out))
''' % root
    res = {}
    for cus in ("256", "248"):
        env = dict(os.environ, KDCC_PERSIST_CUS=cus)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and line, (r.stdout[-1000:], r.stderr[-2000:])
        res[cus] = json.loads(line[0][7:])
    assert res["256"]["row"][1] == ["conv_row_lw_kernel"] and res["256"]["pw"][1] == ["conv_igemm_persist_kernel<pp>"] and \
        res["256"]["pp128"][1] == ["conv_row_tall_kernel"], res["256"]
    assert res["248"] == res["256"]


def test_persist_cus_set_at_run_time_is_bit_identical_and_reducer_sets_no_cap():
    """kd_conv_set_persist_cus (the run-time form of KDCC_PERSIST_CUS, what parallel.GradReducer calls for world > 1): the K-concatenated
    1x1 kernel and the 3x3 lone-wave kernel give the same bits on 256, 248 and 64 workgroups, in one process; GradReducer's measured
    default is no cap (profiles/r05_sidestream.json)."""
    import kdcc_amd
    from kdcc_amd import _lib, ops, parallel
    g = torch.Generator(device="cuda").manual_seed(21)
    x1 = torch.randn(1, 160, 512, 64, device="cuda", generator=g).bfloat16()
    x2 = torch.randn(1, 160, 512, 128, device="cuda", generator=g).relu().bfloat16()
    wc = (torch.randn(256, 1, 1, 192, device="cuda", generator=g) * 0.07).bfloat16()
    x3 = torch.randn(2, 96, 512, 128, device="cuda", generator=g).bfloat16()
    w3 = (torch.randn(512, 3, 3, 128, device="cuda", generator=g) * 0.05).bfloat16()
    outs = {}
    try:
        for cus in (0, 248, 64):
            _lib.check(_lib.lib().kd_conv_set_persist_cus(cus), "kd_conv_set_persist_cus")
            ya = torch.zeros(1, 160, 512, 256, device="cuda", dtype=torch.bfloat16)
            yb = torch.zeros(2, 96, 512, 512, device="cuda", dtype=torch.bfloat16)
            with _lib.kernel_log() as log:
                ops.conv2d(x1, wc, x2=x2, out_raw=ya)
                ops.conv2d(x3, w3, 1, 2, 2, out_raw=yb)
            assert sorted(k for k, v in log.counts.items() if v) == ["conv_igemm_persist_kernel<pp,dual>", "conv_row_lw_kernel"], log.counts
            outs[cus] = (ya.clone(), yb.clone())
    finally:
        _lib.lib().kd_conv_set_persist_cus(0)
    for cus in (248, 64):
        assert torch.equal(outs[cus][0], outs[0][0]) and torch.equal(outs[cus][1], outs[0][1]), cus
    assert _lib.lib().kd_conv_set_persist_cus(-1) != 0           # refused, not clamped
    assert parallel.GradReducer.CONV_GRID_CAP == 0
