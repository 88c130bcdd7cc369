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
