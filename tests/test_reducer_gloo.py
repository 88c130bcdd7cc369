"""world_size-2 gloo test of the bucketed gradient reducer (the N>1 path of bench.py / the trainers), on CPU."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import kdcc_amd  # noqa: F401
    from kdcc_amd.parallel import GradReducer
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in [(7, 3), (1000,), (5, 5, 2), (64,)]]
    red = GradReducer(params, bucket_bytes=4 * 1024)       # several buckets
    assert len(red.buckets) >= 2
    for step in range(2):                                  # two steps: buckets re-arm
        for i, p in enumerate(params):                     # "backward": write, then announce, in production order
            red.grad_buffer(p).copy_(torch.full(p.shape, float(rank + 1 + i + 10 * step)))
            red.grad_ready(p)
        red.finish()
        for i, p in enumerate(params):
            expect = sum(r + 1 + i + 10 * step for r in range(world)) / world
            assert torch.allclose(red.grad_buffer(p), torch.full(p.shape, expect)), (rank, i, step)
    out.put(rank)
    dist.destroy_process_group()


def test_bucketed_allreduce_mean_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]
