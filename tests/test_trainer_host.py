"""Trainer host logic on CPU (no kernels involved): a real 2-rank run launched the way INTEGRATION.md says
(torch.distributed.run-style environment, nothing initialised by the caller), and checkpoint save -> resume ->
update_optimizer (reference trainer/layerwise_trainer.py:152-173,404-427, base/base_trainer.py:162-185)."""
import json
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp
from torch import nn

from _netutil import trainer_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _TorchKL(nn.Module):
    """Stand-in criterion for CPU runs (the product's criteria are HIP kernels and refuse host tensors)."""

    def forward(self, s, t):
        return nn.functional.kl_div(nn.functional.log_softmax(s, 1), nn.functional.softmax(t, 1), reduction="batchmean")


def _cls_config(save_dir, epochs=2):
    cfg = trainer_config([], lr=0.05, len_epoch=1, save_dir=save_dir, n_gpu=0)
    cfg.update(name="host_cls", teacher={"type": "resnet20", "args": {}}, optimizer={"type": "SGD", "args": {"lr": 0.05}})
    cfg["trainer"].update(name="ClassificationTrainer", epochs=epochs, save_period=1, monitor="min loss", early_stop=10)
    cfg["lr_scheduler"]["args"].update(patience=0, threshold=10.0)   # "no improvement" every epoch: the LR halves
    return cfg


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, save_dir, out):
    try:
        _rank_body(rank, world, port, save_dir, out)
    except BaseException as e:   # surface the failure in the parent instead of a queue timeout
        import traceback
        out.put(dict(rank=rank, error=f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
        raise


def _rank_body(rank, world, port, save_dir, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch.distributed as dist
    import kdcc_amd
    from kdcc_amd import ConfigParser
    from kdcc_amd.models import cifar_models
    from kdcc_amd.models.students import DepthwiseStudent
    from kdcc_amd.trainer import ClassificationTrainer
    from kdcc_amd.utils import WeightScheduler
    from kdcc_amd.utils import optim as optim_module
    torch.set_num_threads(2)
    assert not dist.is_initialized()
    config = ConfigParser(_cls_config(save_dir))          # no run_id: rank 0's timestamp must reach every rank
    assert dist.is_initialized() and dist.get_world_size() == world and dist.get_backend() == "gloo"
    torch.manual_seed(100 + rank)                         # replicas start DIFFERENT: the trainer must broadcast rank 0's student
    teacher = config.init_obj("teacher", cifar_models).eval()
    model = DepthwiseStudent(teacher, config)
    crit = [nn.CrossEntropyLoss(), _TorchKL(), nn.MSELoss()]
    opt = config.init_obj("optimizer", optim_module, model.student.parameters())
    sched = config.init_obj("lr_scheduler", optim_module.lr_scheduler, opt)
    g = torch.Generator().manual_seed(7 + rank)           # each rank its own shard
    batches = [(torch.randn((8, 3, 32, 32), generator=g), torch.randint(0, 10, (8,), generator=g)) for _ in range(2)]
    tr = ClassificationTrainer(model, crit, [], opt, config, batches, None, sched, WeightScheduler(config["weight_scheduler"]))
    assert tr.world_size == world and tr.rank == rank
    tr.train()
    flat = torch.cat([p.detach().reshape(-1) for p in model.student.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    out.put(dict(rank=rank, run_dir=str(config.save_dir), lr=tr.optimizer.param_groups[0]["lr"],
                 same_params=all(torch.equal(gathered[0], t) for t in gathered), best=tr.mnt_best,
                 files=sorted(os.listdir(config.save_dir))))
    dist.destroy_process_group()


def test_two_rank_trainer_gloo(tmp_path):
    """INTEGRATION.md's launch: every rank just builds ConfigParser + trainer.  Checks: the process group comes up by
    itself, all ranks share one run directory (rank 0 writes it), replicas end with bit-identical parameters (rank-0
    broadcast + averaged gradients), the plateau scheduler and the monitor take the same decisions on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for r in res:
        assert "error" not in r, r.get("error")
    res = sorted(res, key=lambda d: d["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = res
    assert a["run_dir"] == b["run_dir"]
    assert a["same_params"] and b["same_params"]
    assert a["lr"] == b["lr"] == 0.05 * 0.25                    # the plateau scheduler cut twice, on both ranks alike
    assert a["best"] == b["best"]                               # monitor decided on the rank mean
    assert "config.json" in a["files"] and "checkpoint-epoch2.pth" in a["files"]
    assert sum(f.startswith("checkpoint-epoch") for f in a["files"]) == 2   # written once (rank 0), not per rank


def test_save_resume_update_optimizer(tmp_path):
    """Incremental plan on a small CIFAR student (3x3 cheap convs at epochs 1 and 2, the second with its own lr):
    epoch 1 rebuilds the optimizer, epoch 2 adds a param group (update_optimizer), the checkpoint carries both copies
    of the network plus the optimizer, and `trainer.resume_path` replays the plan before loading."""
    import kdcc_amd
    from kdcc_amd import ConfigParser
    from kdcc_amd.models import cifar_models
    from kdcc_amd.models.students import DepthwiseSeparableBlock, DepthwiseStudent
    from kdcc_amd.trainer import LayerwiseTrainer
    from kdcc_amd.utils import WeightScheduler
    from kdcc_amd.utils import optim as optim_module

    def build(extra_trainer=None, run_id="a"):
        cfg = trainer_config([], lr=0.005, len_epoch=1, save_dir=str(tmp_path), n_gpu=0)
        cfg.update(name="host_resume", teacher={"type": "resnet20", "args": {}})
        plan = [{"name": "layer1.0.conv1", "epoch": 1}, {"name": "layer2.1.conv2", "epoch": 2, "lr": 0.001}]
        cfg["pruning"] = {"args": {"kernel_size": 3, "padding": 1, "dilation": 1}, "pruning_plan": plan,
                          "hint": [{"name": p["name"], "epoch": p["epoch"]} for p in plan], "unfreeze": plan}
        cfg["trainer"].update(extra_trainer or {})
        config = ConfigParser(cfg, run_id=run_id)
        torch.manual_seed(3)
        teacher = config.init_obj("teacher", cifar_models).eval()
        model = DepthwiseStudent(teacher, config)
        opt = config.init_obj("optimizer", optim_module, model.student.parameters())
        sched = config.init_obj("lr_scheduler", optim_module.lr_scheduler, opt)
        tr = LayerwiseTrainer(model, [nn.CrossEntropyLoss(), _TorchKL(), nn.MSELoss()], [], opt, config, [], None, sched,
                              WeightScheduler(config["weight_scheduler"]))
        return tr, model, opt

    tr, model, opt0 = build()
    tr.prepare_train_epoch(1)
    assert tr.optimizer is not opt0 and len(tr.optimizer.param_groups) == 1          # rebuilt from requires_grad params
    assert isinstance(model.student.layer1[0].conv1, DepthwiseSeparableBlock)
    assert isinstance(model.student.layer2[1].conv2, nn.Conv2d)
    tr.prepare_train_epoch(2)
    assert isinstance(model.student.layer2[1].conv2, DepthwiseSeparableBlock)
    assert [g["lr"] for g in tr.optimizer.param_groups] == [0.005, 0.001]             # update_optimizer: new group, own lr
    assert model.hint_block_names == ["layer2.1.conv2"]                               # a new hint list replaces the old
    with torch.no_grad():
        for p in model.student.parameters():
            if p.requires_grad:
                p.add_(0.25)
    tr._save_checkpoint(2)
    path = tr.checkpoint_dir / "checkpoint-epoch2.pth"
    ck = torch.load(str(path), map_location="cpu", weights_only=False)
    assert set(ck) == {"arch", "epoch", "state_dict", "optimizer", "monitor_best", "config"} and ck["arch"] == "DepthwiseStudent"
    assert "student.layer1.0.conv1.separable_conv.weight" in ck["state_dict"] and "teacher.layer1.0.conv1.weight" in ck["state_dict"]

    tr2, model2, _ = build({"resume_path": str(path)}, run_id="b")
    assert tr2.start_epoch == 3
    assert isinstance(model2.student.layer1[0].conv1, DepthwiseSeparableBlock)
    assert isinstance(model2.student.layer2[1].conv2, DepthwiseSeparableBlock)
    assert [g["lr"] for g in tr2.optimizer.param_groups] == [0.005, 0.001]
    for (n1, p1), (n2, p2) in zip(model.student.named_parameters(), model2.student.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2) and p1.requires_grad == p2.requires_grad, n1
    assert json.load(open(tr2.config.save_dir / "config.json"))["trainer"]["resume_path"] == str(path)


def test_device_batches_lookahead():
    """LayerwiseTrainer._device_batches: with look-ahead (trainer.teacher_overlap = 'backward') every batch is handed out together with
    the NEXT batch's data -- the very tensor object that comes back as `data` one step later, which is how
    DepthwiseStudent.prefetch_teacher recognises it -- and the last one with None; without it, always None."""
    import types
    from kdcc_amd.trainer import LayerwiseTrainer
    self = types.SimpleNamespace(device=torch.device("cpu"))
    loader = [(torch.full((1, 2), float(i)), torch.tensor([i])) for i in range(4)]
    got = list(LayerwiseTrainer._device_batches(self, loader, True))
    assert [int(t.item()) for _, t, _ in got] == [0, 1, 2, 3]
    assert all(got[i][2] is got[i + 1][0] for i in range(3)) and got[3][2] is None
    assert [n for _, _, n in LayerwiseTrainer._device_batches(self, loader, False)] == [None] * 4
    assert list(LayerwiseTrainer._device_batches(self, [], True)) == []


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (the form the driver types) must start two ranks itself and relay
    their exit code.  Without a GPU each rank gets as far as the process group (gloo) and then refuses loudly -- the point here is
    that the parent neither dies with 'launch with torch.distributed.run' nor touches a device."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-sub-records"], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert "starting 2 ranks" in r.stderr, r.stderr[-2000:]
    assert "launch with torch.distributed.run" not in r.stderr
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs an MI355X" in r.stderr, r.stderr[-2000:]
        assert r.stdout.strip() == ""
