"""Taylor filter importance on the GPU: engine probes + supervised-loss backward vs the reference's own
TaylorPruneStudent run (tests/golden/taylor.npz), the CrossEntropyLoss2d gradient kernel, and a TaylorPruneTrainer epoch
whose dumped table drives WeightedHintMSELoss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _netutil import trainer_config  # noqa: E402
from _seeded import seeded_fill_, seeded_input  # noqa: E402


def test_ce2d_gradient_matches_torch(golden):
    import torch.nn.functional as F
    from kdcc_amd import losses
    g = golden("losses")
    x, t = torch.from_numpy(g["ce.x"]), torch.from_numpy(g["ce.target"])
    xr = x.clone().requires_grad_(True)
    (F.cross_entropy(xr, t, ignore_index=255) * 3.0).backward()
    for fmt in (torch.contiguous_format, torch.channels_last):
        xd = x.cuda().contiguous(memory_format=fmt).requires_grad_(True)
        loss = losses.CrossEntropyLoss2d(ignore_index=255)(xd, t.cuda())
        np.testing.assert_allclose(loss.item(), float(g["ce.loss"]), rtol=1e-5)
        (loss * 3.0).backward()
        np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-8)


def _taylor_model(dtype):
    import kdcc_amd
    from kdcc_amd.models import DeepWV3Plus
    from kdcc_amd.models.students import TaylorPruneStudent
    teacher = DeepWV3Plus(num_classes=19)
    seeded_fill_(teacher, "teacher.")
    teacher.eval()
    return TaylorPruneStudent(teacher, None, dtype=dtype).cuda()


def test_gate_importance_matches_reference(golden):
    from kdcc_amd import losses
    g = golden("taylor")
    names = [str(s) for s in g["names"]]
    model = _taylor_model(torch.float32)
    model.replace([{"name": n, "epoch": 1, "num_features": len(g[f"imp:{n}"])} for n in names])
    x = seeded_input("taylor.x", (2, 3, 64, 128)).cuda()
    tgt = torch.from_numpy(g["target"].astype(np.int64)).cuda()
    out_st, _ = model(x)
    loss = losses.CrossEntropyLoss2d(ignore_index=255)(out_st, tgt)
    loss.backward()
    imp = model.get_gate_importance()
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-3)
    for n in names:
        ref = g[f"gate_grad:{n}"].astype(np.float64)
        got = model.added_gates[n].weight.grad.cpu().numpy().astype(np.float64)
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-3, n
        assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max(), n
        ri = g[f"imp:{n}"]
        assert np.abs(imp[n] - ri).max() < 4e-3 * ri.max(), n


def test_taylor_trainer_feeds_weighted_hint_loss(tmp_path):
    """TaylorPruneTrainer epoch (frozen student, supervised loss, importance dump) -> LayerwiseTrainer with
    WeightedHintMSELoss reading that table as filter weights."""
    import kdcc_amd
    from kdcc_amd import ConfigParser, losses, models
    from kdcc_amd.models.students import DepthwiseStudent, TaylorPruneStudent
    from kdcc_amd.trainer import LayerwiseTrainer, TaylorPruneTrainer
    from kdcc_amd.utils import WeightScheduler
    from kdcc_amd.utils import optim as optim_module
    gates = [("mod4.block2.convs.conv2", 512), ("aspp.features.1.0", 256)]
    cfg = trainer_config([], lr=1e-3, len_epoch=1, save_dir=str(tmp_path))
    ent = [{"name": n, "epoch": 1, "num_features": c} for n, c in gates]
    cfg["pruning"].update(pruning_plan=ent, hint=[], unfreeze=[])
    cfg["trainer"].update(name="TaylorPruneTrainer", importance_log_interval=1)
    config = ConfigParser(cfg, run_id="taylor")
    teacher = config.init_obj("teacher", models)
    seeded_fill_(teacher, "teacher.")
    teacher.eval()
    model = TaylorPruneStudent(teacher, config)
    crit = [config.init_obj(k, losses) for k in ("supervised_loss", "kd_loss", "hint_loss")]
    batches = [(seeded_input(f"ttr.x{i}", (1, 3, 64, 128)), torch.randint(0, 19, (1, 64, 128), generator=torch.Generator().manual_seed(i)))
               for i in range(2)]
    opt = optim_module.RAdam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    tr = TaylorPruneTrainer(model, crit, [], opt, config, batches, None, None, WeightScheduler(config["weight_scheduler"]))
    log = tr._train_epoch(1)
    assert tr.optimizer is None and log["supervised_loss"] > 0
    path = tr.checkpoint_dir / "importance_filter_ep1_batch_idx1.pth"
    table = torch.load(str(path))
    for n, c in gates:
        assert table[n].shape == (c,) and abs(float(table[n].sum()) - 1.0) < 1e-4 and bool((table[n] >= 0).all())
    # consume it: hint KD on those two layers with the importance as WeightedHintMSELoss filter weights
    plan = [n for n, _ in gates]
    cfg2 = trainer_config(plan, lr=1e-3, len_epoch=0, save_dir=str(tmp_path))
    cfg2["hint_loss"] = {"type": "WeightedHintMSELoss", "args": {}}
    cfg2["trainer"]["hint_filter_weight"] = str(path)
    config2 = ConfigParser(cfg2, run_id="whint")
    model2 = DepthwiseStudent(teacher, config2)
    crit2 = [config2.init_obj(k, losses) for k in ("supervised_loss", "kd_loss", "hint_loss")]
    opt2 = config2.init_obj("optimizer", optim_module, model2.student.parameters())
    tr2 = LayerwiseTrainer(model2, crit2, [], opt2, config2, batches, None, None, WeightScheduler(config2["weight_scheduler"]))
    log2 = tr2._train_epoch(1)
    assert np.isfinite(log2["hint_loss"]) and log2["hint_loss"] > 0
    w = tr2._filter_weight(0, 512, torch.device("cuda"))
    assert torch.allclose(w.cpu(), table[plan[0]], atol=1e-7)
