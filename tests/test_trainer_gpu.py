"""LayerwiseTrainer drop-in test: the same config dict / seeded batches the reference's own LayerwiseTrainer was run on
(tools/make_golden.py: g_trainer_epoch) -> same epoch log and same parameters after 3 RAdam steps (fp32 parity mode)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _netutil import trainer_config  # noqa: E402
from _seeded import sample_idx, seeded_fill_, seeded_input  # noqa: E402


@pytest.mark.parametrize("teacher_overlap", ["none", "backward"])
def test_layerwise_trainer_epoch_matches_reference(golden, tmp_path, teacher_overlap):
    """teacher_overlap = "backward": the same epoch with the teacher's forward for batch i + 1 launched on the side stream before
    batch i's loss.backward() (trainer.teacher_overlap, DepthwiseStudent.prefetch_teacher) -- same reference golden."""
    import kdcc_amd
    from kdcc_amd import ConfigParser, losses, models
    from kdcc_amd.models.students import DepthwiseStudent
    from kdcc_amd.trainer import LayerwiseTrainer
    from kdcc_amd.utils import WeightScheduler
    from kdcc_amd.utils import optim as optim_module

    g = golden("trainer_epoch_g4")
    plan = [str(s) for s in g["plan"]]
    cfg = trainer_config(plan, lr=float(g["lr"]), len_epoch=2, save_dir=str(tmp_path))
    cfg["trainer"]["teacher_overlap"] = teacher_overlap
    config = ConfigParser(cfg, run_id="t")
    teacher = config.init_obj("teacher", models)          # resolves DeepWV3Plus by name, like train.py:35
    seeded_fill_(teacher, "teacher.")
    teacher.eval()
    model = DepthwiseStudent(teacher, config)
    assert model.dtype == torch.float32                   # trainer.dtype = fp32 -> parity mode
    orig_replace = model.replace

    def replace_and_seed(blocks, **kw):
        orig_replace(blocks, **kw)
        for b in blocks:
            seeded_fill_(model.get_block(b["name"], model.student), f"student.{b['name']}.")
    model.replace = replace_and_seed
    crit = [config.init_obj(k, losses) for k in ("supervised_loss", "kd_loss", "hint_loss")]
    opt = config.init_obj("optimizer", optim_module, model.student.parameters())
    sched = config.init_obj("lr_scheduler", optim_module.lr_scheduler, opt)
    batches = [(seeded_input(f"trainer.x{i}", (2, 3, 64, 128)), torch.from_numpy(g["targets"][i].astype(np.int64)))
               for i in range(3)]
    tr = LayerwiseTrainer(model, crit, [], opt, config, batches, None, sched, WeightScheduler(config["weight_scheduler"]))
    log = tr._train_epoch(1)
    assert getattr(model, "prefetch_hits", 0) == (2 if teacher_overlap == "backward" else 0)     # batches 1 and 2 of the three
    for k in ("loss", "supervised_loss", "kd_loss", "hint_loss", "teacher_loss"):
        np.testing.assert_allclose(log[k], float(g["log:" + k]), rtol=1e-3, err_msg=k)
    for k in ("train_teacher_mIoU", "train_student_mIoU"):
        np.testing.assert_allclose(log[k], float(g["log:" + k]), rtol=2e-2, atol=1e-4, err_msg=k)   # argmax ties on random nets
    assert isinstance(tr.optimizer, optim_module.RAdam) and tr.optimizer is not opt   # rebuilt at epoch 1
    n_train = 0
    for n, p in model.student.named_parameters():
        if p.requires_grad:
            n_train += 1
            f = p.detach().float().contiguous().reshape(-1).cpu()
            ref = g[f"param:{n}.sample"].astype(np.float64)
            got = f[sample_idx(f.numel())].numpy().astype(np.float64)
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-3, n
    assert n_train == 8


def test_sliding_window_test_epoch(tmp_path):
    """LayerwiseTrainer.test(): sliding-window + flip inference.  The tiling equals the reference's get_crops_image boxes;
    on an image that is exactly one window wide the result is the mean of the plain and the mirrored forward; overlapping
    windows are count-normalised; the epoch returns supervised_loss / mIoU and writes the submission PNG."""
    import kdcc_amd
    from kdcc_amd import ConfigParser, losses, models
    from kdcc_amd.models.students import DepthwiseStudent
    from kdcc_amd.trainer import LayerwiseTrainer
    from kdcc_amd.utils import WeightScheduler
    from kdcc_amd.utils import optim as optim_module
    # reference tiling (utils/tta_process.py:83-101) for a 1024x2048 frame, crop 1024, overlap 1/3
    assert DepthwiseStudent.sliding_windows(1024, 2048, 1024) == [(0, 0, 1024, 1024), (683, 0, 1707, 1024), (1024, 0, 2048, 1024)]
    cfg = trainer_config([], lr=1e-3, len_epoch=1, save_dir=str(tmp_path))
    cfg["test"] = {"type": "sliding", "args": {"scales": [1.0], "crop_size": 64}}
    cfg["submission"] = {"save_output": True, "path_output": str(tmp_path / "submission"), "ext": "png"}
    config = ConfigParser(cfg, run_id="test")
    teacher = config.init_obj("teacher", models)
    seeded_fill_(teacher, "teacher.")
    teacher.eval()
    model = DepthwiseStudent(teacher, config).cuda()
    x = seeded_input("tta.x", (1, 3, 64, 160)).cuda()
    args = cfg["test"]["args"]
    out = model.inference_test(x, dict(args, window_count="pixel"))      # (the reference's count indexing: tests/test_host_logic.py)
    assert tuple(out.shape) == (1, 19, 64, 160)
    boxes = model.sliding_windows(64, 160, 64)
    assert boxes == [(0, 0, 64, 64), (43, 0, 107, 64), (86, 0, 150, 64), (96, 0, 160, 64)]
    # hand-built expectation from plain forwards of the same windows
    def stitched(src):
        full = torch.zeros((19, 64, 160), device="cuda"); cnt = torch.zeros((1, 64, 160), device="cuda")
        for (x1, y1, x2, y2) in boxes:
            full[:, y1:y2, x1:x2] += model.inference(src[:, :, y1:y2, x1:x2].contiguous())[0].float(); cnt[:, y1:y2, x1:x2] += 1
        return full / cnt
    exp = (stitched(x) + torch.flip(stitched(torch.flip(x, dims=[3])), dims=[2])) / 2
    assert torch.allclose(out[0], exp, rtol=1e-4, atol=1e-5)
    crit = [config.init_obj(k, losses) for k in ("supervised_loss", "kd_loss", "hint_loss")]
    opt = config.init_obj("optimizer", optim_module, model.student.parameters())
    tgt = torch.randint(0, 19, (1, 64, 160), generator=torch.Generator().manual_seed(3))
    loader = [(["frankfurt_000000"], x.cpu(), tgt)]
    tr = LayerwiseTrainer(model, crit, [], opt, config, [], loader, None, WeightScheduler(config["weight_scheduler"]))
    res = tr._test_epoch(1)
    assert res["supervised_loss"] > 0 and 0.0 <= res["mIoU"] <= 1.0
    assert (tmp_path / "submission" / "frankfurt_000000.png").exists()
    # multi-scale passes (utils/tta_process.py:53-66: resampled image, tile = int(scale * crop)): the mean over scales; a scale
    # listed twice changes nothing, a second scale does
    ref1 = model.inference_test(x, {"scales": [1.0], "crop_size": 64})
    assert torch.allclose(model.inference_test(x, {"scales": [1.0, 1.0], "crop_size": 64}), ref1, rtol=1e-5, atol=1e-6)
    ms = model.inference_test(x, {"scales": [0.5, 1.0, 1.5], "crop_size": 64})
    assert tuple(ms.shape) == (1, 19, 64, 160) and torch.isfinite(ms).all() and not torch.allclose(ms, ref1, rtol=1e-3, atol=1e-3)


def test_gscnn_config_is_drop_in(tmp_path):
    """BASELINE config 5's shape through the reference's own plug points: `teacher.type = "GSCNN"` resolved by init_obj,
    DepthwiseStudent + LayerwiseTrainer on the shipped GSCNN plan's layer names, bf16 (the measured dtype): one epoch of two
    steps runs, the hint loss is finite and positive, the cheap-conv parameters move, the shape stream stays frozen."""
    import kdcc_amd
    from kdcc_amd import ConfigParser, losses, models
    from kdcc_amd.models.students import DepthwiseStudent
    from kdcc_amd.trainer import LayerwiseTrainer
    from kdcc_amd.utils import WeightScheduler
    from kdcc_amd.utils import optim as optim_module
    plan = ["mod4.block2.convs.conv2", "mod7.block1.convs.conv2", "aspp.features.2.0"]
    cfg = trainer_config(plan, lr=1e-3, len_epoch=1, save_dir=str(tmp_path), dtype="bf16")
    cfg["teacher"] = {"type": "GSCNN", "args": {"num_classes": 19}}
    config = ConfigParser(cfg, run_id="gscnn")
    teacher = config.init_obj("teacher", models)
    assert type(teacher).__name__ == "GSCNN"
    seeded_fill_(teacher, "gscnn.")
    teacher.eval()
    model = DepthwiseStudent(teacher, config)
    assert model.fused and model.dtype == torch.bfloat16
    crit = [config.init_obj(k, losses) for k in ("supervised_loss", "kd_loss", "hint_loss")]
    opt = config.init_obj("optimizer", optim_module, model.student.parameters())
    sched = config.init_obj("lr_scheduler", optim_module.lr_scheduler, opt)
    batches = [(seeded_input(f"gs.x{i}", (1, 3, 64, 256), scale=30.0), torch.randint(0, 19, (1, 64, 256), generator=torch.Generator().manual_seed(i)))
               for i in range(2)]
    tr = LayerwiseTrainer(model, crit, [], opt, config, batches, None, sched, WeightScheduler(config["weight_scheduler"]))
    before = {n: p.detach().clone() for n, p in model.student.named_parameters() if "gate1" in n or "res1" in n}
    log = tr._train_epoch(1)
    assert np.isfinite(log["hint_loss"]) and log["hint_loss"] > 0 and np.isfinite(log["supervised_loss"])
    moved = [n for n, p in model.student.named_parameters() if p.requires_grad]
    assert len(moved) == 6 and all("separable_conv" in n or "pointwise_conv" in n for n in moved)
    for n, p in model.student.named_parameters():
        if n in before:
            assert torch.equal(p, before[n].to(p.device)), n


def test_gscnn_mode_b_through_the_trainer(tmp_path):
    """`trainer.backprop = "kd+hint"` on a GSCNN config with `pruning.unfreeze` naming shape-stream modules: LayerwiseTrainer tells
    the model that the logits carry gradient (model.logits_need_grad), the engine differentiates the shape stream, the named
    shape-stream parameters move and stay finite; the KD term is in the logged loss."""
    import kdcc_amd
    from kdcc_amd import ConfigParser, losses, models
    from kdcc_amd.models.students import DepthwiseStudent
    from kdcc_amd.trainer import LayerwiseTrainer
    from kdcc_amd.utils import WeightScheduler
    from kdcc_amd.utils import optim as optim_module
    plan = ["mod4.block2.convs.conv2", "aspp.features.2.0"]
    cfg = trainer_config(plan, lr=1e-4, len_epoch=1, save_dir=str(tmp_path))
    cfg["teacher"] = {"type": "GSCNN", "args": {"num_classes": 19}}
    cfg["kd_loss"] = {"type": "KLDivergenceLoss", "args": {"temperature": 1}}
    cfg["trainer"]["backprop"] = "kd+hint"
    cfg["pruning"]["unfreeze"] = cfg["pruning"]["unfreeze"] + [{"name": n, "epoch": 1} for n in ("gate2", "d3", "res3", "dsn4", "fuse", "aspp.edge_conv")]
    cfg["pruning"]["hint"] = cfg["pruning"]["hint"] + [{"name": "aspp", "epoch": 1}]
    config = ConfigParser(cfg, run_id="gscnn_b")
    teacher = config.init_obj("teacher", models)
    seeded_fill_(teacher, "gscnn.")
    teacher.eval()
    model = DepthwiseStudent(teacher, config)
    crit = [config.init_obj(k, losses) for k in ("supervised_loss", "kd_loss", "hint_loss")]
    opt = config.init_obj("optimizer", optim_module, model.student.parameters())
    sched = config.init_obj("lr_scheduler", optim_module.lr_scheduler, opt)
    batches = [(seeded_input(f"gsb.x{i}", (1, 3, 64, 128), scale=30.0), torch.randint(0, 19, (1, 64, 128), generator=torch.Generator().manual_seed(i)))
               for i in range(2)]
    tr = LayerwiseTrainer(model, crit, [], opt, config, batches, None, sched, WeightScheduler(config["weight_scheduler"]))
    assert model.logits_need_grad is True
    watched = {n: p.detach().clone() for n, p in model.student.named_parameters()
               if n.split(".")[0] in ("gate2", "d3", "res3", "dsn4", "fuse") or n.startswith("aspp.edge_conv")}
    log = tr._train_epoch(1)
    assert np.isfinite(log["loss"]) and log["kd_loss"] > 0 and abs(log["loss"] - log["kd_loss"] - log["hint_loss"]) < 1e-3 * abs(log["loss"])
    assert model.student_hint_names[-1] == "aspp"
    # the shape stream's gradients are ~1e-6 .. 1e-9 at this init (lr 1e-4 moves the weights by less than an ulp), so look at what the
    # optimizer received: a first-moment estimate for every named parameter, finite and not identically zero
    reached = 0
    for n, p in model.student.named_parameters():
        if n in watched:
            assert p.requires_grad and torch.isfinite(p).all(), n
            st = tr.optimizer.state.get(p, {})
            assert "exp_avg" in st and torch.isfinite(st["exp_avg"]).all(), n
            reached += int(float(st["exp_avg"].abs().max()) > 0)
    assert reached >= len(watched) - 1, (reached, len(watched))
