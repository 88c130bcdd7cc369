"""Every shipped Cityscapes config of the reference (cfg/cityscapes/*.json) is a drop-in at the plan level: teacher type
resolves, every pruning-plan / hint / unfreeze name of every epoch applies to the fused student, parameter counts match the
README rows.  Reads the reference's JSON files as data when /root/reference is present (the build container); skipped on
boxes without it.  CPU only (meta device)."""
import glob
import json
import os

import pytest
import torch

import kdcc_amd  # noqa: F401
from kdcc_amd import models
from kdcc_amd.models.students import DepthwiseStudent

CFG_DIR = "/root/reference/cfg/cityscapes"
CONFIGS = sorted(glob.glob(os.path.join(CFG_DIR, "*.json")))
EXPECTED = {"58M_deeplab_all.json": 79752256, "51M_deeplab_all.json": 85960768, "51M_gscnn_all.json": 86135022,
            "51M_deeplab_incremental.json": 85960768}


@pytest.mark.skipif(not CONFIGS, reason="reference configs not present on this box")
@pytest.mark.parametrize("path", CONFIGS, ids=[os.path.basename(p) for p in CONFIGS])
def test_shipped_config_plan_applies(path):
    cfg = json.load(open(path))
    ttype = cfg["teacher"]["type"]
    if not hasattr(models, ttype):
        pytest.skip(f"teacher {ttype} is outside the scope contract (SURVEY section 2)")
    with torch.device("meta"):
        teacher = getattr(models, ttype)(**cfg["teacher"]["args"])
        model = DepthwiseStudent(teacher, None)
    pr = cfg["pruning"]
    epochs = sorted({e["epoch"] for k in ("pruning_plan", "hint", "unfreeze") for e in pr[k]})
    for ep in epochs:   # what LayerwiseTrainer.prepare_train_epoch does at each scheduled epoch
        with torch.device("meta"):
            model.replace([e for e in pr["pruning_plan"] if e["epoch"] == ep], **pr["args"])
        model.register_hint_layers([e["name"] for e in pr["hint"] if e["epoch"] == ep])     # validated against the fused graph
        model.unfreeze([e["name"] for e in pr["unfreeze"] if e["epoch"] == ep])
    assert model.fused
    name = os.path.basename(path)
    if name in EXPECTED:
        assert sum(p.numel() for p in model.student.parameters()) == EXPECTED[name]
    trainable = [n for n, p in model.student.named_parameters() if p.requires_grad]
    assert trainable and all("separable_conv" in n or "pointwise_conv" in n for n in trainable)
    for key in ("supervised_loss", "kd_loss", "hint_loss"):
        assert hasattr(kdcc_amd.losses, cfg[key]["type"]), cfg[key]["type"]
    assert hasattr(kdcc_amd.utils.optim, cfg["optimizer"]["type"])
    assert hasattr(kdcc_amd.utils.optim.lr_scheduler, cfg["lr_scheduler"]["type"])
    assert hasattr(kdcc_amd.trainer, cfg["trainer"]["name"])


TAYLOR_CFG = "/root/reference/cfg/taylor_importance_track.json"


@pytest.mark.skipif(not os.path.exists(TAYLOR_CFG), reason="reference configs not present on this box")
def test_shipped_taylor_config_plan_applies():
    """cfg/taylor_importance_track.json -- the only Taylor config the reference ships: `pruning` has just {args, pruning_plan},
    and the plan gates block convs, bnrelu ReLUs (`...bn2.1`, `...bn3.1`) and ASPP branch ReLUs (`aspp.features.N.2`).  Every
    entry of every epoch must apply to the fused student and be seen by the engine as a gate at a supported site."""
    from kdcc_amd.engine import StudentEngine, _Site, _act_gate, _gate_split
    from kdcc_amd.models.students import TaylorPruneStudent
    cfg = json.load(open(TAYLOR_CFG))
    assert set(cfg["pruning"]) == {"args", "pruning_plan"} and cfg["trainer"]["name"] == "TaylorPruneTrainer"
    with torch.device("meta"):
        teacher = getattr(models, cfg["teacher"]["type"])(**cfg["teacher"]["args"])
        model = TaylorPruneStudent(teacher, None)
    plan = cfg["pruning"]["pruning_plan"]
    for ep in sorted({e["epoch"] for e in plan}):
        with torch.device("meta"):
            model.replace([e for e in plan if e["epoch"] == ep], **cfg["pruning"]["args"])
    assert sorted(model.added_gates) == sorted(e["name"] for e in plan)
    trainable = [n for n, p in model.student.named_parameters() if p.requires_grad]
    assert len(trainable) == len(plan) and all(n.endswith(".1.weight") for n in trainable)     # the gates, nothing else
    # the engine finds each gate where it will fold it
    eng = StudentEngine(model.student, torch.bfloat16)
    seen = 0
    for name, blk in eng._flat_blocks():
        for n, m in blk.convs.named_children():
            if n.startswith("conv"):
                seen += _Site(f"{name}.convs.{n}", m).gate is not None
            elif n.startswith("bn"):
                seen += _act_gate(m) is not None
    for br in model.student.aspp.features:
        seen += _gate_split(br[0])[1] is not None
        seen += len(br) > 2 and _gate_split(br[2])[1] is not None
    assert seen == len(plan)
    order = eng.grad_production_order()
    assert len(order) == len(plan) and {id(p) for p in order} == {id(g.weight) for g in model.added_gates.values()}
    # a site the fused graph cannot fold a gate into is refused when the plan is applied
    with torch.device("meta"), pytest.raises(ValueError):
        model.replace([{"name": "mod4.block2.bn1.1", "epoch": 1, "num_features": 256}])
