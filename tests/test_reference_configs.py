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
