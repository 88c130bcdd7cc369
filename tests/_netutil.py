"""Seeded teacher / student state dicts (by checkpoint key) for the network-level oracle and the GPU tests."""
import json
import os

import torch

from _seeded import seeded_value

_HERE = os.path.dirname(os.path.abspath(__file__))


def key_inventory():
    with open(os.path.join(_HERE, "golden", "deepwv3plus_keys.json")) as f:
        return json.load(f)


def seeded_teacher_sd(dtype=torch.float32):
    sd = {}
    for k, shape in key_inventory().items():
        if k.endswith("num_batches_tracked"):
            continue
        sd[k] = seeded_value("teacher." + k, torch.empty(shape)).to(dtype)
    return sd


def seeded_cheap_weights(teacher_sd, plan, k=9, dtype=torch.float32):
    out = {}
    for n in plan:
        cout, cin = teacher_sd[n + ".weight"].shape[:2]
        out[f"{n}.separable_conv.weight"] = seeded_value(f"student.{n}.separable_conv.weight", torch.empty(cin, 1, k, k)).to(dtype)
        out[f"{n}.pointwise_conv.weight"] = seeded_value(f"student.{n}.pointwise_conv.weight", torch.empty(cout, cin, 1, 1)).to(dtype)
    return out


def trainer_config(plan, lr, len_epoch, save_dir, n_gpu=1, dtype="fp32"):
    """A config dict in the reference's JSON schema (cfg/cityscapes/*.json) for a tiny synthetic run
    (the same dict tools/make_golden.py feeds the reference's ConfigParser / LayerwiseTrainer)."""
    ent = [{"name": n, "epoch": 1} for n in plan]
    return {
        "name": "golden_trainer", "n_gpu": n_gpu,
        "teacher": {"type": "DeepWV3Plus", "args": {"num_classes": 19}},
        "optimizer": {"type": "RAdam", "args": {"lr": lr}},
        "supervised_loss": {"type": "CrossEntropyLoss2d", "args": {"ignore_index": 255}},
        "kd_loss": {"type": "MSELoss", "args": {"reduction": "mean", "num_classes": 1}},
        "hint_loss": {"type": "MSELoss", "args": {"reduction": "mean", "num_classes": 1000}},
        "metrics": [],
        "lr_scheduler": {"type": "MyReduceLROnPlateau", "args": {"mode": "min", "threshold": 0.01, "factor": 0.5, "patience": 0,
                                                               "verbose": True, "min_lr": 1e-05, "threshold_mode": "rel"}},
        "trainer": {"name": "LayerwiseTrainer", "epochs": 1, "save_dir": save_dir, "save_period": 100, "verbosity": 0,
                    "monitor": "off", "accumulation_steps": 1, "log_step": 100, "do_validation_interval": 100,
                    "len_epoch": len_epoch, "tensorboard": False, "dtype": dtype},
        "pruning": {"args": {"dilation": 5, "padding": 20, "kernel_size": 9}, "pruning_plan": ent, "hint": ent, "unfreeze": ent},
        "weight_scheduler": {"alpha": {"value": 0.0001, "anneal_rate": 2, "max": 0}, "beta": {"value": 0.99, "anneal_rate": 0.95, "min": 0.99},
                             "gamma": {"value": 1, "anneal_rate": 1}},
    }


def gscnn_key_inventory():
    with open(os.path.join(_HERE, "golden", "gscnn_keys.json")) as f:
        return json.load(f)


def seeded_gscnn_sd(dtype=torch.float32):
    """Seeded GSCNN(19) state dict (prefix 'gscnn.', as tools/make_golden.py:g_gscnn filled the reference's module)."""
    sd = {}
    for k, shape in gscnn_key_inventory().items():
        if k.endswith("num_batches_tracked"):
            continue
        sd[k] = seeded_value("gscnn." + k, torch.empty(shape)).to(dtype)
    return sd


def canny_stub_map(shape, seed):
    """The seeded 0/255 edge map the GSCNN goldens were generated with in place of cv2.Canny (tools/make_golden.py)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) < 0.12).float() * 255.0
