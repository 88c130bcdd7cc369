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
